#!/usr/bin/env python3
"""Throughput of OTTER's observation -> action-chunk path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic observations: per GPU 64 concurrent episodes, each
observation two 224x224 uint8 camera images + a tokenised prompt + a 10-d proprio vector -> a (12, 10) action chunk,
16-bit MFMA towers + fp32 MFMA policy, history window 12 with the GPU-resident K/V cache (episodes restart every 24 steps, so the timed
steps cover the cached fill phase and the recompute-after-slide phase; the text tower runs once per episode).
Inputs are resident in HBM before the timed region.  One process per GPU (torchrun for N > 1), observations sharded
by episode, no collective on the data path; value = observations of all ranks / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_EPISODES = 64
EPISODE_LEN = 24            # 12 cached steps + 12 sliding-window steps, then reset
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X
PEAK_HBM_GBS = 8000.0       # HBM3E peak
POOL_UNIT_BYTES = 256 * 768 * 2 + 15 * 768 * 2 + 15 * 768 * 2     # 439,296: 16-bit patches in + text in + out per (obs, camera)
GEMM_SOURCES = ("otter_amd/csrc/gemm256.hip", "otter_amd/csrc/gemm.hip", "otter_amd/csrc/common.hpp")


def kernel_sources_sha256() -> str:
    """Fingerprint of the GEMM kernel sources: a counter file under profiles/ is only quoted while it matches."""
    import hashlib
    h = hashlib.sha256()
    for rel in GEMM_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_counters_of_this_kernel():
    """(traffic bytes per launch, MFMA-busy fraction, graphics clock in GHz, source tag) from the newest
    profiles/*_gemm_traffic.json -- the PMC passes of this same command, collected by scripts/collect_profiles.sh -- or
    (None, None, None, reason) when that file was
    made from other kernel sources than the ones in this tree (the counters cannot be collected inside an un-profiled run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))
    if not files:
        return None, None, None, "no profiles/*_gemm_traffic.json"
    path = files[-1]
    try:
        with open(path) as fh:
            pmc = json.load(fh)
        if pmc.get("kernel_sources_sha256") != kernel_sources_sha256():
            print(f"bench: {os.path.basename(path)} was collected from other GEMM sources than this tree's: roofline.traffic left null "
                  "(re-run scripts/collect_profiles.sh)", file=sys.stderr)
            return None, None, None, f"profiles/{os.path.basename(path)} is stale (kernel sources changed since it was collected)"
        return pmc["traffic_bytes_per_launch"], pmc.get("mfma_busy_fraction_of_simd_cycles"), pmc.get("gfx_clock_ghz_in_kernel"), \
            f"profiles/{os.path.basename(path)} (rocprofv3 --pmc passes of this command; kernel sources sha256 {pmc['kernel_sources_sha256'][:12]})"
    except (OSError, KeyError, ValueError) as exc:
        return None, None, None, f"unreadable counter file: {exc}"


def build_model(device, seed=0, clip_dtype=torch.float16, sd=None):
    """Default-config OTTER with random-init weights (``sd``: reuse another instance's checkpoint instead of drawing one)."""
    from otter_amd.config import ModelConfig, SharedConfig
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    mc, sc = ModelConfig(), SharedConfig()
    model = OTTER(mc, sc)
    if sd is None:
        sd = random_state_dict(otter_specs(mc, sc), seed=seed)
    model.load_state_dict(sd, strict=True)
    model.clip_compute_dtype = clip_dtype
    return model.to(device).eval(), sd, mc, sc


def synthetic_tokens(n, rng):
    """Pre-tokenised prompts (no BPE vocabulary file ships with the repo): SOT + ids + EOT, zero padded; two prompts cycled."""
    base = []
    for length in (5, 4):
        row = np.zeros(77, dtype=np.int32)
        row[0] = 49406
        row[1:1 + length] = rng.integers(1, 49406, length)
        row[1 + length] = 49407
        base.append(row)
    return torch.from_numpy(np.stack([base[i % 2] for i in range(n)]))


class Workload:
    def __init__(self, device, seed, clip_dtype=torch.float16, sd=None):
        from otter_amd.engine import BatchedRollout
        self.model, self.sd, self.mc, self.sc = build_model(device, clip_dtype=clip_dtype, sd=sd)
        self.dev = self.model.device_model()
        self.n = PER_GPU_EPISODES
        rng = np.random.default_rng(seed)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.n_sets = 4
        self.images = [torch.randint(0, 256, (self.n * 2, 3, 224, 224), dtype=torch.uint8, generator=g).to(device)
                       for _ in range(self.n_sets)]
        self.proprio = [torch.randn(self.n, 10, generator=g).to(device) for _ in range(self.n_sets)]
        self.tokens = synthetic_tokens(self.n, rng)
        self.rollout = BatchedRollout(self.dev, self.n)
        self.t = 0
        self.text = None

    def step(self):
        if self.t % EPISODE_LEN == 0:
            self.rollout.reset()
            self.dev.clip._text_cache.clear()                       # a new episode pays for its text tower
            self.text = self.dev.clip.text_tokens_cached(self.tokens)
        i = self.t % self.n_sets
        out = self.rollout.step(self.images[i], self.text, self.proprio[i])
        self.t += 1
        return out


def measure_with_h2d(work: "Workload", steps: int, warmup: int = 2) -> float:
    """Seconds for `steps` steps when the uint8 frames and the proprioception start in PINNED HOST memory every step (what
    a camera driver hands over): uploads run on a copy stream one step ahead of the compute stream, two device buffers.
    `warmup` untimed steps (at least 12) go through the same pipeline first: the first uploads of a process are slow
    (several take 7 ms instead of the 0.34 ms of a 19.3 MB DMA transfer, about 30 ms in all), which belongs to
    start-up, not to a step."""
    dev = work.images[0].device
    host_img = [t.cpu().pin_memory() for t in work.images]
    host_prop = [t.cpu().pin_memory() for t in work.proprio]
    d_img = [torch.empty_like(work.images[0]) for _ in range(2)]
    d_prop = [torch.empty_like(work.proprio[0]) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream(dev)

    def upload(t):
        b, i = t % 2, t % work.n_sets
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[b])              # the step that read this buffer two steps ago is done
            d_img[b].copy_(host_img[i], non_blocking=True)
            d_prop[b].copy_(host_prop[i], non_blocking=True)
            ready[b].record(copy_stream)

    def run(n):
        for b in range(2):
            consumed[b].record(main)
        work.t = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upload(0)
        for t in range(n):
            if t + 1 < n:
                upload(t + 1)
            b = t % 2
            main.wait_event(ready[b])
            if work.t % EPISODE_LEN == 0:
                work.rollout.reset()
                work.dev.clip._text_cache.clear()
                work.text = work.dev.clip.text_tokens_cached(work.tokens)
            work.rollout.step(d_img[b], work.text, d_prop[b])
            consumed[b].record(main)
            work.t += 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    if warmup > 0:
        run(max(warmup, 12))             # measured: the slow uploads (7 ms instead of 0.34) all fall in the first dozen steps
    return run(steps)


class StubWorkload:
    """OTTER_BENCH_STUB=1: the launch wiring of this file (rank / world parsing, process group, fenced timing, rank-0 JSON
    line, tear-down) rehearsed on a machine without GPUs -- a step sleeps instead of computing.  Never a measurement."""

    def __init__(self, rank: int):
        self.n, self.rank, self.t = PER_GPU_EPISODES, rank, 0

    def step(self):
        time.sleep(0.002 * (1 + self.rank))
        self.t += 1


class TrainWorkload:
    """Training step of the reference's data-parallel configuration: global batch 128 windows over 8 GPUs = 16 windows of
    12 observations per GPU and step (384 camera images), frozen towers + hand-written backward of the 31 M trainable
    parameters + gradient all-reduce + AdamW."""
    PER_GPU_WINDOWS = 16

    def __init__(self, device, seed, clip_dtype=torch.float16, sd=None):
        from otter_amd.trainer import Trainer
        self.model, self.sd, self.mc, self.sc = build_model(device, clip_dtype=clip_dtype, sd=sd)
        self.model.train()
        self.trainer = Trainer(self.model, lr=3e-4, weight_decay=0.01, warmup_steps=10, num_steps=1000)
        self.dev = None
        b, t = self.PER_GPU_WINDOWS, self.sc.seq_length
        self.n = b * t                                            # observations per step
        g = torch.Generator(device="cpu").manual_seed(seed)
        rng = np.random.default_rng(seed)
        self.images = torch.randint(0, 256, (b, t, 2, 3, 224, 224), dtype=torch.uint8, generator=g).to(device)
        self.proprio = torch.randn(b, t, 10, generator=g).to(device)
        self.gt = torch.randn(b, t, self.sc.action_horizon, self.mc.action_dim, generator=g).to(device)
        self.tokens = synthetic_tokens(b, rng)
        self.last = None

    def step(self):
        self.last = self.trainer.step(self.images, self.tokens, self.proprio, self.gt)
        return self.last


def measure_kernels(work, steps: int):
    """Per-launch device time of the GEMM and pooling kernels over `steps` extra steps, with HIP events recorded on
    the stream the kernels run on (torch's current stream)."""
    from otter_amd import ops
    rec = {"gemm": [], "tave": [], "stats": []}
    real_gemm, real_tave, real_stats, real_grs = ops.gemm, ops.text_aware_pool, ops.row_stats, ops.gemm_residual_stats_

    def timed(kind, fn, flops_or_bytes):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            rec[kind].append((e0, e1, flops_or_bytes(a, k, out), 0.0))
            return out
        return wrapper

    def gemm_wrapper(a, wp, n, bias=None, *, act=0, out_f32=False, residual=None, out=None, force_tile_128=False,
                     ln_stats=None, ln_colsum=None):
        """Times the dominant kernel only: a big 16-bit GEMM is issued as its whole 256-row tiles (the 256x256 kernel, timed)
        and its remainder rows (the 64x64 kernel, not timed) -- rows are independent, so the results are those of one call."""
        m, kk = a.shape
        if a.dtype == torch.float32 or m < 1024 or n < 256 or force_tile_128:
            return real_gemm(a, wp, n, bias, act=act, out_f32=out_f32, residual=residual, out=out, force_tile_128=force_tile_128,
                             ln_stats=ln_stats, ln_colsum=ln_colsum)
        m_main = m // 256 * 256
        if out is None:
            odt = residual.dtype if residual is not None else (torch.float32 if out_f32 else a.dtype)
            out = torch.empty(m, n, dtype=odt, device=a.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_gemm(a[:m_main], wp, n, bias, act=act, out_f32=out_f32, residual=None if residual is None else residual[:m_main], out=out[:m_main],
                  ln_stats=None if ln_stats is None else ln_stats[:m_main], ln_colsum=ln_colsum)
        e1.record()
        k_real = 588 if kk == 640 else kk          # algorithmic FLOPs: the un-padded K of the patch embedding
        out_bytes = out.element_size()
        nbytes = m_main * kk * 2 + n * kk * 2 + m_main * n * out_bytes + (m_main * n * residual.element_size() if residual is not None else 0)
        rec["gemm"].append((e0, e1, 2.0 * m_main * n * k_real, float(nbytes)))
        if m_main < m:
            real_gemm(a[m_main:], wp, n, bias, act=act, out_f32=out_f32, residual=None if residual is None else residual[m_main:], out=out[m_main:],
                      ln_stats=None if ln_stats is None else ln_stats[m_main:], ln_colsum=ln_colsum)
        return out

    def tave_bytes(a, k, out):
        # compulsory traffic per (observation, camera) at 16-bit: patches in + text in + out
        patches, text = a[0], a[1]
        units = patches.shape[0]
        p, d = (k.get("n_patches") or patches.shape[1]), patches.shape[2]
        kq = text.shape[1]
        return units * (p * d * 2 + kq * d * 2 + kq * d * 2)

    def residual_stats_wrapper(a, wp, n, bias, x, eps=1e-5):
        """The residual product and the statistics behind it as the separate launches they are made of (same results): the
        256x256 kernel is timed on its own, the statistics pass as `stats`."""
        ops.gemm(a, wp, n, bias, residual=x, out=x)
        return ops.row_stats(x, eps)

    ops.gemm = gemm_wrapper
    ops.gemm_residual_stats_ = residual_stats_wrapper
    ops.text_aware_pool = timed("tave", real_tave, tave_bytes)
    # row statistics of the folded LayerNorms: one read of the residual stream, 8 bytes out per row
    ops.row_stats = timed("stats", real_stats, lambda a, k, out: a[0].numel() * a[0].element_size() + out.numel() * 4)
    try:
        for _ in range(steps):
            work.step()
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.text_aware_pool, ops.row_stats, ops.gemm_residual_stats_ = real_gemm, real_tave, real_stats, real_grs
    out = {}
    for kind, items in rec.items():
        ms = [e0.elapsed_time(e1) for e0, e1, _, _ in items]
        out[kind] = dict(launches=len(items), total_ms=float(sum(ms)), total_work=float(sum(w for _, _, w, _ in items)),
                         total_bytes=float(sum(b for _, _, _, b in items)))
    return out


PEAK_COPY_GBS = 6290.0      # what a float4 device copy reaches on this chip (79 % of the nominal 8 TB/s): the practical streaming ceiling
POOL_SOURCES = ("otter_amd/csrc/text_aware_pool.hip", "otter_amd/csrc/common.hpp")


def pool_sources_sha256() -> str:
    import hashlib
    h = hashlib.sha256()
    for rel in POOL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pool_traffic_counters():
    """(counter bytes per launch of the 2048-unit pooling launch with 16-bit edges, source tag) from the newest
    profiles/*_pool_traffic.json (FETCH_SIZE and WRITE_SIZE passes of scripts/pool_saturated.py), or (None, reason) when the
    kernel source changed since."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pool_traffic.json")))
    if not files:
        return None, "no profiles/*_pool_traffic.json"
    try:
        with open(files[-1]) as fh:
            pmc = json.load(fh)
        if pmc.get("kernel_sources_sha256") != pool_sources_sha256():
            return None, f"profiles/{os.path.basename(files[-1])} is stale (kernel source changed since it was collected)"
        return pmc["traffic_bytes_per_launch"], f"profiles/{os.path.basename(files[-1])} ({pmc.get('units')} units, 16-bit text in / result out)"
    except (OSError, KeyError, ValueError) as exc:
        return None, f"unreadable counter file: {exc}"


def pool_saturated(device, dtype, units: int = 2048, reps: int = 10, io16: bool = False):
    """The pooling kernel alone on enough (observation, camera) units to occupy every CU (the step's own launch has
    128 units: one per workgroup, half the CUs idle): compulsory bytes / launch time, HIP events on the launch stream.
    ``io16``: text tokens in and result out in the patches' 16-bit type (the kernel then moves exactly the 439,296
    algorithmic bytes per unit); otherwise both are fp32 as the step uses them (485,376 bytes moved per unit)."""
    from otter_amd import ops
    p, d, kq, cams = 256, 768, 15, 2
    g = torch.Generator(device=device).manual_seed(5)
    patches = torch.randn(units, p, d, generator=g, device=device).to(dtype)
    text = torch.randn(units // cams, kq, d, generator=g, device=device)
    if io16:
        text = text.to(dtype)
    odt = dtype if io16 else torch.float32
    pos = torch.randn(cams, p, d, generator=g, device=device)
    packed = torch.stack([ops.pos_emb_pack(pos[c].contiguous(), dtype) for c in range(cams)])
    temp = torch.full((cams,), 0.07, device=device)
    for _ in range(2):
        ops.text_aware_pool(patches, text, packed, temp, units // cams, cams, 1, out_dtype=odt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.text_aware_pool(patches, text, packed, temp, units // cams, cams, 1, out_dtype=odt)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    gbs = units * POOL_UNIT_BYTES / (us * 1e-6) / 1e9               # the algorithmic 439,296 B per unit
    edge = 2 if io16 else 4
    moved_unit = p * d * 2 + 2 * kq * d * edge
    moved = units * moved_unit / (us * 1e-6) / 1e9                  # what this launch moves
    return {"units": units, "edges": "16-bit text in / result out" if io16 else "fp32 text in / result out", "avg_launch_us": us,
            "achieved": gbs, "frac": gbs / PEAK_HBM_GBS, "frac_of_copy_ceiling": gbs / PEAK_COPY_GBS, "bytes_per_unit": POOL_UNIT_BYTES,
            "moved_gbs": moved, "moved_frac": moved / PEAK_HBM_GBS, "moved_bytes_per_unit": moved_unit}


def library_gemm_rate(device, dtype, m: int = 32768, reps: int = 10):
    """Measuring stick, not part of the path: the vendor library's plain products (torch.mm = hipBLASLt; no bias, activation
    or residual) on the four GEMM shapes of one CLIP block, weighted as the block runs them.  TFLOP/s."""
    flops = ms = 0.0
    for n, k in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        a = torch.randn(m, k, device=device).to(dtype)
        w = (torch.randn(n, k, device=device) / k ** 0.5).to(dtype)
        out = torch.empty(m, n, device=device, dtype=dtype)
        for _ in range(3):
            torch.mm(a, w.t(), out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.mm(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / reps
        flops += 2.0 * m * n * k
    return flops / (ms * 1e-3) / 1e12


def cpu_baseline(work: Workload, budget_s: float = 15.0):
    """fp32 CPU oracle (a port of the reference's path, validated against its fixtures) on this host's cores."""
    from oracle import otter_ref as R
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))
    cores = torch.get_num_threads()
    ro = R.Rollout(work.sd, R.Dims())
    imgs = work.images[0][:2].cpu()
    tok = work.tokens[0]
    prop = work.proprio[0][0].cpu()
    ro.step(R.clip_transform(imgs), tok, prop)          # warm-up (thread pools, allocations)
    n, t0 = 0, time.perf_counter()
    while True:
        ro.step(R.clip_transform(imgs), tok, prop)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    out = dict(value=n / el, unit="obs/s", cores=cores, kind="port",
               sample=f"{n} single-episode observations (2 cams 224^2, default config, fp32 torch-CPU oracle, {el:.1f} s)")
    # a batched point: 8 observations per call through the oracle's full-sequence forward (B = 8, T = 1), a few calls
    b = 8
    imgs8 = R.clip_transform(work.images[0][: 2 * b].cpu().view(b, 1, 2, 3, 224, 224))
    tok8, prop8 = work.tokens[:b], work.proprio[0][:b].cpu().view(b, 1, -1)
    with torch.no_grad():
        R.forward(imgs8, tok8, prop8, work.sd, R.Dims())
        calls, t0 = 0, time.perf_counter()
        while True:
            R.forward(imgs8, tok8, prop8, work.sd, R.Dims())
            calls += 1
            el8 = time.perf_counter() - t0
            if el8 >= budget_s / 2 or calls >= 8:
                break
    out["batch8"] = dict(value=calls * b / el8, unit="obs/s", sample=f"{calls} calls of {b} observations (B = 8, T = 1) in {el8:.1f} s, same oracle and cores")
    return out


# ------------------------------------------------------------------------------- the other configurations, one block each
IMAGE_GFLOP = 158.11         # tower work one 224^2 image needs (patch embed + 23 blocks + last attention + patch projection)
TEXT_GFLOP_15 = 2.57         # text tower over the first 15 positions of one prompt


def _timed(fn, steps: int) -> float:
    """Seconds for `steps` calls of fn, device drained on both sides."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _step_roofline(tflop_per_step: float, seconds_per_step: float) -> dict:
    """Whole-step MFMA roofline: the towers' algorithmic FLOPs of one step over the step's wall time (everything else in
    the step -- attention softmax, pooling, policy, launch gaps -- counts as time and not as work)."""
    tf = tflop_per_step / seconds_per_step
    return {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS,
            "work": "algorithmic tower FLOPs of the step / wall time of the step"}


def bf16_point(device, base: "Workload", steps: int = 12, warmup: int = 3) -> dict:
    """The same workload with bf16 MFMA operands in the vision tower (what the configuration names; 99 % of the work) and an
    fp32 residual stream; the text tower, run once per prompt, keeps fp16 operands -- its bf16 rounding alone exceeds the
    1e-3 action tolerance (profiles/r03_bf16_error_attribution.txt), with it in fp16 the configuration meets the tolerance
    (tests/test_gpu_model.py).  Also the largest action difference between the two operand types on one identical first
    step of an episode."""
    w = Workload(device, seed=0, clip_dtype=torch.bfloat16, sd=base.sd)
    w.images, w.proprio, w.tokens = base.images, base.proprio, base.tokens
    for _ in range(warmup):
        w.step()
    w.t = 0
    el = _timed(w.step, steps)
    base.t = w.t = 0
    diff = float((base.step().float() - w.step().float()).abs().max())
    base.t = 0
    out = {"value": w.n * steps / el, "unit": "obs/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "dtype": "bf16",
           "roofline": _step_roofline(2 * w.n * IMAGE_GFLOP / 1e3, el / steps), "max_abs_action_diff_vs_f16_operands": diff,
           "note": "bf16 vision-tower operands with an fp32 residual stream, fp16 text tower (run once per prompt); held to the "
                   "path's 1e-3 action tolerance by the parity tests"}
    del w
    return out


def config3_train(device, base: "Workload", clip_dtype, steps: int = 3, warmup: int = 2) -> dict:
    """One data-parallel rank of the global-batch-128 training step (16 windows x 12 observations = 384 images)."""
    w = TrainWorkload(device, seed=0, clip_dtype=clip_dtype, sd=base.sd)
    for _ in range(warmup):
        w.step()
    el = _timed(w.step, steps)
    out = {"value": w.n * steps / el, "unit": "obs/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "per_gpu_batch": w.n,
           "roofline": _step_roofline(2 * w.n * IMAGE_GFLOP / 1e3, el / steps), "last": w.last,
           "workload": "training step: frozen 16-bit MFMA towers, fp32 hand-written backward of the 31M trainable parameters, flat-buffer "
                       "gradient all-reduce (a no-op on one rank), AdamW; 16 windows x 12 two-camera observations per GPU"}
    del w
    return out


def config4_openclip_extract(device, clip_dtype, pairs: int = 256, steps: int = 5, warmup: int = 2) -> dict:
    """OpenCLIP ViT-L/14 (GELU) text-aware feature extraction on `pairs` (image, prompt) pairs per step: both towers, the
    pooled text feature as a 16th query, temperature 0.005 (script/open_clip_visualization.py:38-117)."""
    from otter_amd import ops
    from otter_amd.config import clip_arch
    from otter_amd.device_model import ClipDevice
    from otter_amd.params import clip_specs, random_state_dict, sincos_position_embedding
    arch = clip_arch("open_clip:ViT-L-14")
    sd = random_state_dict(clip_specs(arch), seed=3)
    clip = ClipDevice(sd, arch, 15, device, clip_dtype)
    g = torch.Generator(device="cpu").manual_seed(4)
    images = torch.randint(0, 256, (pairs, 3, 224, 224), dtype=torch.uint8, generator=g).to(device)
    ids = synthetic_tokens(pairs, np.random.default_rng(4)).to(device)
    pos_t = ops.pos_emb_pack(sincos_position_embedding(256, arch.embed_dim).to(device), clip_dtype)[None].contiguous()
    temp = torch.full((1,), 0.005, device=device)

    def step():
        tokens, p, _ = clip.vision_tokens(images)
        tf = clip.text_tokens(ids, pooled_row=True)                       # (pairs, 16, D): no prompt cache, as the script runs it
        return ops.text_aware_pool(tokens, tf, pos_t, temp, pairs, 1, 1, n_patches=p)

    for _ in range(warmup):
        step()
    el = _timed(step, steps)
    return {"value": pairs * steps / el, "unit": "pairs/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "pairs_per_step": pairs,
            "roofline": _step_roofline(pairs * (IMAGE_GFLOP + TEXT_GFLOP_15) / 1e3, el / steps),
            "workload": "OpenCLIP ViT-L/14 (GELU-erf MLP) image + text towers and the fused text-aware pooling with 16 queries, "
                        "random-init weights, prompts tokenised ahead"}


def config5_long_history(device, base: "Workload", window: int = 32, warmup: int = 2) -> dict:
    """64 concurrent episodes (one GPU's share of 512) with a history window of 32: `window` steps filling the per-layer
    K/V cache, then `window` steps with the window sliding (whole-window recompute: 2048 rows through the policy)."""
    from otter_amd.device_model import OtterDevice
    from otter_amd.engine import BatchedRollout
    from otter_amd.params import sinusoidal_table
    m = base.model
    sd = dict(base.sd)
    sd["pe.pe"] = sinusoidal_table(window, base.mc.transformer_dim)
    dev5 = OtterDevice(sd, m.clip_arch, first_k=m.first_k_tokens, pooling_heads=m.pooling_heads, transformer_heads=m.transformer_heads,
                       num_cameras=len(m.camera_keys), seq_length=window, action_horizon=m.action_horizon, action_dim=m.action_dim,
                       pool_true_text=m.pool_true_text, device=device, clip_dtype=m.clip_compute_dtype, clip=base.dev.clip)
    ro = BatchedRollout(dev5, base.n)
    text = base.dev.clip.text_tokens_cached(base.tokens)
    t = [0]

    def step():
        i = t[0] % base.n_sets
        t[0] += 1
        return ro.step(base.images[i], text, base.proprio[i])

    for _ in range(warmup):
        step()
    ro.reset()
    fill = _timed(step, window)
    slide = _timed(step, window)
    el, steps = fill + slide, 2 * window
    return {"value": base.n * steps / el, "unit": "obs/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
            "ms_per_step_fill": 1e3 * fill / window, "ms_per_step_slide": 1e3 * slide / window, "per_gpu_episodes": base.n, "history": window,
            "roofline": _step_roofline(2 * base.n * IMAGE_GFLOP / 1e3, el / steps),
            "workload": f"{base.n} episodes per GPU, history window {window}: {window} cached-K/V fill steps + {window} sliding steps"}


def self_launch(args) -> int:
    """`bench.py --gpus N` started bare (no torchrun around it): this parent, which has made no GPU call, starts the N
    ranks as a child `python -m torch.distributed.run ...` of the same command and relays rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    for ln in res.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    return res.returncode if res.returncode != 0 or lines else 1


def timed_composition(steps: int) -> str:
    fill = sum(1 for t in range(steps) if t % EPISODE_LEN < 12)
    return f"{fill} fill + {steps - fill} slide steps"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--clip-dtype", choices=["f16", "bf16"], default="f16",
                    help="16-bit MFMA operand type of the CLIP towers (same MFMA rate; fp16 tracks the fp32 reference 8x closer)")
    ap.add_argument("--workload", choices=["infer", "train"], default="infer",
                    help="infer: batched rollout (the headline metric); train: one data-parallel training step per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side blocks of the N=1 line (bf16 operands, training step, OpenCLIP extraction, history 32)")
    ap.add_argument("--no-reference-points", action="store_true",
                    help="skip the two side measurements (vendor-library GEMM rate, pooling kernel at a chip-filling size): profiles of "
                         "the step then hold the step's kernels only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))              # started bare: become the launcher of N ranks
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    # One process per GPU.  OTTER_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks
    # share the visible devices and synchronise through the host (RCCL refuses two ranks on one device).
    stub = os.environ.get("OTTER_BENCH_STUB", "0") == "1"
    backend = "gloo" if stub else os.environ.get("OTTER_BENCH_BACKEND", "nccl")
    import torch.distributed as dist
    if stub:
        device = torch.device("cpu")
    else:
        local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    sync_device = device if backend == "nccl" else torch.device("cpu")

    clip_dtype = torch.float16 if args.clip_dtype == "f16" else torch.bfloat16
    if stub:
        work = StubWorkload(rank)
        args.no_kernel_timing = args.no_cpu_baseline = args.no_reference_points = True
    else:
        work = Workload(device, seed=rank, clip_dtype=clip_dtype) if args.workload == "infer" else TrainWorkload(device, seed=rank, clip_dtype=clip_dtype)
    from otter_amd.sharding import timed_region
    for _ in range(args.warmup):
        work.step()
    work.t = 0                                           # timed region starts at an episode boundary
    # barrier + device sync on both sides of exactly `steps` steps; MAX over ranks
    elapsed = timed_region(work.step, args.steps, world=world, sync=(lambda: None) if stub else torch.cuda.synchronize, device=sync_device)

    result = None
    if rank == 0:
        obs = world * work.n * args.steps
        result = {
            "metric": "observations/sec (obs->action chunk)", "value": obs / elapsed, "unit": "obs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.clip_dtype, "data": "synthetic" if not stub else "none (OTTER_BENCH_STUB rehearsal: no compute)",
            "config": {"workload": f"OTTER inference, batch 64 two-camera 224^2 observations per GPU, history 12 with GPU-resident KV cache; episodes of "
                                   f"{EPISODE_LEN} steps (12 filling the cache + 12 sliding), the {args.steps} timed steps start at an episode boundary = "
                                   f"{timed_composition(args.steps)}; default ViT-L/14 config, random-init weights; towers on 16-bit MFMA with "
                                   f"{'fp16' if args.clip_dtype == 'f16' else 'bf16'} operands"
                                   + (" (fp16: same width and MFMA rate as bf16, 8x smaller rounding, 5 % faster because the residual stream can stay "
                                      "16-bit; the bf16 point of the same workload is the value_bf16 block)"
                                      if args.clip_dtype == "f16" else " in the vision tower, fp32 residual stream, fp16 text tower")
                                   + ", fp32 MFMA policy", "per_gpu_batch": PER_GPU_EPISODES, "parallelism": f"dp{world} (episodes sharded, no collective)"},
        }
        if args.workload == "train" and not stub:
            result["config"] = {"workload": "OTTER training step (frozen 16-bit MFMA towers, fp32 hand-written backward of the 31M trainable parameters, "
                                            "flat-buffer gradient all-reduce, AdamW), 16 windows x 12 two-camera 224^2 observations per GPU, "
                                            "default ViT-L/14 config, random-init weights", "per_gpu_batch": work.n,
                                "parallelism": f"dp{world} (windows sharded, bucketed all-reduce of 31M fp32 gradients)", "last": work.last}
        if not args.no_kernel_timing:
            k = measure_kernels(work, 2)
            gm = k["gemm"]
            # dominant kernel: the bf16 MFMA GEMM (CLIP tower launches are >99.9% of its FLOPs)
            tf = gm["total_work"] / (gm["total_ms"] * 1e-3) / 1e12
            # hardware counters cannot be read from inside an un-profiled run: quoted from the PMC passes of this same
            # command, and only while the kernel sources are the ones those passes ran (else null)
            traffic, mfma_busy, pmc_clock, pmc_source = pmc_counters_of_this_kernel()
            result["roofline"] = {"bound": "mfma", "kernel": "gemm_256x256 (16-bit operands, CLIP towers; whole 256-row tiles)", "achieved": tf, "peak": PEAK_BF16_TFLOPS,
                                  "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "traffic": traffic,
                                  "traffic_source": pmc_source,
                                  "algorithmic_bytes_per_launch": gm["total_bytes"] / gm["launches"], "pmc_mfma_busy_frac": mfma_busy,
                                  # graphics clock inside the kernel by the same counter pass (cycles of the dispatch / its duration);
                                  # `peak` is the nominal figure at the 2.4 GHz boost clock
                                  "pmc_gfx_clock_ghz": pmc_clock,
                                  "launches": gm["launches"], "avg_launch_us": 1e3 * gm["total_ms"] / gm["launches"],
                                  "gemm_ms_per_step": gm["total_ms"] / 2}
            if world == 1 and args.workload == "infer" and not args.no_reference_points:
                lib_tf = library_gemm_rate(device, clip_dtype)
                result["roofline"]["vendor_library_plain_product"] = {"tflops": lib_tf, "frac_of_peak": lib_tf / PEAK_BF16_TFLOPS,
                                                                       "note": "torch.mm (hipBLASLt) on the block's four shapes, no epilogue; same box, same run"}
            tv = k["tave"]
            gbs = tv["total_work"] / max(tv["total_ms"] * 1e-3, 1e-12) / 1e9
            result["roofline_pool"] = {"bound": "hbm", "kernel": "tave_kernel", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": gbs / PEAK_HBM_GBS, "traffic": None, "launches": tv["launches"],
                                       "avg_launch_us": 1e3 * tv["total_ms"] / tv["launches"],
                                       "units_per_launch": 2 * work.n, "bytes_per_unit": POOL_UNIT_BYTES}
            st = k["stats"]
            if st["launches"]:
                sg = st["total_work"] / max(st["total_ms"] * 1e-3, 1e-12) / 1e9
                result["roofline_layernorm"] = {"bound": "hbm", "kernel": "row_stats16 (row statistics of the LayerNorms folded into the GEMMs)",
                                                "achieved": sg, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": sg / PEAK_HBM_GBS,
                                                "launches": st["launches"], "avg_launch_us": 1e3 * st["total_ms"] / st["launches"]}
            if world == 1 and args.workload == "infer" and not args.no_reference_points:
                # the kernel alone on a chip-filling launch: as the step calls it (fp32 edges) and moving exactly the
                # algorithmic bytes (16-bit edges); counter traffic of the latter from the PMC passes under profiles/
                result["roofline_pool"]["saturated"] = pool_saturated(device, clip_dtype)
                sat16 = pool_saturated(device, clip_dtype, io16=True)
                sat16["traffic"], sat16["traffic_source"] = pool_traffic_counters()
                result["roofline_pool"]["saturated_16bit_edges"] = sat16
                result["roofline_pool"]["peaks"] = {"hbm_nominal_gbs": PEAK_HBM_GBS, "device_copy_gbs": PEAK_COPY_GBS,
                                                    "note": "frac is against the nominal 8 TB/s; frac_of_copy_ceiling against the 6.29 TB/s a float4 device copy reaches on this chip"}
        if world == 1 and args.workload == "infer" and not args.no_reference_points:
            el = measure_with_h2d(work, args.steps, max(args.warmup, 1))
            result["value_incl_h2d"] = {"value": work.n * args.steps / el, "unit": "obs/s", "ms_per_step": 1e3 * el / args.steps,
                                        "note": "same steps with the uint8 frames (19.3 MB per step) and proprioception uploaded from pinned host "
                                                "memory every step on a copy stream, one step ahead, after 12 untimed steps through the same pipeline; never the headline value. "
                                                "Host work in front of the upload (PIL decode / resize, BPE) is not in it: the frames start as uint8 "
                                                "arrays at the model's resolution and the prompts as token ids"}
        if world == 1 and args.workload == "infer" and not stub and not args.no_other_configs:
            # the other configurations, each a block of its own (never the headline value); a failure in one is reported
            # in its block and does not take the line down
            for name, fn in (("value_bf16", lambda: bf16_point(device, work)),
                             ("config5_long_history", lambda: config5_long_history(device, work)),
                             ("config4_openclip_extract", lambda: config4_openclip_extract(device, clip_dtype)),
                             ("config3_train", lambda: config3_train(device, work, clip_dtype))):
                try:
                    result[name] = fn()
                except Exception as exc:                     # noqa: BLE001
                    result[name] = {"error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and args.workload == "infer":
            result["cpu_baseline"] = cpu_baseline(work)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
