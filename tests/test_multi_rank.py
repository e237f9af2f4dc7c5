"""N > 1 path on CPU: two gloo ranks, episodes sharded, no data-path collective (the per-rank compute is a stand-in:
the device kernels need a GPU; what is covered here is the sharding, the fenced max-over-ranks timing and the gather)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from otter_amd.sharding import episode_shard, gather_actions, timed_region


def test_episode_shard_partitions():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [episode_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        episode_shard(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_episodes, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = episode_shard(n_episodes, rank, world)
        ids = torch.arange(lo, hi, dtype=torch.float32)
        calls = []

        def step():                                   # rank 1 is the slow one
            calls.append(1)
            time.sleep(0.02 * (1 + 2 * rank))

        elapsed = timed_region(step, 5, world=world)
        assert len(calls) == 5
        actions = ids[:, None, None].expand(hi - lo, 12, 10).contiguous() + 0.5 * rank
        full = gather_actions(actions, n_episodes, rank, world)
        torch.save({"elapsed": elapsed, "full": full}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_episodes", [7, 64])
def test_two_ranks_gloo(tmp_path, n_episodes):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_episodes, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    # every rank reports the same, slowest-rank time: rank 1 sleeps 5 * 60 ms
    assert res[0]["elapsed"] == res[1]["elapsed"]
    assert res[0]["elapsed"] >= 0.29
    assert res[1]["full"] is None
    full = res[0]["full"]
    assert full.shape == (n_episodes, 12, 10)
    split = episode_shard(n_episodes, 0, world)[1]
    expect = torch.arange(n_episodes, dtype=torch.float32)
    expect[split:] += 0.5
    assert torch.equal(full[:, 0, 0], expect) and torch.equal(full[:, 11, 9], expect)


def test_single_rank_passthrough():
    x = torch.zeros(3, 12, 10)
    assert gather_actions(x, 3, 0, 1) is x
    assert timed_region(lambda: None, 3) >= 0.0


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from otter_amd.trainer import allreduce_flat
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * [0..999]
        allreduce_flat(flat, bucket_elems=300)                               # 4 buckets, the last one short
        torch.save(flat, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_two_ranks_gloo(tmp_path):
    """Data-parallel gradient exchange of the training step: bucketed SUM all-reduce of the flat gradient buffer."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = torch.arange(1000, dtype=torch.float32) * 3
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f"g{r}.pt"), want)


def _overlap_worker(rank, world, port, out_dir, algo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from otter_amd.trainer import BucketedExchange
        # five parameters in forward order a..e (e is produced after the tape, like the extraction temperatures); buckets of 300
        where = {"a": (0, 200), "b": (256, 556), "c": (576, 876), "d": (896, 996), "e": (996, 1000)}
        first_use = {"a": 0, "b": 3, "c": 5, "d": 9}
        flat = torch.zeros(1000)
        ex = BucketedExchange(flat, where, bucket_elems=300, algo=algo)
        assert ex.active and len(ex.bounds) == 4
        log = []
        ex.begin(first_use)
        for i in range(11, -1, -1):                    # the tape replays 12 closures, last to first
            for n, (lo, hi) in where.items():          # a parameter's gradient is written by the closures at or after its first use
                if first_use.get(n, -1) == i:
                    flat[lo:hi] = float(rank + 1) * (1 + "abcde".index(n))
            before = len(ex.order)
            ex.after_step(i)
            log.append((i, list(ex.order[before:])))
        flat[996:1000] = float(rank + 1) * 5           # "e": after the tape
        ex.finish()
        ex.wait()
        torch.save({"flat": flat, "log": log, "order": ex.order}, os.path.join(out_dir, f"o{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["all_reduce", "reduce_scatter_all_gather"])
def test_bucketed_exchange_overlaps_backward_two_ranks_gloo(tmp_path, algo):
    """Buckets go out from inside the backward pass, last bucket first, each as soon as every parameter with elements in it
    is final; the bucket holding a parameter that is produced after the tape waits for finish(); sums are those of one
    all-reduce over the whole buffer -- as all-reduces and as explicit reduce-scatter + all-gather pairs."""
    mp.spawn(_overlap_worker, args=(2, _free_port(), str(tmp_path), algo), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"o{r}.pt", weights_only=False) for r in range(2)]
    # buckets: [0,300) a+b, [300,600) b+c, [600,900) c+d, [900,1000) d+e
    # bucket 2 (c from step 5, d from step 9) is final after closure 5; bucket 1 (b, c) after closure 3; bucket 0 (a, b) after closure 0;
    # bucket 3 holds e -> finish()
    launched = {i: bs for i, bs in res[0]["log"] if bs}
    assert launched == {5: [2], 3: [1], 0: [0]} and res[0]["order"] == [2, 1, 0, 3]
    want = torch.zeros(1000)
    for n, (lo, hi) in {"a": (0, 200), "b": (256, 556), "c": (576, 876), "d": (896, 996), "e": (996, 1000)}.items():
        want[lo:hi] = 3.0 * (1 + "abcde".index(n))
    for r in range(2):
        assert torch.equal(res[r]["flat"], want)


def test_bench_launch_wiring_two_ranks(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), with the device step
    stubbed (OTTER_BENCH_STUB=1: gloo, a step sleeps): rank / world parsing, process group, fenced max-over-ranks timing,
    exactly one JSON line from rank 0, clean tear-down."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OTTER_BENCH_STUB="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak" and out["higher_is_better"]
    assert out["unit"] == "obs/s" and out["vs_baseline"] is None and "dp2" in out["config"]["parallelism"]
    # whole-job rate: both ranks' observations over the SLOWEST rank's time (rank 1 sleeps 4 ms per step, rank 0 2 ms)
    assert out["ms_per_step"] >= 4.0
    assert out["value"] == pytest.approx(2 * 64 * 6 / (out["ms_per_step"] * 6 / 1e3), rel=1e-6)
    # a world-size mismatch (a launcher that started the wrong number of ranks) is refused before anything is initialised
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "needs torchrun" in (bad.stderr + bad.stdout)


def test_bench_bare_invocation_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself (a child
    torch.distributed.run) and relays exactly one JSON line from rank 0 on its stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OTTER_BENCH_STUB"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and "dp2" in out["config"]["parallelism"]
    assert out["value"] == pytest.approx(2 * 64 * 4 / (out["ms_per_step"] * 4 / 1e3), rel=1e-6)


# ---------------------------------------------------------------------------------- data-parallel Trainer on two ranks
class _HostKernels:
    """The three device kernels Trainer.step launches, in torch on the host, so that the step's own logic (flat buffers,
    bucketed exchange, 1 / (scale * world) fold, skip-on-overflow, schedule) runs unchanged on CPU ranks."""

    def __init__(self, real):
        self._lib = real._lib

    @staticmethod
    def grad_stats(x, workspace=None):
        finite = torch.isfinite(x)
        return torch.stack([(x[finite].double() ** 2).sum().float(), (~finite).sum().float()])

    @staticmethod
    def adamw_step(p, g, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay, step, inv_scale=1.0, skip_flag=None):
        if skip_flag is not None and float(skip_flag) != 0:
            return
        g = g * inv_scale
        p.mul_(1 - lr * weight_decay)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / (1 - betas[1] ** step) ** 0.5).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - betas[0] ** step))


def _dp_case():
    import golden_inputs as GI
    from oracle import otter_ref as R
    case = dict(GI.E2E_CASES[0], B=4)                      # 4 windows: 2 per rank
    mc, sc = GI.e2e_configs(case)
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length, num_cameras=len(sc.camera_keys),
                  pool_true_text=mc.pool_true_text, use_mse=sc.mse_loss)
    return case, mc, sc, dims, GI.e2e_inputs(case)


def _oracle_forward_backward(model, dims):
    """Stand-in for TrainableTail.forward_backward: the fp32 CPU oracle differentiated by torch autograd."""
    from oracle import otter_ref as R

    def fb(images, ids, proprio, gt, mask, grads, grad_scale=1.0, exchange=None):
        # (the stand-in reports no progress to `exchange`: Trainer.step then sends every bucket after the pass)
        sd = dict(model.state_dict(keep_vars=True))
        loss = R.forward(images, ids, proprio, sd, dims, gt)
        names = list(grads)
        for n, g in zip(names, torch.autograd.grad(loss * grad_scale, [sd[n] for n in names])):
            grads[n].add_(g)
        return loss.detach().view(1)
    return fb


def _trainer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from otter_amd import ops
        from otter_amd.otter import OTTER
        from otter_amd.params import otter_specs, random_state_dict
        from otter_amd.trainer import Trainer
        case, mc, sc, dims, (images, text, proprio, gt) = _dp_case()
        model = OTTER(mc, sc)
        model.load_state_dict(random_state_dict(otter_specs(mc, sc), seed=case["seed"]), strict=True)
        tr = Trainer(model, lr=3e-4, weight_decay=0.01, warmup_steps=1, num_steps=20, bucket_elems=100_000)
        assert tr.world == world
        tr.ops = _HostKernels(ops)
        tr.tail.forward_backward = _oracle_forward_backward(model, dims)
        lo, hi = episode_shard(images.shape[0], rank, world)          # per-rank batch = global / world (script/train.py:152)
        log = [tr.step(images[lo:hi], text[lo:hi], proprio[lo:hi], gt[lo:hi]) for _ in range(3)]
        torch.save({"flat": tr.flat_p.clone(), "log": log, "names": {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}},
                   os.path.join(out_dir, f"t{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_trainer_two_ranks_match_single_process_training(tmp_path):
    """Two data-parallel ranks (gloo), each with half of the windows, through Trainer.step's real exchange (bucketed SUM
    all-reduce, mean folded into the unscale factor, loss scaling, schedule) end up with the parameters that ONE process
    reaches with torch.optim.AdamW on the whole batch -- DistributedDataParallel's contract (script/train.py:116-119,
    otter/util/misc.py:334-345)."""
    from oracle import otter_ref as R
    from otter_amd.params import otter_specs, random_state_dict
    from otter_amd.trainer import cosine_with_warmup
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"t{r}.pt", weights_only=False) for r in range(2)]
    assert torch.equal(res[0]["flat"], res[1]["flat"])                 # replicas stay bit-identical
    case, mc, sc, dims, (images, text, proprio, gt) = _dp_case()
    sd = random_state_dict(otter_specs(mc, sc), seed=case["seed"])
    names = list(res[0]["names"])
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    decay = [sdg[n] for n in names if not (sdg[n].dim() == 1 or n.endswith(".bias"))]
    no_decay = [sdg[n] for n in names if (sdg[n].dim() == 1 or n.endswith(".bias"))]
    opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.01}], lr=3e-4)
    for it in range(3):
        for grp in opt.param_groups:
            grp["lr"] = 3e-4 * cosine_with_warmup(it, 1, 20)
        opt.zero_grad()
        loss = R.forward(images, text, proprio, sdg, dims, gt)
        loss.backward()
        norm = sum(float((p.grad.double() ** 2).sum()) for p in decay + no_decay) ** 0.5
        opt.step()
        # each rank logs its own shard's loss; the mean over ranks is the global loss; the norm is that of the MEAN gradient
        shard_mean = 0.5 * (res[0]["log"][it]["loss"] + res[1]["log"][it]["loss"])
        assert abs(shard_mean - loss.item()) <= 1e-5
        assert res[0]["log"][it]["grad_norm"] == pytest.approx(norm, rel=1e-4)
        assert not res[0]["log"][it]["skipped"] and res[0]["log"][it]["lr"] == pytest.approx(3e-4 * cosine_with_warmup(it, 1, 20))
    # Adam normalises every element's step to ~lr (two steps of 3e-4 here), so fp32 summation-order noise in a tiny gradient
    # shows up magnified.  (A wrong mean would hide here -- Adam is invariant to the gradient's scale -- and is caught by
    # the gradient-norm comparison above; a missing exchange by the replicas' bit-equality.)
    worst = max((res[0]["names"][n] - sdg[n].detach()).abs().max().item() for n in names)
    assert worst <= 3e-5, worst
