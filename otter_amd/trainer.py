"""Data-parallel training step (``script/train.py:100-171``, ``otter/util/engine.py:15-66``, ``otter/util/misc.py``).

What the reference assembles from torch pieces -- ``add_weight_decay`` parameter groups, ``torch.optim.AdamW``,
``torch.amp.GradScaler`` inside ``NativeScalerWithGradNormCount``, the cosine-with-warmup ``LambdaLR`` and
``DistributedDataParallel`` -- is one object here, laid out for the device:

* every trainable tensor lives in ONE flat fp32 buffer (no-decay group first, decay group second; the module's
  parameters are views into it), with flat gradient / first-moment / second-moment buffers beside it;
* the hand-written backward (:mod:`otter_amd.train`) accumulates straight into the flat gradient buffer;
* ranks exchange gradients with a bucketed all-reduce over the flat buffer (RCCL on the GPUs; the SUM is turned
  into DistributedDataParallel's mean by folding 1/world into the unscale factor);
* gradient norm + non-finite check are one two-stage reduction, AdamW is one kernel launch per group and skips
  itself on the device when the check failed (GradScaler semantics without a host round trip before the update).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------ pure host-side pieces
def add_weight_decay(model, weight_decay: float = 1e-5, skip_list: Iterable[str] = ()) -> List[dict]:
    """``otter/util/misc.py:442-454``: 1-D tensors, ``*.bias`` and names in ``skip_list`` get no weight decay.
    (A 0-d tensor such as the extraction temperature has ``len(shape) == 0`` and therefore DOES decay, as upstream.)"""
    decay, no_decay = [], []
    skip = set(skip_list)
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (len(p.shape) == 1 or name.endswith(".bias") or name in skip) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def cosine_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """Multiplier of the base learning rate at scheduler step ``step`` (``otter/util/lr_sched.py:148-179``)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class LossScale:
    """``torch.amp.GradScaler`` bookkeeping (defaults: 2**16, x2 after 2000 clean steps, x0.5 on a non-finite gradient)."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval, self.enabled = growth_factor, backoff_factor, growth_interval, enabled
        self._good = 0

    def update(self, found_inf: bool) -> None:
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good == self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0

    def state_dict(self) -> dict:
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._good}

    def load_state_dict(self, sd: dict) -> None:
        self.scale, self._good = float(sd["scale"]), int(sd["_growth_tracker"])


FLAT_ALIGN = 64      # elements


def bucket_bounds(n: int, bucket_elems: int) -> List[Tuple[int, int]]:
    if bucket_elems <= 0:
        raise ValueError("bucket size must be positive")
    return [(lo, min(n, lo + bucket_elems)) for lo in range(0, n, bucket_elems)]


def allreduce_flat(flat: Tensor, bucket_elems: int = 8 << 20) -> None:
    """SUM-all-reduce a flat buffer across the default process group in buckets, all in flight together.
    xGMI is a point-to-point mesh: a few large buckets (default 32 MiB) keep every link busy without paying the
    per-collective latency hundreds of times."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in bucket_bounds(flat.numel(), bucket_elems)]
    for w in works:
        w.wait()


class BucketedExchange:
    """Gradient exchange overlapped with the backward pass.

    The flat gradient buffer is cut into buckets of ``bucket_elems`` elements.  A bucket's SUM all-reduce is launched (async,
    on the collective library's own stream) as soon as every parameter that has elements in it is final -- the backward
    tape reports its progress through :meth:`after_step` -- so the exchange of the layers that finish first runs beside
    the backward kernels of the rest, as DistributedDataParallel's bucket hooks do (``script/train.py:116-119``,
    ``otter/util/misc.py:335``).  Parameters are laid out in forward order and finish in reverse order: the buckets at the
    end of the buffer go first.  :meth:`finish` launches what is left (parameters whose gradient is produced after the
    tape, parameters the pass never touched), :meth:`wait` makes the current stream wait for all of them.

    ``algo``: "all_reduce" (the library picks ring or tree), or "reduce_scatter_all_gather": each bucket as an explicit
    reduce-scatter (every rank ends up with the sum of its 1/world slice) followed by an all-gather of the slices.  On a
    point-to-point xGMI mesh both halves are single-hop exchanges between every pair of GPUs, where a ring pushes the whole
    bucket 2 (world - 1) times through one link after the other.  The two give the same sums up to the order of addition.

    Without a process group (or on one rank, unless ``single_rank`` asks for the collectives anyway: a rehearsal of the
    RCCL path on a one-GPU box) every method is a no-op."""

    ALGOS = ("all_reduce", "reduce_scatter_all_gather")

    def __init__(self, flat: Tensor, where: Dict[str, Tuple[int, int]], bucket_elems: int = 8 << 20, single_rank: bool = False,
                 algo: str = "all_reduce"):
        if algo not in self.ALGOS:
            raise ValueError(f"exchange algorithm {algo!r}: one of {self.ALGOS}")
        self.flat = flat
        self.algo = algo
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank)
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        # collectives of one NCCL / RCCL communicator run in issue order on its stream, so the all-gather can be queued right
        # behind its reduce-scatter; other backends run queued work on a thread pool: there the all-gathers wait for wait()
        self._chained = self.active and dist.get_backend() == "nccl"
        self._shards: Dict[int, Tensor] = {}
        self._pending_gather: List[int] = []
        self.bounds = bucket_bounds(flat.numel(), bucket_elems)
        # names with elements in each bucket (a parameter may straddle two buckets)
        self.members: List[List[str]] = [[n for n, (a, b) in where.items() if a < hi and b > lo] for lo, hi in self.bounds]
        self.ready_at: List[int] = []
        self.launched: List[bool] = []
        self.works: list = []
        self.order: List[int] = []           # bucket indices in launch order (for tests and logs)

    def begin(self, first_use: Dict[str, int]) -> None:
        """``first_use``: parameter name -> tape position of its first use; a parameter that is not listed counts as never
        final before :meth:`finish`."""
        self.ready_at = [min((first_use.get(n, -1) for n in names), default=-1) for names in self.members]
        self.launched = [False] * len(self.bounds)
        self.works, self.order, self._pending_gather = [], [], []

    def _launch(self, b: int) -> None:
        self.launched[b] = True
        self.order.append(b)
        if not self.active:
            return
        lo, hi = self.bounds[b]
        n = hi - lo
        if self.algo == "all_reduce" or n % self.world:
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            return
        if b not in self._shards:
            self._shards[b] = torch.empty(n // self.world, dtype=self.flat.dtype, device=self.flat.device)
        self.works.append(dist.reduce_scatter_tensor(self._shards[b], self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        if self._chained:
            self.works.append(dist.all_gather_into_tensor(self.flat[lo:hi], self._shards[b], async_op=True))
        else:
            self._pending_gather.append(b)

    def after_step(self, i: int) -> None:
        """Every tape closure at position >= i has run."""
        for b in range(len(self.bounds) - 1, -1, -1):
            if not self.launched[b] and self.ready_at[b] >= i:
                self._launch(b)

    def finish(self) -> None:
        if not self.launched and self.bounds:
            self.begin({})
        for b in range(len(self.bounds) - 1, -1, -1):
            if not self.launched[b]:
                self._launch(b)

    def wait(self) -> None:
        for w in self.works:
            w.wait()
        self.works = []
        if self._pending_gather:
            works = [dist.all_gather_into_tensor(self.flat[self.bounds[b][0]:self.bounds[b][1]], self._shards[b], async_op=True)
                     for b in self._pending_gather]
            self._pending_gather = []
            for w in works:
                w.wait()


def flatten_parameters(groups: List[dict], device=None) -> Tuple[Tensor, List[Tuple[int, int, float]], Dict[int, Tuple[int, int]]]:
    """Move every parameter of the groups into one flat fp32 buffer (group after group) and re-point ``p.data`` at its
    slice.  Every tensor starts on a 256-byte boundary (the kernels use 16-byte vector accesses); the padding is zero
    and stays zero under AdamW.  Returns (flat, [(lo, hi, weight_decay) per group], {id(p): (lo, hi)})."""
    align = FLAT_ALIGN
    total = sum(-(-p.numel() // align) * align for g in groups for p in g["params"])
    first = next((p for g in groups for p in g["params"]), None)
    if first is None:
        raise ValueError("no trainable parameters")
    device = device or first.device
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    spans, where, off = [], {}, 0
    for g in groups:
        lo = off
        for p in g["params"]:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            where[id(p)] = (off, off + n)
            off += -(-n // align) * align
        spans.append((lo, off, float(g["weight_decay"])))
    return flat, spans, where


# ------------------------------------------------------------------------------------------------------ the trainer
class Trainer:
    """One object = optimizer + loss scaler + lr schedule + gradient exchange for an :class:`otter_amd.otter.OTTER`."""

    def __init__(self, model, *, lr: float = 3e-4, weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8,
                 warmup_steps: int = 0, num_steps: int = 1, accum_iter: int = 1, loss_scaling: bool = True,
                 bucket_elems: int = 8 << 20, overlap_exchange: bool = True, exchange_single_rank: bool = False,
                 exchange_algo: str = "all_reduce"):
        from . import ops
        from .train import TrainableTail
        self.ops = ops
        self.model = model
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        groups = add_weight_decay(model, weight_decay)
        self.flat_p, self.spans, where = flatten_parameters(groups)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.tail = TrainableTail(model)
        self.grads = {n: self.flat_g[where[id(p)][0]:where[id(p)][1]].view(p.shape) for n, p in self.tail.params.items() if id(p) in where}
        # gradient exchange: buckets of the flat buffer, launched from the backward tape as their parameters become final
        self.overlap_exchange = overlap_exchange
        self.exchange = BucketedExchange(self.flat_g, {n: where[id(p)] for n, p in self.tail.params.items() if id(p) in where},
                                         bucket_elems, single_rank=exchange_single_rank, algo=exchange_algo)
        self.base_lr, self.betas, self.eps = lr, betas, eps
        self.warmup_steps, self.num_steps, self.accum_iter = warmup_steps, num_steps, max(1, accum_iter)
        self.scaler = LossScale(enabled=loss_scaling)
        self.bucket_elems = bucket_elems
        self.iteration = 0            # data_iter_step of engine.py:44
        self.opt_steps = 0            # AdamW's step count (bias correction)
        self._stats_ws = None

    @property
    def lr(self) -> float:
        return self.base_lr * cosine_with_warmup(self.iteration, self.warmup_steps, self.num_steps)

    def step(self, images: Tensor, text: Tensor, proprio: Tensor, gt_actions: Tensor) -> dict:
        """One iteration of ``engine.train`` (forward, backward, and -- every ``accum_iter`` iterations -- gradient exchange,
        unscale, norm, AdamW, scaler update).  ``text``: (B, 77) token ids.  Returns loss / grad_norm / lr / skipped."""
        from .otter import create_text_mask
        ops, model = self.ops, self.model
        if self.iteration % self.accum_iter == 0:
            self.flat_g.zero_()
        mask = create_text_mask(text, model.sot_token, model.eot_token, model.first_k_tokens) if model.pool_true_text else None
        ids = text.detach().to("cpu", torch.int32).view(-1, text.shape[-1])
        device = self.flat_p.device
        last_micro = (self.iteration + 1) % self.accum_iter == 0
        ex = self.exchange if (last_micro and self.exchange.active) else None     # accumulating iterations exchange nothing
        if ex is not None:
            ex.begin({})
        loss = self.tail.forward_backward(images.to(device), ids, proprio, gt_actions, mask, self.grads,
                                          grad_scale=self.scaler.scale / self.accum_iter,
                                          **({"exchange": ex} if (ex is not None and self.overlap_exchange) else {}))
        out = {"lr": self.lr, "grad_norm": None, "skipped": False}
        if last_micro:
            if ex is not None:
                ex.finish()                                       # whatever the backward pass did not send (everything, without overlap)
                ex.wait()
            if self._stats_ws is None:
                self._stats_ws = torch.empty(ops._lib.load().otter_grad_stats_workspace_floats(), dtype=torch.float32, device=device)
            stats = ops.grad_stats(self.flat_g, self._stats_ws)
            inv = 1.0 / (self.scaler.scale * self.world)          # unscale_ and DistributedDataParallel's gradient mean
            self.opt_steps += 1
            for lo, hi, wd in self.spans:
                ops.adamw_step(self.flat_p[lo:hi], self.flat_g[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr=self.lr,
                               betas=self.betas, eps=self.eps, weight_decay=wd, step=self.opt_steps, inv_scale=inv, skip_flag=stats[1:2])
            sumsq, bad = stats.tolist()                           # the iteration's one device -> host read (with the loss)
            found_inf = bad != 0
            if found_inf:
                self.opt_steps -= 1                               # the kernels skipped the update
            out["grad_norm"] = float("inf") if found_inf else math.sqrt(sumsq) * inv
            out["skipped"] = found_inf
            self.scaler.update(found_inf)
            model.invalidate_device_weights(towers=False)         # AdamW wrote the flat buffer through raw pointers
        out["loss"] = float(loss.item())
        self.iteration += 1
        return out

    def state_dict(self) -> dict:
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "opt_steps": self.opt_steps, "iteration": self.iteration,
                "scaler": self.scaler.state_dict()}

    def load_state_dict(self, sd: dict) -> None:
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.opt_steps, self.iteration = int(sd["opt_steps"]), int(sd["iteration"])
        self.scaler.load_state_dict(sd["scaler"])
