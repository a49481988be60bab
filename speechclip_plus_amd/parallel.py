"""Data parallelism for the contrastive step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference trains with nn.DataParallel (``strategy: dp``): per step it broadcasts all weights,
scatters the batch, gathers the per-replica feature dicts to GPU 0 and computes the loss there on the full
batch (avssl/model/kwClip.py:149-189).  Here parameters stay resident on every rank and the step needs two
collectives only:

* ONE all-gather of a packed row per sample  [audio_feat (E) | image_feat (E) | id (int64 as 2 x f32)]
  (global 512 x (2*512+2) x 4 B ~= 2 MB: latency bound, so it is a single call), after which every rank
  evaluates the full (Bg x Bg) loss; autograd hands each rank the gradient of ITS rows only.
* ONE all-reduce (SUM - the loss is already the global-batch mean) of the flat gradient buffer
  (speechclip_plus_amd.optim.FlatAdam.flat_g), issued on a side stream so it overlaps the tail of the backward.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


class CommTimer:
    """Brackets around the step's collectives, for the N > 1 diagnostics of bench.py (``collectives`` field): HIP events recorded
    on the stream the collective is issued on (the all-gather on the main stream, the all-reduce on its side stream, the main
    stream's wait for the side stream at the join point), wall clock for CPU tensors (gloo rehearsal).  Off unless installed with
    ``set_comm_timer``: the timed region of the bench runs without it."""

    def __init__(self):
        self.spans = {}

    class _Span:
        def __init__(self, timer, name, cuda):
            self.timer, self.name, self.cuda = timer, name, cuda

        def __enter__(self):
            if self.cuda:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.e0.record()
            else:
                import time
                self.t0 = time.perf_counter()
            return self

        def __exit__(self, *exc):
            if self.cuda:
                self.e1.record()
                self.timer.spans.setdefault(self.name, []).append((self.e0, self.e1))
            else:
                import time
                self.timer.spans.setdefault(self.name, []).append(time.perf_counter() - self.t0)
            return False

    def span(self, name: str, cuda: bool):
        return CommTimer._Span(self, name, cuda)

    def summary(self, steps: int) -> dict:
        """microseconds per step and calls per step of every bracket (synchronises)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {}
        for name, spans in self.spans.items():
            us = sum(s * 1e6 if isinstance(s, float) else s[0].elapsed_time(s[1]) * 1e3 for s in spans)
            out[name] = {"us_per_step": round(us / max(steps, 1), 1), "calls_per_step": round(len(spans) / max(steps, 1), 2)}
        return out


_comm_timer: Optional["CommTimer"] = None


def set_comm_timer(t: Optional["CommTimer"]) -> None:
    global _comm_timer
    _comm_timer = t


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def comm_span(name: str, cuda: bool):
    return _comm_timer.span(name, cuda) if _comm_timer is not None else _NoSpan()


def dp_world(group: Optional[dist.ProcessGroup] = None) -> int:
    """World size as the data-parallel code sees it: 1 without an initialised process group.  With SC_FORCE_COLLECTIVES=1 a
    one-rank group reports 2 to the callers' "is there anything to exchange" checks, so every collective of the step is
    issued through the backend (RCCL rehearsal on a one-GPU box; the results are unchanged by one-rank collectives)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    w = dist.get_world_size(group)
    if w == 1 and os.environ.get("SC_FORCE_COLLECTIVES", "0") == "1":
        return 2
    return w


class _AllGatherRows(torch.autograd.Function):
    """all_gather along dim 0 (equal per-rank row counts); backward = this rank's slice of the gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.n = x.shape[0]
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        with comm_span("all_gather", x.is_cuda):
            dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n: (ctx.rank + 1) * ctx.n], None


def gather_loss_feats(audio_feat, image_feat: torch.Tensor, ids: torch.Tensor,
                      group: Optional[dist.ProcessGroup] = None):
    """Pack, all-gather once, unpack -> (audio_all, image_all, ids_all) with autograd through audio/image.
    ``audio_feat`` may be one (B, E) tensor or a list of them (parallel + cascaded embeddings of the hybrid recipes):
    all of them travel in the same packed row, still ONE collective."""
    single = isinstance(audio_feat, torch.Tensor)
    feats = [audio_feat] if single else list(audio_feat)
    if dp_world(group) == 1:
        return audio_feat, image_feat, ids
    B, E = image_feat.shape
    id_bits = ids.to(torch.int64).contiguous().view(torch.float32).view(B, 2)
    packed = torch.cat([f.float() for f in feats] + [image_feat.float(), id_bits], dim=1)   # (B, (n+1)E + 2) fp32
    allp = _AllGatherRows.apply(packed, group)
    n = len(feats)
    ids_all = allp[:, (n + 1) * E:].detach().contiguous().view(torch.int64).view(-1)
    outs = [allp[:, i * E: (i + 1) * E] for i in range(n)]
    return (outs[0] if single else outs), allp[:, n * E: (n + 1) * E], ids_all


def scale_replicated_grads(params, group: Optional[dist.ProcessGroup] = None) -> None:
    """Parameters of the loss itself (the trainable temperature) see the WHOLE global-batch loss on every rank, so each rank
    already holds their full gradient: divide by the world size before the SUM all-reduce (every other parameter only
    receives the gradient that flows through this rank's own rows, whose sum over ranks is the global gradient)."""
    if dp_world(group) == 1:
        return
    world = dist.get_world_size(group)
    for p in params:
        if p.grad is not None:
            p.grad.mul_(1.0 / world)


class GradAllReduce:
    """Sum the flat gradient buffer across ranks on a side HIP stream."""

    def __init__(self, flat_grad: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
        self.flat_grad, self.group = flat_grad, group
        if flat_grad.is_cuda:
            from .ops import shared_stream
            self.stream = shared_stream("allreduce", flat_grad.device)
        else:
            self.stream = None
        self.work = None

    def launch(self) -> None:
        if dp_world(self.group) == 1:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            # synchronous form ON the side stream: with the NCCL / RCCL backend the collective runs on the process group's own
            # stream and a blocking call makes the CURRENT (= side) stream wait for it - so the closing event of the bracket sits
            # behind the collective and `all_reduce_us` measures it, not just its launch; the host does not block
            with torch.cuda.stream(self.stream), comm_span("all_reduce", True):
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.work = None
        else:
            with comm_span("all_reduce", False):
                self.work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if not self.flat_grad.is_cuda:
                    self.work.wait()             # CPU tensors (gloo rehearsal): the bracket is wall clock, so it must cover the wait
                    self.work = None

    def wait(self) -> None:
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class AccumulationSchedule:
    """``trainer.accumulate_grad_batches`` (config/speechCLIP+/model_large/coco/spchclip_h+.yaml:138; Lightning semantics): every
    micro-step back-propagates loss / n into the SAME gradient buffer; the gradient collective, the clip + optimiser step, the
    learning-rate schedule and ``global_step`` advance only on every n-th micro-step (the boundary).  No collective is issued on
    the micro-steps in between - their gradients are partial sums that only this rank needs."""

    def __init__(self, n: int = 1):
        self.n = max(1, int(n))
        self.micro = 0                       # micro-steps taken since the last boundary

    @property
    def loss_scale(self) -> float:
        return 1.0 / self.n

    def advance(self) -> bool:
        """count one micro-step -> True when it is a boundary (collectives + optimiser run now)"""
        self.micro += 1
        if self.micro >= self.n:
            self.micro = 0
            return True
        return False
