"""WeightedSumLayer (avssl/module/weighted_sum.py:9-45) over the HIP weighted-sum kernel.

Two entry points:
* ``forward(list_of_tensors)``  - the reference signature; stacks into the kernel layout.
* ``forward_padded(hidden, B, R, T, D)`` - zero-copy path used by the speech encoder: ``hidden`` is the
  resident [NL, B*R, D] bf16 buffer the transformer layers wrote; the sum lands in a fresh [B, R, D] buffer
  at row offset 1 (row 0 is left for the CLS token of the parallel branch) and the returned ``feat`` view
  carries a handle so the branch can run its pooled backward straight into d(weights) without
  materialising a bf16 gradient of the features.
"""
from typing import List, Optional

import torch
from torch import nn

from . import ops


class PaddedFeatHandle:
    """Side-channel from the encoder to the branch head (same process, same step)."""

    def __init__(self, src, hidden, ws_layer, w_soft, B, R, T, D, normalize=False, plan=None):
        self.src, self.hidden, self.ws_layer, self.w_soft = src, hidden, ws_layer, w_soft
        self.normalize = normalize
        self.B, self.R, self.T, self.D = B, R, T, D
        self.layers_bwd = None       # set by the encoder when transformer layers are unfrozen: callable(dX, w_soft)
        # ``hidden`` is the encoder plan's RESIDENT workspace: the next forward with the same (B, L) geometry overwrites it through
        # raw-pointer kernels (no autograd version bump).  One outstanding forward per plan: a backward that arrives after the
        # plan has been re-used must fail loudly instead of differentiating against the wrong states.
        self.plan, self.generation = plan, (plan.generation if plan is not None else None)
        self.lazy = getattr(plan, "lazy", None)      # ops.LazyStates: the hidden states are raw rows + row statistics
        self.seg = getattr(plan, "seg", None)        # ops.RowSegments: ``hidden`` is in the ragged row layout, ``src`` uniform [B, R, D]
        # the attention block of the cascaded+/hybrid+ branches may read ``src`` in place (mha_block.resident_rows): pitch a multiple
        # of 64, every row finite, and a few zero rows allocated behind the buffer (the cascaded layout starts one row in)
        self.inplace_ok = bool(getattr(plan, "branch_rows", 0)) and R % 64 == 0 and src.dtype == torch.bfloat16

    def release(self) -> None:
        """the backward has enqueued its last read of the plan's resident buffers (speech_encoder._Plan.release)"""
        if self.plan is not None:
            self.plan.release()

    def check_fresh(self) -> None:
        if self.plan is not None and self.plan.generation != self.generation:
            raise RuntimeError("the encoder ran another forward with the same batch geometry before this backward: its hidden "
                               "states (a resident workspace) have been overwritten.  Run backward before the next forward, or "
                               "clone what must outlive it (one outstanding forward per (B, L) plan; INTEGRATION.md)")


class _WeightedSumFn(torch.autograd.Function):
    """Generic autograd path: out[M, D] = sum_n softmax(w)_n h[n]; grads to w only (h is a constant)."""

    @staticmethod
    def forward(ctx, weights, h, B, R, D, normalize):
        w_soft = torch.softmax(weights.float(), dim=0).contiguous()
        out = torch.empty(B, R, D, device=h.device, dtype=torch.bfloat16)      # every row is written (row offset 0)
        ops.wsum_fwd(h, w_soft, out, B, R, D, 0, normalize)
        ctx.save_for_backward(h, w_soft)
        ctx.dims = (B, R, D, normalize)
        return out

    @staticmethod
    def backward(ctx, g):
        h, w_soft = ctx.saved_tensors
        B, R, D, normalize = ctx.dims
        return ops.wsum_bwd_logits(h, g.float().contiguous(), w_soft, B, R, D, 0, normalize=normalize), None, None, None, None, None


class _WeightedSumSrcFn(torch.autograd.Function):
    """Encoder fast path with autograd: the sum lands at row offset 1 of a fresh [B, R, D] buffer.  The parallel
    branch never differentiates through this node (it gets d(weights) from the pooled backward via the handle); the
    cascaded+/hybrid+ branches consume ``feat`` with ordinary torch ops, and their gradient arrives here."""

    @staticmethod
    def forward(ctx, weights, hidden, B, R, D, normalize, plan, w_soft=None):
        ctx.plan, ctx.generation = plan, (plan.generation if plan is not None else None)
        if w_soft is None:                   # (forward_padded hands its own softmax of the same weights over)
            w_soft = torch.softmax(weights.detach().float(), dim=0).contiguous()
        extra = int(getattr(plan, "branch_rows", 0))       # zero rows behind the buffer for a consumer that reads it in place, shifted
        if extra:
            flat = torch.empty((B * R + extra) * D, device=hidden.device, dtype=torch.bfloat16)
            flat[B * R * D:].zero_()
            # (a tensor of its own over the front of that storage - not a view, which autograd would refuse to see modified in place
            # when the hybrid branch writes its CLS token into row 0)
            src = torch.empty(0, device=hidden.device, dtype=torch.bfloat16).set_(flat.untyped_storage(), 0, (B, R, D), (R * D, D, 1))
        else:
            src = torch.empty(B, R, D, device=hidden.device, dtype=torch.bfloat16)
        ctx.lazy = getattr(plan, "lazy", None)
        ctx.seg = getattr(plan, "seg", None)
        if ctx.seg is None:
            src[:, 0].zero_()                # rows 1 .. R - 1 are written by the kernel; row 0 is the CLS slot
        # (segment layout: the kernel writes EVERY row of src - frames at row t + 1, zeros in the CLS slot and behind an utterance)
        ops.wsum_fwd(hidden, w_soft, src, B, R, D, 1, normalize, lazy=ctx.lazy, seg=ctx.seg)
        ctx.save_for_backward(hidden, w_soft)
        ctx.dims = (B, R, D, normalize)
        return src

    @staticmethod
    def backward(ctx, g):
        hidden, w_soft = ctx.saved_tensors
        B, R, D, normalize = ctx.dims
        if ctx.plan is not None and ctx.plan.generation != ctx.generation:
            raise RuntimeError("the encoder ran another forward with the same batch geometry before this backward (its resident "
                               "hidden states were overwritten): one outstanding forward per (B, L) plan")
        if not (g.is_contiguous() and g.dtype in (torch.float32, torch.bfloat16)):      # (bf16 rows as the attention block returns them)
            g = g.float().contiguous()
        d_w = ops.wsum_bwd_logits(hidden, g, w_soft, B, R, D, 1, normalize=normalize, lazy=ctx.lazy, seg=ctx.seg)
        if ctx.plan is not None:
            ctx.plan.release()               # last read of the plan's resident states by this backward (speech_encoder._claim)
        return (d_w, None, None, None, None, None, None, None)


class WeightedSumLayer(nn.Module):
    def __init__(self, n_weights: int, normalize_features: bool = False):
        super().__init__()
        self.n_weights = n_weights
        self.weights = nn.Parameter(torch.zeros((n_weights,), dtype=torch.float))
        self.normalize_features = normalize_features      # per-layer non-affine layer_norm (HuBERT-large recipes)

    def forward(self, x: List[torch.Tensor]) -> torch.Tensor:
        assert len(x) == self.n_weights, len(x)
        shape = x[0].shape
        D = shape[-1]
        h = torch.stack([t.reshape(-1, D).to(torch.bfloat16) for t in x], dim=0).contiguous()
        out = _WeightedSumFn.apply(self.weights, h, 1, h.shape[1], D, self.normalize_features)
        return out.view(*shape)

    def forward_padded(self, hidden: torch.Tensor, B: int, R: int, T: int, D: int, plan=None) -> torch.Tensor:
        w_soft = torch.softmax(self.weights.detach().float(), dim=0).contiguous()
        src = _WeightedSumSrcFn.apply(self.weights, hidden, B, R, D, self.normalize_features, plan, w_soft)
        feat = src[:, 1: T + 1]
        feat._sc_handle = PaddedFeatHandle(src, hidden, self, w_soft, B, R, T, D, self.normalize_features, plan)
        return feat
