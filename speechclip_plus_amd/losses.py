"""MaskedContrastiveLoss (avssl/module/losses.py:129-245) on the HIP loss kernels.

Same constructor, ``forward(feat_A, feat_B, index=None) -> scalar``, ``current_temperature`` and the
``temperature`` parameter/attribute (log(1/T) when trainable, 1/T otherwise).  No MAX_EYE = 256 cap
(losses.py:126: the reference cannot run B > 256; the masks are computed in-kernel from ``index``).
``margin`` and ``dcl`` are accepted only at their shipped values (0.0 / False).
"""
import numpy as np
import torch
from torch import nn

from . import ops


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_A, feat_B, inv_temp, index):
        A = feat_A.detach().float().contiguous()
        Bm = feat_B.detach().float().contiguous()
        Bg, E = A.shape
        it = float(inv_temp.detach()) if isinstance(inv_temp, torch.Tensor) else float(inv_temp)
        logits = ops.sgemm(A, E, 1, Bm, E, 1, Bg, Bg, E, alpha=it)
        ids = index.contiguous() if index is not None else None
        loss, lse_row, lse_col = ops.infonce_lse(logits, ids)
        ctx.save_for_backward(A, Bm, logits, lse_row, lse_col, ids if ids is not None else torch.empty(0))
        ctx.has_ids = ids is not None
        ctx.it = it
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        A, Bm, logits, lse_row, lse_col, ids = ctx.saved_tensors
        ids = ids if ctx.has_ids else None
        Bg, E = A.shape
        G, dot = ops.infonce_grad(logits, ids, lse_row, lse_col, g.float().reshape(1).contiguous())
        dA = dB = dT = None
        if ctx.needs_input_grad[0]:
            dA = ops.sgemm(G, Bg, 1, Bm, 1, E, Bg, E, Bg, alpha=ctx.it)      # G . B
        if ctx.needs_input_grad[1]:
            dB = ops.sgemm(G, 1, Bg, A, 1, E, Bg, E, Bg, alpha=ctx.it)       # G^T . A
        if ctx.needs_input_grad[2]:
            dT = dot.sum() / ctx.it                                          # d loss / d inv_temp
        return dA, dB, dT, None


class MaskedContrastiveLoss(nn.Module):
    def __init__(self, temperature: float = 0.07, temperature_trainable: bool = False, margin: float = 0.0,
                 dcl: bool = False, a2b: bool = True, b2a: bool = True):
        super().__init__()
        assert a2b or b2a, "Cannot set both `a2b` and `b2a` to False."
        if margin != 0.0 or dcl or not (a2b and b2a):
            raise NotImplementedError("only the shipped loss configuration (margin 0, dcl false, a2b & b2a) is built")
        self.temperature_trainable = temperature_trainable
        self.margin, self.dcl, self.a2b, self.b2a = margin, dcl, a2b, b2a
        if temperature_trainable:
            self.temperature = nn.Parameter(torch.ones([]) * np.log(1 / temperature))
        else:
            self.temperature = 1 / temperature

    @property
    def current_temperature(self) -> float:
        if self.temperature_trainable:
            temp = self.temperature.data.cpu().detach().float().exp().item()
        else:
            temp = self.temperature
        return float(temp)

    @property
    def temperature_for_logging(self):
        """``current_temperature`` without the device -> host synchronisation of ``.item()``: a 0-dim device tensor when the
        temperature is trainable (loggers accept tensors), the python float otherwise."""
        if self.temperature_trainable:
            return self.temperature.detach().float().exp()
        return float(self.temperature)

    def forward(self, feat_A: torch.Tensor, feat_B: torch.Tensor, index: torch.LongTensor = None) -> torch.Tensor:
        assert feat_A.shape == feat_B.shape, (feat_A.shape, feat_B.shape)
        if index is not None:
            assert index.shape[0] == feat_A.shape[0], (index.shape, feat_A.shape)
        inv_temp = torch.exp(self.temperature) if self.temperature_trainable else self.temperature
        return _InfoNCEFn.apply(feat_A, feat_B, inv_temp, index)
