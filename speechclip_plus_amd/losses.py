"""MaskedContrastiveLoss (avssl/module/losses.py:129-245) on the HIP loss kernels.

Same constructor (temperature, temperature_trainable, margin, dcl, a2b, b2a - every option built), ``forward(feat_A, feat_B,
index=None) -> scalar``, ``current_temperature`` and the ``temperature`` parameter / attribute (log(1/T) when trainable, 1/T
otherwise).  No MAX_EYE = 256 cap (losses.py:126: the reference cannot run B > 256; the masks are computed in-kernel from
``index``).  Forward = one launch (csrc/loss_optim.hip: sc_infonce_fwd); the temperature reaches the kernels as a DEVICE scalar,
so a trainable temperature costs no host synchronisation in the step.
"""
import numpy as np
import torch
from torch import nn

from . import ops


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_A, feat_B, inv_temp, index, margin, dcl, a2b, b2a):
        A = feat_A.detach().float().contiguous()
        Bm = feat_B.detach().float().contiguous()
        it = inv_temp.detach().float().reshape(1).contiguous()
        ids = index.contiguous() if index is not None else None
        loss, logits, lse_row, lse_col = ops.infonce_fwd(A, Bm, ids, it, margin, dcl, a2b, b2a)
        ctx.save_for_backward(A, Bm, logits, lse_row, lse_col, it, ids if ids is not None else torch.empty(0))
        ctx.has_ids = ids is not None
        ctx.opts = (margin, dcl, a2b, b2a)
        return loss.reshape(())            # (a fresh [1] buffer of this call: no copy)

    @staticmethod
    def backward(ctx, g):
        A, Bm, logits, lse_row, lse_col, it, ids = ctx.saved_tensors
        ids = ids if ctx.has_ids else None
        G, dot = ops.infonce_grad(logits, ids, lse_row, lse_col, g.float().reshape(1).contiguous(), it, *ctx.opts)
        dA = dB = dT = None
        if ctx.needs_input_grad[0]:
            dA = ops.sgemm_mfma(G, Bm, b_kmajor=True)                          # G . B     (G already carries inv_temp)
        if ctx.needs_input_grad[1]:
            dB = ops.sgemm_mfma(G, A, a_kmajor=True, b_kmajor=True)            # G^T . A
        if ctx.needs_input_grad[2]:
            dT = dot.sum()                                                     # d loss / d inv_temp (device scalar)
        return dA, dB, dT, None, None, None, None, None


class MaskedContrastiveLoss(nn.Module):
    def __init__(self, temperature: float = 0.07, temperature_trainable: bool = False, margin: float = 0.0,
                 dcl: bool = False, a2b: bool = True, b2a: bool = True):
        super().__init__()
        assert a2b or b2a, "Cannot set both `a2b` and `b2a` to False."
        self.temperature_trainable = temperature_trainable
        self.margin, self.dcl, self.a2b, self.b2a = float(margin), bool(dcl), bool(a2b), bool(b2a)
        if temperature_trainable:
            self.temperature = nn.Parameter(torch.ones([]) * np.log(1 / temperature))
        else:
            self.temperature = 1 / temperature
        self._inv_temp_dev = {}

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
        if self.temperature_trainable:
            inv_temp = torch.exp(self.temperature)
        else:                                              # constant: one device scalar per device, built once
            inv_temp = self._inv_temp_dev.get(feat_A.device)
            if inv_temp is None:
                inv_temp = self._inv_temp_dev[feat_A.device] = torch.full((1,), float(self.temperature), device=feat_A.device)
        # losses.py:219-220 subtracts the margin only ``if self.margin > 0.0``: a negative margin is the plain loss
        return _InfoNCEFn.apply(feat_A, feat_B, inv_temp, index, max(self.margin, 0.0), self.dcl, self.a2b, self.b2a)
