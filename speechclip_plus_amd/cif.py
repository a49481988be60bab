"""Continuous integrate-and-fire downsampler: mirror of avssl/module/cif.py:24-311 (cascaded+/hybrid+ branches).

On a GPU the accumulation itself runs on the library's kernels (csrc/cif.hip through ``CifFireFn``: one deterministic pass over
the frames instead of the reference's atomically-accumulating scatter_add_ calls, forward and backward); the slot boundaries come
from torch.cumsum either way, and the tiny (B, S) / (B, T) bookkeeping around it stays device-side torch.  Same constructor keywords, sub-module names (``conv.0``,
``weight_proj.1``) and result-dict keys as the reference, including its quirks that change numbers:
``nn.Dropout()`` (p = 0.5) in the weight generator, ``MAX_FEAT_LEN = 75``, alpha clipped to [0, 1], the
quantity output taken BEFORE scaling, train-time tail drop vs inference-time tail firing.
"""
import logging
from typing import Optional

import torch
from torch import nn

logger = logging.getLogger(__name__)

MAX_FEAT_LEN = 75   # cif.py:11


def _length_mask(max_length: int, lens: torch.Tensor) -> torch.Tensor:
    """True = padding (cif.py:14-21), built on the lengths' device."""
    return torch.arange(max_length, device=lens.device).unsqueeze(0) >= lens.unsqueeze(1)


class CifFireFn(torch.autograd.Function):
    """out[B, T + 1, C] = integrate-and-fire accumulation of x under (alpha, csum); the slot indices are constants of the graph
    (computed under no_grad in the reference too), gradients flow to x, alpha and csum."""

    @staticmethod
    def forward(ctx, x, alpha, csum, T, thr):
        from . import ops
        xf, af, cf = x.detach().float().contiguous(), alpha.detach().float().contiguous(), csum.detach().float().contiguous()
        ctx.save_for_backward(xf, af, cf)
        ctx.meta = (int(T), float(thr), x.dtype, alpha.dtype)
        return ops.cif_fwd(xf, af, cf, int(T), float(thr)).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        xf, af, cf = ctx.saved_tensors
        T, thr, xdt, adt = ctx.meta
        dx, da, dc = ops.cif_bwd(xf, af, cf, g.float().contiguous(), T, thr)
        return dx.to(xdt), da.to(adt), dc.to(adt), None, None


class CIF(nn.Module):
    def __init__(self, cif_threshold=1.0, cif_output_dim=768, encoder_embed_dim=768, produce_weight_type="conv",
                 num_layer=1, conv_cif_width=3, conv_cif_dropout=0.1, apply_scaling=True, apply_tail_handling=True,
                 tail_handling_firing_threshold=0.5, scaling_step=-1, **config):
        super().__init__()
        self.cif_threshold = cif_threshold
        self.cif_output_dim = cif_output_dim
        self.encoder_embed_dim = encoder_embed_dim
        self.produce_weight_type = produce_weight_type
        self.conv_cif_width = conv_cif_width
        self.conv_cif_dropout = conv_cif_dropout
        self.apply_scaling = apply_scaling
        self.apply_tail_handling = apply_tail_handling
        self.tail_handling_firing_threshold = tail_handling_firing_threshold
        self.scaling_step = scaling_step
        self.num_layer = num_layer
        if produce_weight_type != "conv":
            raise NotImplementedError("only produce_weight_type='conv' is runnable in the reference (cif.py:116-135)")
        if cif_output_dim != encoder_embed_dim:
            raise NotImplementedError("cif_output_dim != encoder_embed_dim is not used by any shipped config")
        layers = []
        for _ in range(num_layer):
            layers += [nn.Conv1d(encoder_embed_dim, encoder_embed_dim, conv_cif_width, stride=1,
                                 padding=int(conv_cif_width / 2)), nn.Dropout(), nn.ReLU()]
        self.conv = nn.Sequential(*layers)
        self.weight_proj = nn.Sequential(nn.Dropout(), nn.Linear(encoder_embed_dim, 1), nn.Sigmoid())

    def forward(self, input_dict, target_lengths=None, eps=1e-5):
        feats = input_dict["audio_feat"]                       # B x T x D
        pad = input_dict["audio_feat_pad_mask"].bool()         # B x T, True = padding
        original_length = (~pad).sum(-1).long()
        if self.scaling_step >= 0 and self.apply_scaling and input_dict["global_step"] >= self.scaling_step:
            self.apply_scaling = False                         # cif.py:110-112: permanent once the step is reached
        logits = self._weight_conv(feats)
        alpha = self.weight_proj(logits).clip(min=0.0, max=1.0).float().squeeze(-1)
        alpha = alpha.masked_fill(pad, 0.0)
        orig_alpha = alpha
        alpha_sum = alpha.sum(1)
        assert (alpha_sum > 0).any(), f"alphas are all zero:\n{alpha_sum}"
        if self.apply_scaling and target_lengths is not None:
            desired = self.cif_threshold * target_lengths.type_as(alpha) + eps
            alpha = alpha * (desired / alpha_sum).unsqueeze(1)
        out = {"quantity_out": alpha_sum, "orig_alpha": orig_alpha, "original_length": original_length,
               "target_len": target_lengths}
        out.update(self.integrate_and_fire(feats, alpha, target_lengths=target_lengths))
        out["input_feats_pad_mask"] = pad
        return out

    def _weight_conv(self, feats: torch.Tensor) -> torch.Tensor:
        """``self.conv(feats^T)^T`` (Conv1d k, stride 1, 'same' padding -> Dropout -> ReLU per layer) evaluated channels-last as
        one GEMM per layer over the k shifted copies of the input: MIOpen's fp32 NCHW path for this 768 x 768 x 3 conv costs
        ~7 ms per call at B = 64 (rocprofv3), the GEMM form ~1 ms; same parameters, same arithmetic (fp32)."""
        x = feats
        for i in range(0, len(self.conv), 3):
            conv, drop, act = self.conv[i], self.conv[i + 1], self.conv[i + 2]
            k, pad = conv.kernel_size[0], conv.padding[0]
            B, T, C = x.shape
            xp = torch.nn.functional.pad(x, (0, 0, pad, pad))                       # (B, T + 2 pad, C)
            cols = torch.cat([xp[:, j: j + T + 2 * pad - k + 1] for j in range(k)], dim=-1)   # (B, T', k C), tap-major
            w = conv.weight.permute(0, 2, 1).reshape(conv.out_channels, k * C)    # [C_out, k, C_in] flattened tap-major
            if self.training and cols.is_cuda:
                # training: bf16 operands / fp32 accumulation on the library's GEMM with its dgrad + weight-gradient products
                # (the reference trains under precision-16 autocast; the keyword COUNT is pinned by the target-length scaling
                # of alpha, so the discrete part of CIF does not depend on this rounding).  Inference stays fp32: there the
                # count is floor(sum alpha), which must be the fp32 oracle's.
                from .linear_fn import linear_bf16_autograd
                y = linear_bf16_autograd(cols, w, conv.bias)
            else:
                y = torch.nn.functional.linear(cols, w, conv.bias)
            x = act(drop(y))
        return x

    def integrate_and_fire(self, input: torch.Tensor, alpha: torch.Tensor,
                           target_lengths: Optional[torch.Tensor] = None) -> dict:
        """cif.py:157-311.  Frame s with cumulative weight c_s contributes to output slots floor(c_{s-1}/thr) ..
        floor(c_s/thr): the part up to the first boundary goes left, whole thresholds go to the slots in between, the
        remainder goes right."""
        B, S, C = input.shape
        thr = self.cif_threshold
        assert tuple(alpha.shape) == (B, S), f"{alpha.shape} != {(B, S)}"
        feat_lengths = (alpha.sum(1) / thr).floor().clip(min=1, max=MAX_FEAT_LEN).long()
        T = int(feat_lengths.max())
        csum = alpha.cumsum(-1)
        with torch.no_grad():
            right_idx = (csum / thr).floor().long().clip(min=0, max=T)
            left_idx = right_idx.roll(1, dims=1)
            left_idx[:, 0] = 0
            fire_num = right_idx - left_idx
            extra = (fire_num - 1).clip(min=0)
        fire_mask = fire_num > 0
        zero = alpha.new_zeros((1,))
        right_w = torch.where(fire_mask, csum - right_idx.type_as(alpha) * thr, zero).type_as(input)
        left_w = (alpha - right_w - extra.type_as(alpha) * thr).type_as(input)
        if input.is_cuda and C % 4 == 0 and S <= 2048:
            output = CifFireFn.apply(input, alpha, csum, T, thr)   # slot T collects the tail
        else:
            output = input.new_zeros((B, T + 1, C))
            output.scatter_add_(1, right_idx.unsqueeze(-1).expand(-1, -1, C), right_w.unsqueeze(-1) * input)
            output.scatter_add_(1, left_idx.unsqueeze(-1).expand(-1, -1, C), left_w.unsqueeze(-1) * input)
            if extra.ge(0).any():
                steps = int(extra.max())
                tgt = left_idx
                whole = input * thr
                for _ in range(steps):
                    tgt = (tgt + 1).clip(max=T)
                    output.scatter_add_(1, tgt.unsqueeze(-1).expand(-1, -1, C), whole * (extra > 0).unsqueeze(2))
                    extra = extra - 1
        if self.apply_tail_handling:
            if target_lengths is not None:
                output = output[:, :T, :]                      # training: the tail is dropped
            else:
                zero = right_w.new_zeros((1,))
                tail_w = torch.where(right_idx == feat_lengths.unsqueeze(1), right_w, zero).sum(-1)
                tail_w = tail_w + torch.where(left_idx == feat_lengths.unsqueeze(1), left_w, zero).sum(-1)
                extend = tail_w >= self.tail_handling_firing_threshold
                if extend.any():
                    factor = (thr / tail_w.masked_fill(~extend, thr)).view(B, 1, 1).expand(-1, -1, C).to(output.dtype)
                    upscale = torch.ones_like(output).scatter(1, feat_lengths.view(B, 1, 1).expand(-1, -1, C), factor).detach()
                    output = output * upscale
                    feat_lengths = feat_lengths + extend.long()
                    cols = feat_lengths - 1                    # cif.py:281-283 (diagnostic mask only)
                    fire_mask[:, cols] = fire_mask[:, cols] + extend
                    feat_lengths = feat_lengths.clip(max=MAX_FEAT_LEN)
                    T = int(feat_lengths.max())
                output = output[:, :T, :]
                tail_mask = torch.arange(T, device=output.device).unsqueeze(0) >= feat_lengths.unsqueeze(1)
                output = output.masked_fill(tail_mask.unsqueeze(-1), 0)
        else:
            output = output[:, :T, :]
        return {"dsample_feats_pad_mask": _length_mask(output.shape[1], feat_lengths), "dsample_feats": output,
                "dsample_feats_length": feat_lengths, "alpha": alpha, "fired_marks": fire_mask}
