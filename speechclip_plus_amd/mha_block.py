"""``LayerNorm(MultiheadAttention(x, x, x, key_padding_mask) + x)`` - the attention block of the cascaded+/hybrid+ branches
(avssl/module/kw_modules/TransformerModels.py:101-126, one head of 768 in the base recipes, 8 heads of 128 in the large ones) -
forward and backward on the library's kernels, any head_dim that is a multiple of 64 (scope row f3).

The flash kernels of the encoder are head_dim 64; here the S x S core runs as batched bf16 GEMMs on ``sc_gemm_bf16`` (batch
dimensions = utterance x head, operands addressed in place inside the fused projection output):

    layout    every utterance is padded to Sp = roundup(S, 64) rows: x_b [B Sp, D] bf16 (zero pad rows)
    forward   qkv = x_b Wi^T + bi                                   [B Sp, 3D]
              scores[b,h] = q_h k_h^T                                fp32 [B, H, Sp, Sp]           (K = head_dim)
              P = dropout(softmax(scale * scores | key mask))        bf16, pad keys exactly 0   (csrc/softmax.hip: one pass,
                                                                     dropout = the stateless hash mask, regenerated in the backward)
              ctx[b,h] = P V_h   (W operand = V^T, ONE 2-D transpose of the V columns serves all utterances: ldw = B Sp)
              out = LayerNorm(ctx Wo^T + bo + x_b)                   (bias + residual in the GEMM epilogue)
    backward  LayerNorm' -> out_proj dgrad / wgrad -> dP = dctx V_h^T -> dS = P (dP - rowsum(P dP)) scale
              -> dV = P^T dctx, dQ = dS K, dK = dS^T Q   (transposed operands: 2-D transposes of P, dS, dctx, q, k)
              -> in_proj dgrad (+ residual gradient) / wgrad
Weight gradients: ops.wgrad_bf16 (bf16 transposes + split-K batched GEMM into fp32 partials).  The reference trains this block
under precision-16 autocast; here operands are bf16 with fp32 accumulation and fp32 softmax / LayerNorm statistics.
"""
from typing import Optional

import torch

from . import ops


def _roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


_calls = 0


def _next_seed() -> int:
    """Seed of one train-mode call: torch's seed, the rank and a call counter (the mask itself is a stateless hash of
    (element, seed) evaluated inside the softmax kernels, forward and backward - nothing is stored)."""
    global _calls
    from .speech_encoder import _mix32
    _calls += 1
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    return _mix32(_mix32(torch.initial_seed() & 0xffffffff) ^ _mix32(0x51ED270B * (rank + 1)) ^ (_calls * 0x9E3779B9))


class MhaNormFn(torch.autograd.Function):
    """inputs: x [B, S, D], in_proj_weight [3D, D], in_proj_bias [3D], out_proj.weight [D, D], out_proj.bias [D],
    LayerNorm weight / bias [D]; constants: key_padding_mask [B, S] bool (True = padding), H, eps, p_drop (0 in eval)."""

    @staticmethod
    def forward(ctx, x, Wi, bi, Wo, bo, g, beta, kpm, H, eps, p_drop, seed):
        B, S, D = x.shape
        dh = D // H
        assert D % H == 0 and dh % 64 == 0 and D % 64 == 0, "head_dim must be a multiple of 64"
        dev, bf = x.device, torch.bfloat16
        Sp = _roundup(S, 64)
        M = B * Sp
        xb = torch.zeros(B, Sp, D, device=dev, dtype=bf)
        xb[:, :S] = x.detach()
        xb = xb.view(M, D)
        Wi_b, Wo_b = Wi.detach().to(bf).contiguous(), Wo.detach().to(bf).contiguous()
        qkv = ops.linear_bf16(xb, Wi_b, bi.detach().float().contiguous())                    # [M, 3D]
        # ---- S x S core
        scores = torch.empty(B, H, Sp, Sp, device=dev, dtype=torch.float32)
        ops.gemm_raw(qkv, 3 * D, qkv[:, D:], 3 * D, scores, Sp, Sp, Sp, dh, out_f32=True, nb1=B, nb2=H,
                     sA=(Sp * 3 * D, dh), sW=(Sp * 3 * D, dh), sC=(H * Sp * Sp, Sp * Sp))
        key_pad = torch.ones(B, Sp, device=dev, dtype=torch.uint8)
        key_pad[:, :S] = kpm
        P, Pd = ops.softmax_fwd(scores, key_pad, H * Sp, dh ** -0.5, p_drop, seed)               # P un-dropped (softmax backward), Pd dropped
        del scores
        vT = ops.transpose_bf16(qkv[:, 2 * D:])                                               # [D, M]: V^T of every utterance
        cx = torch.empty(M, D, device=dev, dtype=bf)
        ops.gemm_raw(Pd, Sp, vT, M, cx, D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp * Sp, Sp * Sp), sW=(Sp, dh * M), sC=(Sp * D, dh))
        pre = ops.linear_bf16(cx, Wo_b, bo.detach().float().contiguous(), residual=xb)
        # fresh copies: trainable parameters are views into the optimiser's flat buffer (4-byte aligned), the row kernels read
        # gamma / beta with 16-byte loads
        g32, b32 = g.detach().float().clone(), beta.detach().float().clone()
        out = ops.layernorm_bf16(pre, g32, b32, eps=eps)
        ctx.save_for_backward(xb, Wi_b, Wo_b, qkv, P, Pd if p_drop > 0.0 else None, cx, pre, g32)
        ctx.meta = (B, S, Sp, D, H, dh, eps, p_drop, seed, x.dtype)
        return out.view(B, Sp, D)[:, :S].to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        xb, Wi_b, Wo_b, qkv, P, Pd, cx, pre, g = ctx.saved_tensors
        B, S, Sp, D, H, dh, eps, p_drop, seed, xdtype = ctx.meta
        dev, bf = dout.device, torch.bfloat16
        M = B * Sp
        if Pd is None:
            Pd = P
        dy = torch.zeros(B, Sp, D, device=dev, dtype=bf)
        dy[:, :S] = dout
        dy = dy.view(M, D)
        # ---- LayerNorm, out_proj
        dpre, dg, dbeta = ops.layernorm_bwd(pre, dy, g, eps, want_param_grads=True)
        gWo = torch.empty(D, D, device=dev, dtype=torch.float32)
        gbo = torch.empty(D, device=dev, dtype=torch.float32)
        ops.wgrad_bf16(dpre, cx, gWo, gbo, beta=0.0)
        dcx = ops.linear_bf16(dpre, Wo_b.t().contiguous())                                    # [M, D]
        # ---- core: dP = dctx V^T ; dS = P (dP - rowsum(P dP)) scale
        dP = torch.empty(B, H, Sp, Sp, device=dev, dtype=torch.float32)
        ops.gemm_raw(dcx, D, qkv[:, 2 * D:], 3 * D, dP, Sp, Sp, Sp, dh, out_f32=True, nb1=B, nb2=H,
                     sA=(Sp * D, dh), sW=(Sp * 3 * D, dh), sC=(H * Sp * Sp, Sp * Sp))
        dS = ops.softmax_bwd(dP, P, dh ** -0.5, p_drop, seed)
        del dP
        dqkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
        # dV = Pd^T dctx   (A = Pd^T from one 2-D transpose [Sp, B H Sp]; W = dctx^T [D, M])
        PT = ops.transpose_bf16(Pd.view(B * H * Sp, Sp))
        dcxT = ops.transpose_bf16(dcx)
        ops.gemm_raw(PT, B * H * Sp, dcxT, M, dqkv[:, 2 * D:], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp, Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # dQ = dS K   (W = K^T [D, M])
        kT = ops.transpose_bf16(qkv[:, D: 2 * D])
        ops.gemm_raw(dS, Sp, kT, M, dqkv, 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp * Sp, Sp * Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # dK = dS^T Q  (A = dS^T, W = Q^T)
        dST = ops.transpose_bf16(dS.view(B * H * Sp, Sp))
        qT = ops.transpose_bf16(qkv[:, :D])
        ops.gemm_raw(dST, B * H * Sp, qT, M, dqkv[:, D: 2 * D], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp, Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # ---- in_proj (+ the residual branch's gradient)
        gWi = torch.empty(3 * D, D, device=dev, dtype=torch.float32)
        gbi = torch.empty(3 * D, device=dev, dtype=torch.float32)
        ops.wgrad_bf16(dqkv, xb, gWi, gbi, beta=0.0)
        dx = ops.linear_bf16(dqkv, Wi_b.t().contiguous(), residual=dpre)
        dx = dx.view(B, Sp, D)[:, :S].to(xdtype)
        return dx, gWi, gbi, gWo, gbo, dg, dbeta, None, None, None, None, None


def mha_norm(x: torch.Tensor, mha: torch.nn.MultiheadAttention, norm: torch.nn.LayerNorm, key_padding_mask: torch.Tensor,
             training: bool) -> torch.Tensor:
    p = float(mha.dropout) if training else 0.0
    return MhaNormFn.apply(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, norm.weight,
                           norm.bias, key_padding_mask, mha.num_heads, norm.eps, p, _next_seed() if p > 0.0 else 0)
