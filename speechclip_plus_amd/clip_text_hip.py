"""Frozen CLIP text transformer (forward + gradient w.r.t. its input) on the library's bf16 kernels.

The cascaded branches push the keyword embeddings THROUGH the frozen text tower (avssl/module/clip_official.py:222-279 ->
openai/CLIP ``Transformer``: pre-LN residual blocks, causal nn.MultiheadAttention with head_dim 64, QuickGELU MLP), so a
train step needs the tower's forward and its input gradient but no weight gradient.  Per layer, on ``B * 128`` padded rows
(77 tokens per sample, rows 77..127 scratch that stay finite and carry zero gradient):

    forward   LN -> QKV GEMM -> causal attention (+ LSE) -> out-proj GEMM (+ residual) -> LN -> FC GEMM -> QuickGELU
              -> proj GEMM (+ residual)
    backward  dgrad GEMMs against transposed weight copies, QuickGELU' , LayerNorm backward fused with the residual add,
              attention backward (csrc/attention_bwd.hip)

Weights are converted once per device (bf16 + transposed bf16 copies; the tower is frozen).  Used when the tower is frozen, on a
GPU, with head_dim 64 (both published text towers: ViT-B/32 512 / 8, ViT-L/14 768 / 12); otherwise clip_text falls back to
the stock-op blocks it also defines (trainable tower: not a shipped recipe).
"""
from typing import List

import torch

from . import ops

ROWS = 128          # padded rows per sample (CONTEXT_LEN = 77 -> one 128-row attention block)


class _LayerW:
    __slots__ = ("wqkv", "bqkv", "wqkvT", "wo", "bo", "woT", "w1", "b1", "w1T", "w2", "b2", "w2T", "g1", "be1", "g2", "be2", "eps1", "eps2")


def prepare_weights(transformer, device) -> List[_LayerW]:
    bf = lambda t: t.detach().to(device=device, dtype=torch.bfloat16).contiguous()
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
    out = []
    for blk in transformer.resblocks:
        w = _LayerW()
        w.wqkv, w.bqkv = bf(blk.attn.in_proj_weight), f32(blk.attn.in_proj_bias)
        w.wqkvT = bf(blk.attn.in_proj_weight.t())
        w.wo, w.bo, w.woT = bf(blk.attn.out_proj.weight), f32(blk.attn.out_proj.bias), bf(blk.attn.out_proj.weight.t())
        w.w1, w.b1, w.w1T = bf(blk.mlp.c_fc.weight), f32(blk.mlp.c_fc.bias), bf(blk.mlp.c_fc.weight.t())
        w.w2, w.b2, w.w2T = bf(blk.mlp.c_proj.weight), f32(blk.mlp.c_proj.bias), bf(blk.mlp.c_proj.weight.t())
        w.g1, w.be1, w.eps1 = f32(blk.ln_1.weight), f32(blk.ln_1.bias), blk.ln_1.eps
        w.g2, w.be2, w.eps2 = f32(blk.ln_2.weight), f32(blk.ln_2.bias), blk.ln_2.eps
        out.append(w)
    return out


class TextTowerFn(torch.autograd.Function):
    """x [B, T <= 128, W] (any float dtype) -> transformer(x) [B, T, W] fp32; gradient w.r.t. x only."""

    @staticmethod
    def forward(ctx, x, weights, heads):
        B, T, W = x.shape
        dev = x.device
        M = B * ROWS
        X = torch.zeros(B, ROWS, W, device=dev, dtype=torch.bfloat16)
        X[:, :T] = x.detach().to(torch.bfloat16)
        X = X.view(M, W)
        valid = torch.full((B,), T, device=dev, dtype=torch.int32)
        scale = (W // heads) ** -0.5
        saved = []
        for w in weights:
            h = ops.layernorm_bf16(X, w.g1, w.be1, eps=w.eps1)
            qkv = ops.linear_bf16(h, w.wqkv, w.bqkv)
            vt = ops.head_transpose(qkv[:, 2 * W:], B, ROWS, heads)
            att = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
            lse2 = torch.empty(B, heads, ROWS, device=dev, dtype=torch.float32)
            ops.attn_fwd(qkv[:, : 2 * W], vt, valid, att, B, ROWS, heads, W, scale, lse2=lse2, causal=True)
            X2 = ops.linear_bf16(att, w.wo, w.bo, residual=X)
            h2 = ops.layernorm_bf16(X2, w.g2, w.be2, eps=w.eps2)
            u = ops.linear_bf16(h2, w.w1, w.b1)
            f = ops.act_bf16(u, 2)
            Xn = ops.linear_bf16(f, w.w2, w.b2, residual=X2)
            saved.append((X, qkv, att, lse2, X2, u))
            X = Xn
        ctx.weights, ctx.saved, ctx.valid, ctx.dims = weights, saved, valid, (B, T, W, heads, scale)
        return X.view(B, ROWS, W)[:, :T].float()

    @staticmethod
    def backward(ctx, dy):
        B, T, W, heads, scale = ctx.dims
        dev = dy.device
        M = B * ROWS
        dX = torch.zeros(B, ROWS, W, device=dev, dtype=torch.bfloat16)
        dX[:, :T] = dy.to(torch.bfloat16)
        dX = dX.view(M, W)
        for w, (X, qkv, att, lse2, X2, u) in zip(reversed(ctx.weights), reversed(ctx.saved)):
            df = ops.linear_bf16(dX, w.w2T)
            du = ops.act_bf16(u, 2, df=df)
            dh2 = ops.linear_bf16(du, w.w1T)
            dX2 = ops.layernorm_bwd(X2, dh2, w.g2, w.eps2, dres=dX)
            datt = ops.linear_bf16(dX2, w.woT)
            dqkv = torch.empty(M, 3 * W, device=dev, dtype=torch.bfloat16)
            ops.attn_bwd(qkv[:, :W], qkv[:, W: 2 * W], qkv[:, 2 * W:], att, datt, lse2, ctx.valid, dqkv[:, :W], dqkv[:, W: 2 * W],
                         dqkv[:, 2 * W:], B, ROWS, heads, scale, causal=True, q_rows=T)
            dh1 = ops.linear_bf16(dqkv, w.wqkvT)
            dX = ops.layernorm_bwd(X, dh1, w.g1, w.eps1, dres=dX2)
        ctx.saved = None
        return dX.view(B, ROWS, W)[:, :T].float(), None, None
