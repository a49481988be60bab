"""Frozen CLIP text transformer (forward + gradient w.r.t. its input) on the library's bf16 kernels.

The cascaded branches push the keyword embeddings THROUGH the frozen text tower (avssl/module/clip_official.py:222-279 ->
openai/CLIP ``Transformer``: pre-LN residual blocks, causal nn.MultiheadAttention with head_dim 64, QuickGELU MLP), so a
train step needs the tower's forward and its input gradient but no weight gradient.  Per layer, on ``B * SEG`` rows, SEG = 32 / 64 /
128 = the prompt length T rounded up (rows T..SEG-1 scratch that stay finite and carry zero gradient).  The tower is CAUSAL and
only the end-of-text row is read, so ``encode_keywords`` hands over the prompt PREFIX up to the last end-of-text position only
(2 + the batch's largest keyword count, ~27 tokens for 10 s utterances instead of 77): with SEG < 128 the sequences lie back to
back, 128 / SEG to an attention block, and the attention kernels mask causally inside aligned segments (``causal = SEG``) - the
GEMMs, LayerNorms and activations then run on B * SEG dense rows instead of B * 128.  SEG = 32 (every shipped recipe: 8 keywords)
has kernels of its own, one wave per (sequence, head), forward 1 launch and backward 1 launch from qkv and d out alone
(csrc/attn_short.hip) instead of V^T + forward and row sums + dQ + dK/dV on 128-row blocks that are 7/8 mask:

    forward   LN -> QKV GEMM -> causal attention (+ LSE) -> out-proj GEMM (+ residual) -> LN -> FC GEMM -> QuickGELU
              -> proj GEMM (+ residual)        (SC_TOWER_FOLD_LN=1: both LayerNorms in the prologue of the GEMM behind them, 5 launches
              per layer - measured slower, see FOLD_LN)
    backward  dgrad GEMMs against transposed weight copies, QuickGELU' , LayerNorm backward fused with the residual add,
              attention backward (csrc/attention_bwd.hip)

Weights are converted once per device (bf16 + transposed bf16 copies; the tower is frozen).  Used when the tower is frozen, on a
GPU, with head_dim 64 (both published text towers: ViT-B/32 512 / 8, ViT-L/14 768 / 12); otherwise clip_text falls back to
the stock-op blocks it also defines (trainable tower: not a shipped recipe).
"""
from typing import List

import torch

from . import ops

import os as _os
# LayerNorm in the prologue of the tower's QKV / fc GEMMs (round 6: built, parity-green - closer to fp64 than LayerNorm + GEMM, one bf16
# rounding less - and SLOWER: every one of a row tile's 24-48 column workgroups re-reads and re-reduces the tile's rows before its
# K loop; same-box alternating runs: cascaded+ 14.76 -> 14.97 ms per step (one stream 16.1 -> 16.45), hybrid+ large 31.85 -> 32.35).
# Opt-in: SC_TOWER_FOLD_LN=1; the default keeps the LayerNorm launches.
FOLD_LN = _os.environ.get("SC_TOWER_FOLD_LN", "0") == "1"
BLOCK = 128         # rows of an attention block
SHORT = 32          # segment length served by the one-wave attention kernels


def _segment(T: int) -> int:
    """rows per sample: the prompt length rounded up to 32 / 64 / 128"""
    if T > BLOCK:
        raise ValueError(f"text tower: {T} tokens (at most {BLOCK})")
    return 32 if T <= 32 else 64 if T <= 64 else 128


class _LayerW:
    __slots__ = ("wqkv", "bqkv", "wqkvT", "wo", "bo", "woT", "w1", "b1", "w1T", "w2", "b2", "w2T", "g1", "be1", "g2", "be2", "eps1", "eps2",
                 "wqkv_ln", "sqkv", "cqkv", "w1_ln", "s1", "c1")


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
        # round 6: ln_1 / ln_2 folded into the QKV / fc GEMMs (LayerNorm in the GEMM's prologue: ops.fold_layernorm, csrc/gemm_bf16.hip) -
        # the forward runs 5 launches per layer instead of 7; the backward keeps the unfolded weights (it differentiates through the
        # LayerNorm from the saved raw rows)
        w.wqkv_ln, w.sqkv, w.cqkv = ops.fold_layernorm(blk.attn.in_proj_weight.to(device), blk.attn.in_proj_bias.to(device), w.g1, w.be1)
        w.w1_ln, w.s1, w.c1 = ops.fold_layernorm(blk.mlp.c_fc.weight.to(device), blk.mlp.c_fc.bias.to(device), w.g2, w.be2)
        out.append(w)
    return out


_valid_cache = {}


def _full_blocks(NB: int, dev) -> torch.Tensor:
    """valid_len of the attention launch: every 128-row block is full (the causal mask does the rest); one resident vector per size"""
    key = (NB, str(dev))
    if key not in _valid_cache:
        _valid_cache[key] = torch.full((NB,), BLOCK, device=dev, dtype=torch.int32)
    return _valid_cache[key]


def _geometry(B: int, T: int):
    SEG = _segment(T)
    per = BLOCK // SEG                                   # sequences per attention block
    Bp = -(-B // per) * per                              # padded up to whole blocks (pad sequences: zeros)
    return SEG, Bp, Bp * SEG, 1 if SEG == BLOCK else SEG


def tower_forward(X: torch.Tensor, weights, heads: int, causal: int):
    """X [M, W] bf16 packed rows -> (output rows [M, W] bf16, what the backward needs)"""
    M, W = X.shape
    dev = X.device
    NB = M // BLOCK
    valid = _full_blocks(NB, dev)
    scale = (W // heads) ** -0.5
    saved = []
    short = causal == SHORT                              # one 32-row segment per sequence: the one-wave kernels (csrc/attn_short.hip)
    fold = FOLD_LN and W <= 1024
    for w in weights:
        if fold:
            qkv = ops.linear_bf16(X, w.wqkv_ln, w.cqkv, ln_colsum=w.sqkv, ln_eps=w.eps1)          # LN1 in the GEMM's prologue
        else:
            h = ops.layernorm_bf16(X, w.g1, w.be1, eps=w.eps1)
            qkv = ops.linear_bf16(h, w.wqkv, w.bqkv)
        if short:
            att, lse2 = ops.attn32_fwd(qkv, heads, scale), None
        else:
            vt = ops.head_transpose(qkv[:, 2 * W:], NB, BLOCK, heads)
            att = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
            lse2 = torch.empty(NB, heads, BLOCK, device=dev, dtype=torch.float32)
            ops.attn_fwd(qkv[:, : 2 * W], vt, valid, att, NB, BLOCK, heads, W, scale, lse2=lse2, causal=causal)
        X2 = ops.linear_bf16(att, w.wo, w.bo, residual=X)
        u = torch.empty(M, w.w1.shape[0], device=dev, dtype=torch.bfloat16)
        if fold:
            f = ops.linear_bf16(X2, w.w1_ln, w.c1, act=2, aux=u, aux_mode=1, ln_colsum=w.s1, ln_eps=w.eps2)      # LN2 in the prologue
        else:
            h2 = ops.layernorm_bf16(X2, w.g2, w.be2, eps=w.eps2)
            f = ops.linear_bf16(h2, w.w1, w.b1, act=2, aux=u, aux_mode=1)          # u = fc1 pre-activation (kept), f = QuickGELU(u)
        Xn = ops.linear_bf16(f, w.w2, w.b2, residual=X2)
        saved.append((X, qkv, None if short else att, lse2, X2, u))
        X = Xn
    return X, saved


def tower_backward(dX: torch.Tensor, weights, saved, heads: int, causal: int, q_rows: int) -> torch.Tensor:
    """dX [M, W] bf16 (rows that carry no gradient: zero) -> gradient w.r.t. the tower's input rows"""
    M, W = dX.shape
    dev = dX.device
    NB = M // BLOCK
    valid = _full_blocks(NB, dev)
    scale = (W // heads) ** -0.5
    for w, (X, qkv, att, lse2, X2, u) in zip(reversed(weights), reversed(saved)):
        du = ops.linear_bf16(dX, w.w2T, act=2, aux=u, aux_mode=2)                 # (dX W2) * QuickGELU'(u) in the GEMM's epilogue
        dh2 = ops.linear_bf16(du, w.w1T)
        dX2 = ops.layernorm_bwd(X2, dh2, w.g2, w.eps2, dres=dX)
        datt = ops.linear_bf16(dX2, w.woT)
        if causal == SHORT:
            dqkv = ops.attn32_bwd(qkv, datt, heads, scale)
        else:
            dqkv = torch.empty(M, 3 * W, device=dev, dtype=torch.bfloat16)
            ops.attn_bwd(qkv[:, :W], qkv[:, W: 2 * W], qkv[:, 2 * W:], att, datt, lse2, valid, dqkv[:, :W], dqkv[:, W: 2 * W],
                         dqkv[:, 2 * W:], NB, BLOCK, heads, scale, causal=causal, q_rows=q_rows)
        dh1 = ops.linear_bf16(dqkv, w.wqkvT)
        dX = ops.layernorm_bwd(X, dh1, w.g1, w.eps1, dres=dX2)
    return dX


class TextTowerFn(torch.autograd.Function):
    """x [B, T <= 128, W] (any float dtype) -> transformer(x) [B, T, W] fp32; gradient w.r.t. x only."""

    @staticmethod
    def forward(ctx, x, weights, heads):
        B, T, W = x.shape
        SEG, Bp, M, causal = _geometry(B, T)
        X = torch.zeros(Bp, SEG, W, device=x.device, dtype=torch.bfloat16)
        X[:B, :T] = x.detach().to(torch.bfloat16)
        X, saved = tower_forward(X.view(M, W), weights, heads, causal)
        ctx.weights, ctx.saved, ctx.dims = weights, saved, (B, T, W, heads, SEG, Bp, causal)
        return X.view(Bp, SEG, W)[:B, :T].float()

    @staticmethod
    def backward(ctx, dy):
        B, T, W, heads, SEG, Bp, causal = ctx.dims
        M = Bp * SEG
        dX = torch.zeros(Bp, SEG, W, device=dy.device, dtype=torch.bfloat16)
        dX[:B, :T] = dy.to(torch.bfloat16)
        dX = tower_backward(dX.view(M, W), ctx.weights, ctx.saved, heads, causal, BLOCK if SEG < BLOCK else T)
        ctx.saved = None
        return dX.view(Bp, SEG, W)[:B, :T].float(), None, None


class KeywordTowerFn(torch.autograd.Function):
    """The whole keyword path of ``ClipModel.encode_keywords`` (clip_official.py:222-279) around the frozen tower:
    keywords [B, N, W] fp32, count [B] int64 -> the B end-of-text rows of transformer([SOT, kw_1 .. kw_n, EOT, 0 ..] + pos), fp32 [B, W].

    One launch assembles the tower's packed bf16 rows (csrc/prompt.hip), one reads the end-of-text rows back; the backward scatters
    the rows' gradient into a zero [M, W] buffer, runs the tower's input gradient and extracts d(keywords) - four launches around the
    tower where the element-wise formulation took ~35 (zeros / scatter / embedding / where / add / pad / cast / advanced index and
    their autograd nodes).  Same arithmetic: fp32 embedding + position sums rounded once to bf16, gradients rounded to bf16 where the
    tower takes them."""

    @staticmethod
    def forward(ctx, keywords, count, tok, pos, weights, heads, n_pos, clamped):
        B, N, W = keywords.shape
        SEG, Bp, M, causal = _geometry(B, n_pos)
        kw = keywords.detach()
        if kw.dtype != torch.float32 or kw.stride(2) != 1 or (N > 0 and kw.stride(1) != W) or kw.data_ptr() % 16:
            kw = kw.float().contiguous()
        count = count.detach().to(dtype=torch.int64).contiguous()
        X, eot_row = ops.prompt_assemble(kw, count, tok, pos, Bp, SEG, n_pos, clamped)
        X, saved = tower_forward(X, weights, heads, causal)
        ctx.weights, ctx.saved, ctx.dims = weights, saved, (B, N, W, heads, SEG, M, causal, n_pos, keywords.dtype)
        ctx.count, ctx.eot_row = count, eot_row
        return ops.rows_gather(X, eot_row)

    @staticmethod
    def backward(ctx, d_rows):
        B, N, W, heads, SEG, M, causal, n_pos, dtype = ctx.dims
        dX = ops.rows_scatter(d_rows.float().contiguous(), ctx.eot_row, M, SEG)
        dX = tower_backward(dX, ctx.weights, ctx.saved, heads, causal, BLOCK if SEG < BLOCK else n_pos)
        ctx.saved = None
        dk = ops.prompt_assemble_bwd(dX, ctx.count, B, N, SEG, n_pos)
        return (dk if dtype == torch.float32 else dk.to(dtype)), None, None, None, None, None, None, None
