"""``LayerNorm(MultiheadAttention(x, x, x, key_padding_mask) + x)`` - the attention block of the cascaded+/hybrid+ branches
(avssl/module/kw_modules/TransformerModels.py:101-126, one head of 768 in the base recipes, 8 heads of 128 in the large ones) -
forward and backward on the library's kernels, any head_dim (scope rows a10 / f3): head dims that are not a multiple of 64
(96 = 768 / 8 in the hybrid+ base recipe) run with every head zero-padded to the next multiple of 64 inside the projection
weights, which changes nothing in the results (zero columns of q / k add nothing to the scores, zero columns of v give zero
context columns that meet zero columns of the output projection).

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
ROWS_LEAD, ROWS_TRAIL = 8, 8     # zero rows in front of / behind a row buffer handed to the CIF weight conv (>= its padding / kernel width)


grad_target = ops.grad_target


class BranchRows:
    """The attention block's output rows as the CIF module consumes them in place: ``full`` [B, P, D] bf16 (autograd output of
    MhaNormFn, the middle of ``flat`` [lead + B P + trail, D]); the frames of utterance b are its rows head .. head + S - 1, every
    other row of ``flat`` is zero."""

    def __init__(self, full, lead, trail, head, S):
        self.full, self.lead, self.trail, self.head, self.S = full, lead, trail, head, S
        self.B, self.P, self.D = full.shape


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
    def forward(ctx, x, Wi, bi, Wo, bo, g, beta, kpm, H, eps, p_drop, seed, p_res=0.0, seed_res=0, rows=None, out_dtype=None, rows_out=None):
        """``rows`` = (off, S): ``x`` is the encoder's resident bf16 row buffer [B, R, Dm] (R a multiple of 64, at least ``off``
        finite rows behind its end) and the sequence of utterance b is its rows off .. off + S - 1 - the block then reads that buffer
        in place (row pitch R, the rows behind a sequence are masked keys) and returns the gradient in the same layout, so neither a
        padded copy of the input nor a slice / cast of its gradient is made.

        ``rows_out`` = n_cls (0 / 1): the block's own output rows are handed on as they are - returns the bf16 buffer [B, Sp, Dm]
        itself (inside a flat buffer with ROWS_LEAD zero rows in front and ROWS_TRAIL behind; the n_cls leading rows of every
        utterance and the rows from S on zeroed: the zero padding the CIF weight conv reads in place), plus the n_cls CLS rows as
        fp32 [B, Dm]; the gradient then arrives in that layout too (bf16, zero outside the frames) and is used as it is."""
        dev, bf = x.device, torch.bfloat16
        if rows is None:
            B, S, Dm = x.shape
            Sp = _roundup(S, 64)
        else:
            off, S = rows
            B, Sp, Dm = x.shape
            assert x.dtype == bf and x.is_contiguous() and Sp % 64 == 0 and off + S <= Sp
        dh_true = Dm // H
        assert Dm % H == 0 and Dm % 64 == 0, "d_model must be a multiple of 64 and of the head count"
        dh = _roundup(dh_true, 64)                       # padded head dim; D = width of q, k, v and of the context
        D = H * dh
        M = B * Sp
        if rows is None:
            xb = torch.zeros(B, Sp, Dm, device=dev, dtype=bf)
            xb[:, :S] = x.detach()
            xb = xb.view(M, Dm)
        else:
            xb = torch.as_strided(x.detach(), (M, Dm), (Dm, 1), x.storage_offset() + off * Dm)
        if dh == dh_true:
            # bf16 working copy + its transpose (the input-gradient products' operand): one launch per weight and parameter version
            Wi_b, WiT = ops.derived_pair(Wi)
            Wo_b, WoT = ops.derived_pair(Wo)
            bi_f = bi.detach().float().contiguous()
        else:
            Wi_b = torch.zeros(3, H, dh, Dm, device=dev, dtype=bf)
            Wi_b[:, :, :dh_true] = Wi.detach().view(3, H, dh_true, Dm)
            Wi_b = Wi_b.view(3 * D, Dm)
            bi_f = torch.zeros(3, H, dh, device=dev, dtype=torch.float32)
            bi_f[:, :, :dh_true] = bi.detach().view(3, H, dh_true)
            bi_f = bi_f.view(3 * D)
            Wo_b = torch.zeros(Dm, H, dh, device=dev, dtype=bf)
            Wo_b[:, :, :dh_true] = Wo.detach().view(Dm, H, dh_true)
            Wo_b = Wo_b.view(Dm, D)
        qkv = ops.linear_bf16(xb, Wi_b, bi_f)                                                # [M, 3D]
        # ---- S x S core
        scores = torch.empty(B, H, Sp, Sp, device=dev, dtype=torch.float32)
        ops.gemm_raw(qkv, 3 * D, qkv[:, D:], 3 * D, scores, Sp, Sp, Sp, dh, out_f32=True, nb1=B, nb2=H,
                     sA=(Sp * 3 * D, dh), sW=(Sp * 3 * D, dh), sC=(H * Sp * Sp, Sp * Sp))
        lens = getattr(kpm, "_sc_lens", None)
        if lens is not None and tuple(kpm.shape) == (B, S) and int(lens[0].numel()) == B:
            key_pad = ops.len_mask(lens[0], Sp, lens[1]).view(torch.uint8)       # the same mask at the block's pitch (lens + add <= S)
        else:
            key_pad = torch.ones(B, Sp, device=dev, dtype=torch.uint8)
            key_pad[:, :S] = kpm
        P, Pd = ops.softmax_fwd(scores, key_pad, H * Sp, dh_true ** -0.5, p_drop, seed)          # P un-dropped (softmax backward), Pd dropped
        del scores
        vT = ops.transpose_bf16(qkv[:, 2 * D:])                                               # [D, M]: V^T of every utterance
        cx = torch.empty(M, D, device=dev, dtype=bf)
        ops.gemm_raw(Pd, Sp, vT, M, cx, D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp * Sp, Sp * Sp), sW=(Sp, dh * M), sC=(Sp * D, dh))
        pre = ops.linear_bf16(cx, Wo_b, bo.detach().float().contiguous(), residual=xb, drop_p=p_res, drop_seed=seed_res)
        # (the row kernels read gamma / beta with 16-byte loads: FlatAdam lays every parameter out on a 16-byte boundary, anything else
        # gets a copy)
        g32, b32 = ops.aligned16(g.detach().float()), ops.aligned16(beta.detach().float())
        flat = None
        if rows_out is None:
            out = ops.layernorm_bf16(pre, g32, b32, eps=eps)
        else:
            flat = torch.empty(ROWS_LEAD + M + ROWS_TRAIL, Dm, device=dev, dtype=bf)
            out = ops.layernorm_bf16(pre, g32, b32, eps=eps, out=flat[ROWS_LEAD: ROWS_LEAD + M])
        ctx.save_for_backward(xb, Wi_b, Wo_b, qkv, P, Pd if p_drop > 0.0 else None, cx, pre, g32)
        ctx.params = (Wi, bi, Wo, bo, g, beta)
        ctx.meta = (B, S, Sp, D, H, dh, eps, p_drop, seed, x.dtype, Dm, dh_true, p_res, seed_res, rows, rows_out)
        # transposed bf16 copies for the input-gradient products: per parameter version when the weights are used as they are
        ctx.wT = (WiT, WoT) if dh == dh_true else None
        if rows_out is None:
            return out.view(B, Sp, Dm)[:, :S].to(x.dtype if out_dtype is None else out_dtype)
        n_cls = int(rows_out)
        cls_rows = out.view(B, Sp, Dm)[:, 0].float() if n_cls else None
        ops.rows_zero_pad(flat, ROWS_LEAD, B, Sp, n_cls, S, ROWS_TRAIL)
        # (a tensor of its own over the middle of the flat buffer, not an autograd view)
        full = torch.empty(0, device=dev, dtype=bf).set_(flat.untyped_storage(), ROWS_LEAD * Dm, (B, Sp, Dm), (Sp * Dm, Dm, 1))
        return (full, cls_rows) if n_cls else full

    @staticmethod
    def backward(ctx, dout, dcls=None):
        xb, Wi_b, Wo_b, qkv, P, Pd, cx, pre, g = ctx.saved_tensors
        B, S, Sp, D, H, dh, eps, p_drop, seed, xdtype, Dm, dh_true, p_res, seed_res, rows, rows_out = ctx.meta
        dev, bf = pre.device, torch.bfloat16
        M = B * Sp
        if Pd is None:
            Pd = P
        if rows_out is None:
            dy = torch.zeros(B, Sp, Dm, device=dev, dtype=bf)
            dy[:, :S] = dout
        else:
            # the consumers (CIF weight head / integrate-and-fire, cif.py) return bf16 rows in the block's own layout, zero outside
            # the frames; the CLS rows' gradient is added into row 0 of every utterance (this buffer belongs to the backward pass)
            dy = dout if dout is not None else torch.zeros(B, Sp, Dm, device=dev, dtype=bf)
            assert dy.dtype == bf and dy.is_contiguous() and tuple(dy.shape) == (B, Sp, Dm)
            if dcls is not None:
                dy[:, 0] += dcls.to(bf)
        dy = dy.view(M, Dm)
        # parameter gradients go straight into the optimiser's buffers where they exist (grad_target): no temporaries, no add launches
        tWi, tbi, tWo, tbo, tg, tbeta = (grad_target(p_) if dh == dh_true else None for p_ in ctx.params)
        # ---- LayerNorm, out_proj
        if tg is not None and tbeta is not None:
            dpre, dg, dbeta = ops.layernorm_bwd(pre, dy, g, eps, acc=(tg, tbeta)), None, None
        else:
            dpre, dg, dbeta = ops.layernorm_bwd(pre, dy, g, eps, want_param_grads=True)
        dbr = ops.dropout_bf16(dpre, p_res, seed_res) if p_res > 0.0 else dpre                # the dropped branch's gradient
        if tWo is not None and tbo is not None:
            ops.wgrad_bf16(dbr, cx, tWo, tbo, beta=1.0)
            gWo = gbo = None
        else:
            gWo = torch.empty(Dm, D, device=dev, dtype=torch.float32)
            gbo = torch.empty(Dm, device=dev, dtype=torch.float32)
            ops.wgrad_bf16(dbr, cx, gWo, gbo, beta=0.0)
        WiT, WoT = ctx.wT if ctx.wT is not None else (Wi_b.t().contiguous(), Wo_b.t().contiguous())
        dcx = ops.linear_bf16(dbr, WoT)                                                        # [M, D]
        # ---- core: dP = dctx V^T ; dS = P (dP - rowsum(P dP)) scale
        dP = torch.empty(B, H, Sp, Sp, device=dev, dtype=torch.float32)
        ops.gemm_raw(dcx, D, qkv[:, 2 * D:], 3 * D, dP, Sp, Sp, Sp, dh, out_f32=True, nb1=B, nb2=H,
                     sA=(Sp * D, dh), sW=(Sp * 3 * D, dh), sC=(H * Sp * Sp, Sp * Sp))
        dS = ops.softmax_bwd(dP, P, dh_true ** -0.5, p_drop, seed)
        del dP
        dqkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
        tn = Sp % 256 == 0 and dh % 256 == 0                # the TN form reads Pd / dS and dctx / Q in place (sum over their rows)
        # dV = Pd^T dctx
        if tn:
            ops.gemm_raw(Pd, Sp, dcx, D, dqkv[:, 2 * D:], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                         sA=(H * Sp * Sp, Sp * Sp), sW=(Sp * D, dh), sC=(Sp * 3 * D, dh), tn=True)
        else:               # A = Pd^T from one 2-D transpose [Sp, B H Sp]; W = dctx^T [D, M]
            PT = ops.transpose_bf16(Pd.view(B * H * Sp, Sp))
            dcxT = ops.transpose_bf16(dcx)
            ops.gemm_raw(PT, B * H * Sp, dcxT, M, dqkv[:, 2 * D:], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                         sA=(H * Sp, Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # dQ = dS K   (W = K^T [D, M])
        kT = ops.transpose_bf16(qkv[:, D: 2 * D])
        ops.gemm_raw(dS, Sp, kT, M, dqkv, 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                     sA=(H * Sp * Sp, Sp * Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # dK = dS^T Q
        if tn:
            ops.gemm_raw(dS, Sp, qkv, 3 * D, dqkv[:, D: 2 * D], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                         sA=(H * Sp * Sp, Sp * Sp), sW=(Sp * 3 * D, dh), sC=(Sp * 3 * D, dh), tn=True)
        else:               # A = dS^T, W = Q^T
            dST = ops.transpose_bf16(dS.view(B * H * Sp, Sp))
            qT = ops.transpose_bf16(qkv[:, :D])
            ops.gemm_raw(dST, B * H * Sp, qT, M, dqkv[:, D: 2 * D], 3 * D, Sp, dh, Sp, nb1=B, nb2=H,
                         sA=(H * Sp, Sp), sW=(Sp, dh * M), sC=(Sp * 3 * D, dh))
        # ---- in_proj (+ the residual branch's gradient)
        if tWi is not None and tbi is not None:
            ops.wgrad_bf16(dqkv, xb, tWi, tbi, beta=1.0)
            gWi = gbi = None
        else:
            gWi = torch.empty(3 * D, Dm, device=dev, dtype=torch.float32)
            gbi = torch.empty(3 * D, device=dev, dtype=torch.float32)
            ops.wgrad_bf16(dqkv, xb, gWi, gbi, beta=0.0)
        if rows is None:
            dx = ops.linear_bf16(dqkv, WiT, residual=dpre)
            dx = dx.view(B, Sp, Dm)[:, :S].to(xdtype)
        else:
            # the gradient in the input's own layout: row m of the GEMM is row m + off of the buffer; the first ``off`` rows (the CLS
            # slot of the cascaded layout) get no gradient, the rows behind a sequence are exact zeros (dy = 0 there and pad keys
            # have P = 0), and the last ``off`` GEMM rows (padding of the last utterance) land behind the returned view
            off = rows[0]
            flat = torch.empty(M + off, Dm, device=dev, dtype=bf)
            if off:
                flat[:off].zero_()
            ops.linear_bf16(dqkv, WiT, residual=dpre, out=flat[off:])
            dx = flat[:M].view(B, Sp, Dm)
        if dh != dh_true:                                   # drop the gradients of the zero padding (grad_target is off then)
            gWi = gWi.view(3, H, dh, Dm)[:, :, :dh_true].reshape(3 * Dm, Dm)
            gbi = gbi.view(3, H, dh)[:, :, :dh_true].reshape(3 * Dm)
            gWo = gWo.view(Dm, H, dh)[:, :, :dh_true].reshape(Dm, Dm)
        return dx, gWi, gbi, gWo, gbo, dg, dbeta, None, None, None, None, None, None, None, None, None, None


def resident_rows(x: torch.Tensor):
    """(buffer [B, R, D] bf16, off) if ``x`` is the [B, S, D] view at row offset ``off`` of an encoder output buffer the attention
    block may read in place (weighted_sum.PaddedFeatHandle: pitch a multiple of 64, finite rows behind every sequence and behind the
    buffer's end), else None."""
    h = getattr(x, "_sc_handle", None)
    if h is None or not getattr(h, "inplace_ok", False):
        return None
    src = h.src
    D = src.shape[2]
    if x.dtype != torch.bfloat16 or x.dim() != 3 or x.shape[2] != D or x.stride() != src.stride() or x.shape[0] != src.shape[0]:
        return None
    delta = x.storage_offset() - src.storage_offset()
    if x.untyped_storage().data_ptr() != src.untyped_storage().data_ptr() or delta < 0 or delta % D or delta // D + x.shape[1] > src.shape[1]:
        return None
    return src, delta // D


def mha_norm(x: torch.Tensor, mha: torch.nn.MultiheadAttention, norm: torch.nn.LayerNorm, key_padding_mask: torch.Tensor,
             training: bool, p_res: float = 0.0, out_dtype=None, rows_out=None):
    """``norm(x + dropout_res(MHA(x, x, x, key_padding_mask)))``; ``p_res`` = nn.TransformerEncoderLayer's dropout1 (the bare
    MultiheadAttentionAndNorm block has none).  An ``x`` that is a view of the encoder's resident output rows is read in place.
    ``rows_out`` = n_cls (0 / 1): instead of the [B, S, D] result return (BranchRows, CLS rows fp32 [B, D] or None) - the block's
    bf16 output rows for a consumer that reads them in place (cif.CIF)."""
    p = float(mha.dropout) if training else 0.0
    p_res = float(p_res) if training else 0.0
    res = resident_rows(x)
    rows = None
    if res is not None:
        src, off = res
        x, rows = src, (off, x.shape[1])
    S = rows[1] if rows is not None else x.shape[1]
    res = MhaNormFn.apply(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, norm.weight,
                          norm.bias, key_padding_mask, mha.num_heads, norm.eps, p, _next_seed() if p > 0.0 else 0,
                          p_res, _next_seed() if p_res > 0.0 else 0, rows, out_dtype, rows_out)
    if rows_out is None:
        return res
    full, cls_rows = res if rows_out else (res, None)
    return BranchRows(full, ROWS_LEAD, ROWS_TRAIL, int(rows_out), S - int(rows_out)), cls_rows


class FfnNormFn(torch.autograd.Function):
    """``LayerNorm(x + drop2(W2 drop1(gelu(W1 x + b1)) + b2))``: the feed-forward half of a post-LN nn.TransformerEncoderLayer
    (avssl/module/kw_modules/TransformerModels.py:60-70) on the library's kernels: bias + GELU / bias + dropout + residual in the
    GEMM epilogues, hash dropout masks regenerated in the backward, split-K weight gradients."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, g, beta, eps, p_drop, seed1, seed2):
        B, S, D = x.shape
        dev, bf = x.device, torch.bfloat16
        Sp = _roundup(S, 64)
        M = B * Sp
        xb = torch.zeros(B, Sp, D, device=dev, dtype=bf)
        xb[:, :S] = x.detach()
        xb = xb.view(M, D)
        W1b, W2b = W1.detach().to(bf).contiguous(), W2.detach().to(bf).contiguous()
        u = ops.linear_bf16(xb, W1b, b1.detach().float().contiguous())                        # pre-activation (kept for GELU')
        f = ops.act_bf16(u, 1)
        if p_drop > 0.0:
            ops.dropout_bf16(f, p_drop, seed1, out=f)
        pre = ops.linear_bf16(f, W2b, b2.detach().float().contiguous(), residual=xb, drop_p=p_drop, drop_seed=seed2)
        g32, b32 = g.detach().float().clone(), beta.detach().float().clone()
        out = ops.layernorm_bf16(pre, g32, b32, eps=eps)
        ctx.save_for_backward(xb, W1b, W2b, u, f, pre, g32)
        ctx.meta = (B, S, Sp, D, eps, p_drop, seed1, seed2, x.dtype)
        return out.view(B, Sp, D)[:, :S].to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        xb, W1b, W2b, u, f, pre, g = ctx.saved_tensors
        B, S, Sp, D, eps, p_drop, seed1, seed2, xdtype = ctx.meta
        dev, bf = dout.device, torch.bfloat16
        M, F_ = B * Sp, W1b.shape[0]
        dy = torch.zeros(B, Sp, D, device=dev, dtype=bf)
        dy[:, :S] = dout
        dy = dy.view(M, D)
        dpre, dg, dbeta = ops.layernorm_bwd(pre, dy, g, eps, want_param_grads=True)
        dbr = ops.dropout_bf16(dpre, p_drop, seed2) if p_drop > 0.0 else dpre
        gW2 = torch.empty(D, F_, device=dev, dtype=torch.float32)
        gb2 = torch.empty(D, device=dev, dtype=torch.float32)
        ops.wgrad_bf16(dbr, f, gW2, gb2, beta=0.0)
        df = ops.linear_bf16(dbr, W2b.t().contiguous())
        if p_drop > 0.0:
            ops.dropout_bf16(df, p_drop, seed1, out=df)
        du = ops.act_bf16(u, 1, df=df, out=df)
        gW1 = torch.empty(F_, D, device=dev, dtype=torch.float32)
        gb1 = torch.empty(F_, device=dev, dtype=torch.float32)
        ops.wgrad_bf16(du, xb, gW1, gb1, beta=0.0)
        dx = ops.linear_bf16(du, W1b.t().contiguous(), residual=dpre)
        return dx.view(B, Sp, D)[:, :S].to(xdtype), gW1, gb1, gW2, gb2, dg, dbeta, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dimension on the row kernels (bf16 rows, fp32 statistics)."""

    @staticmethod
    def forward(ctx, x, g, beta, eps):
        shape = x.shape
        D = shape[-1]
        xb = x.detach().reshape(-1, D).to(torch.bfloat16).contiguous()
        g32, b32 = g.detach().float().clone(), beta.detach().float().clone()
        ctx.save_for_backward(xb, g32)
        ctx.meta = (shape, eps, x.dtype)
        return ops.layernorm_bf16(xb, g32, b32, eps=eps).view(shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xb, g = ctx.saved_tensors
        shape, eps, dtype = ctx.meta
        dyb = dy.reshape(-1, shape[-1]).to(torch.bfloat16).contiguous()
        dx, dg, db = ops.layernorm_bwd(xb, dyb, g, eps, want_param_grads=True)
        return dx.view(shape).to(dtype), dg, db, None
