"""CLS row of the parallel branch head, forward and backward, on the library's own kernels.

What the reference computes (avssl/model/kw_branches.py:266-280 over nn.TransformerEncoderLayer as built by
avssl/module/kw_modules/TransformerModels.py:48-97): ``Linear(LN_f(layer([cls ; feat], mask)))[:, 0]``.  Only row 0 is kept,
so (transformer_models.py) the CLS query attends over the keys without materialising K / V (csrc/clspool.hip) and the rest
of the post-LN layer is a chain of B-row fp32 products on the master weights (csrc/headtail.hip):

    q = Wq cls + bq ;  a_h = dh^-1/2 Wk_h^T q_h ;  m[b,h] = sum_s softmax_s(a_h . X[b,s]) X[b,s]
    ctx = concat_h(Wv_h m[b,h]) + bv ;  x1 = LN1(cls + Wo ctx + bo) ;  x2 = LN2(x1 + W2 gelu(W1 x1 + b1) + b2)
    out = Wp LN_f(x2) + bp

One autograd node: parameter gradients are ACCUMULATED IN PLACE into ``p.grad`` (the views of the optimiser's flat gradient
buffer, optim.FlatAdam) by the weight-gradient products themselves (beta = 1), so the backward adds no per-parameter
accumulate kernels; only ``cls`` (which guarantees the node is reached), the weighted-sum logits and a generic ``feat``
input receive their gradient through autograd's return values.

Train mode (``module.training`` and dropout p > 0; nn.TransformerEncoderLayer's four dropout sites, p = 0.1 in every
recipe): attention weights (a [B,H,R] multiplier consumed by the pooling kernel), dropout1 on the attention output,
dropout on the activation, dropout2 on the FFN output.  The masks come from torch's device generator (so
``torch.manual_seed`` makes a step reproducible) as 0 / (1/(1-p)) multipliers and are kept for the backward.
"""
import os
from typing import Optional

import torch
from torch import nn

from . import ops


def _gacc(p: torch.Tensor) -> torch.Tensor:
    """Accumulation target of a parameter's gradient (fp32, contiguous, same shape)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    assert p.grad.dtype == torch.float32 and p.grad.is_contiguous()
    return p.grad


def _lin(x, W, b):
    """y[B,N] = x[B,K] W[N,K]^T + b   (the one-row query path: M = 1 or H rows)"""
    Bn, K = x.shape
    N = W.shape[0]
    y = torch.empty(Bn, N, device=x.device, dtype=torch.float32)
    ops.sgemm_ex(x, (K, 1, 0), W, (K, 1, 0), y, N, Bn, N, K, bias=b)
    return y


def _wgrad(dy, x, gW, gb):
    """gW [N, K] += dy[B, N]^T x[B, K] ; gb [N] += column sums of dy - ONE launch (sc_rt_gemm, weight-gradient form)"""
    Bn, N = dy.shape
    K = x.shape[1]
    ops.rt_gemm(dy, x, N, K, Bn, a_kmajor=True, b_kmajor=True, lda=N, ldb=K, out=gW, beta=1.0, gb=gb)


def _dgrad(dy, W):
    """dx[B, K] = dy[B, N] W[N, K] as partial slices (the consumer adds them)"""
    Bn, N = dy.shape
    K = W.shape[1]
    return ops.rt_gemm(dy, W, Bn, K, N, b_kmajor=True, ldb=K, split=True)


def cls_query(module, cls: torch.Tensor):
    """(x0, q, Qm, a): the CLS query folded into the key projection, a_h = dh^-1/2 Wk_h^T (Wq cls + bq)_h.  It depends on
    parameters only, so it is cached per parameter version (optim.param_generation() + the tensors' own versions) and the trainer
    computes it on its SIDE stream right after the optimiser step (train.ContrastiveTrainer._finish -> prefetch), i.e. under the next
    step's encoder forward: six small launches leave the critical path between the weighted sum and the score sweep."""
    from .optim import param_generation
    layer = module.model.layers[0]
    att = layer.self_attn
    key = (param_generation(), cls.data_ptr(), cls._version, att.in_proj_weight.data_ptr(), att.in_proj_weight._version,
           att.in_proj_bias._version)
    cache = getattr(module, "_sc_cls_query", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    D, H = module.d_model, module.nhead
    dh = D // H
    Wi, bi = att.in_proj_weight.detach(), att.in_proj_bias.detach()
    x0 = cls.detach().reshape(1, D).float().contiguous()
    q = _lin(x0, Wi[:D], bi[:D])                                  # [1, D]
    Qm = torch.empty(H, D, device=x0.device, dtype=torch.float32)
    ops.headmask(q, Qm, H, D, dh, gather=False)
    a = torch.empty(H, D, device=x0.device, dtype=torch.float32)
    ops.sgemm_ex(Qm, (D, 1, 0), Wi[D: 2 * D], (1, D, 0), a, D, H, D, D, alpha=dh ** -0.5)
    x0b = x0.to(torch.bfloat16)
    val = (x0, q, Qm, a, x0b)
    module._sc_cls_query = (key, val)
    return val


_aux = {}
_AUX_ON = os.environ.get("SC_HEAD_AUX_STREAM", "1") == "1"


def _aux_stream(dev) -> "torch.cuda.Stream":
    """A second stream for the parameter-only half of the head's backward (it runs beside the weighted-sum sweep)."""
    return ops.shared_stream("head_aux", dev)


class ParallelHeadFn(torch.autograd.Function):
    """inputs : cls [1,1,D] fp32 (parameter), ws_weights (weighted-sum logits or None), feat (generic path or None),
               then constants: the TransformerEncoder module, the projection nn.Linear (or None), the encoder handle,
               src [B,R,D] bf16 (row 0 = CLS slot), lens int32 [B], B, R
       output : [B, E] (or [B, D] without projection) fp32

    Round 3: the B-row products are sc_rt_gemm launches whose contraction is split over the chip; their partial slices are added by
    the consumer (the next product's operand load, a LayerNorm / activation row kernel) together with the bias, so no reduce
    launch and no separate bias / GELU / dropout / residual launches exist; every weight-gradient product also emits its bias
    gradient.  The dropout multipliers of the three B-row sites are hash bits evaluated inside the consumers (the same bits
    ops.dropout_mult would materialise for the same seed)."""

    @staticmethod
    def forward(ctx, cls, ws_weights, feat, module, proj, handle, src, lens, B, R):
        layer = module.model.layers[0]
        D, H = module.d_model, module.nhead
        dh = D // H
        dev = src.device
        att = layer.self_attn
        Wi, bi = att.in_proj_weight.detach(), att.in_proj_bias.detach()
        Wq, Wk, Wv = Wi[:D], Wi[D: 2 * D], Wi[2 * D:]
        # ---- CLS query folded into the key projection: a_h = dh^-1/2 Wk_h^T q_h (bk shifts all scores alike: cancels); cached
        x0, q, Qm, a, x0b = cls_query(module, cls)
        src = src.detach()
        src[:, 0] = x0b                                         # CLS slot of the padded [CLS ; frames] buffer
        scores = ops.cls_scores(src, a, False, B, R, D, H)
        pd = float(module.dropout) if module.training else 0.0
        s_att, s1, sf, s2 = (ops.next_mult_seed() for _ in range(4)) if pd > 0 else (0, 0, 0, 0)   # order = the sites' order in the layer
        mult = ops.dropout_mult((B, H, R), pd, dev, seed=s_att) if pd > 0 else None
        # m [B, H, D]; psum [B, H] = sum_s(p mult): ctx_h = Wv_h m_h + bv_h * psum (the dropped weights no longer sum to 1; eval: 1)
        if mult is not None:
            p, m, psum = ops.cls_pool_fwd(src, scores, lens, B, R, D, H, mult, want_psum=True)
        else:
            (p, m), psum = ops.cls_pool_fwd(src, scores, lens, B, R, D, H, None), None
        cxs = ops.rt_gemm(m, Wv, B, dh, D, nbatch=H, lda=H * D, a_z=D, ldb=D, b_z=dh * D, split=True, ldc=D, c_z=dh)
        cx = ops.rt_elem(cxs, 0, bias=bi[2 * D:], rowscale=psum, group=dh)
        # ---- out_proj -> dropout1 -> + cls -> LN1
        attn_s = ops.rt_gemm(cx, att.out_proj.weight.detach(), B, D, D, split=True)
        x1, xh1, rs1 = ops.rt_ln_fwd(attn_s, att.out_proj.bias.detach(), x0, 0, layer.norm1.weight.detach(), layer.norm1.bias.detach(),
                                     module.layer_norm_eps, drop_p=pd, drop_seed=s1)
        # ---- FFN: linear1 -> GELU -> dropout -> linear2 -> dropout2 -> + x1 -> LN2 -> final LN
        F_ = layer.linear1.out_features
        us = ops.rt_gemm(x1, layer.linear1.weight.detach(), B, F_, D, split=True)
        u, f = ops.rt_elem(us, 1, bias=layer.linear1.bias.detach(), drop_p=pd, drop_seed=sf)
        y2s = ops.rt_gemm(f, layer.linear2.weight.detach(), B, D, F_, split=True)
        fin = module.model.norm
        x2, xh2, rs2, x3, xh3, rs3 = ops.rt_ln_fwd(y2s, layer.linear2.bias.detach(), x1, D, layer.norm2.weight.detach(),
                                                   layer.norm2.bias.detach(), module.layer_norm_eps, fin.weight.detach(),
                                                   fin.bias.detach(), fin.eps, drop_p=pd, drop_seed=s2)
        if proj is not None:
            outs = ops.rt_gemm(x3, proj.weight.detach(), B, proj.weight.shape[0], D, split=True)
            out, e, rn = ops.rt_l2norm_fwd(outs, proj.bias.detach())
            out._sc_unit = (e, rn)           # UnitRowsFn (model.forward's L2 normalisation) re-uses these instead of a second launch
        else:
            out = x3
        ctx.mod, ctx.proj, ctx.handle, ctx.dims = module, proj, handle, (B, R, D, H)
        ctx.feat_meta = None if feat is None else (feat.shape, feat.dtype)
        ctx.drop = (pd, s1, sf, s2)
        ctx.masks = (mult, psum)
        ctx.save_for_backward(src, lens, x0, q, Qm, a, p, m, cx, xh1, rs1, x1, u, f, xh2, rs2, xh3, rs3, x3)
        return out

    @staticmethod
    def backward(ctx, d_out):
        src, lens, x0, q, Qm, a, p, m, cx, xh1, rs1, x1, u, f, xh2, rs2, xh3, rs3, x3 = ctx.saved_tensors
        module, proj = ctx.mod, ctx.proj
        B, R, D, H = ctx.dims
        pd, s1, sf, s2 = ctx.drop
        mult, psum = ctx.masks
        dh = D // H
        dev = src.device
        layer, fin = module.model.layers[0], module.model.norm
        att = layer.self_attn
        Wi = att.in_proj_weight.detach()
        Wq, Wk, Wv = Wi[:D], Wi[D: 2 * D], Wi[2 * D:]
        gWi, gbi = _gacc(att.in_proj_weight), _gacc(att.in_proj_bias)
        d_out = d_out.float().contiguous()
        # ---- projection, final LN, LN2
        if proj is not None:
            _wgrad(d_out, x3, _gacc(proj.weight), _gacc(proj.bias))
            dx3 = _dgrad(d_out, proj.weight.detach())
        else:
            dx3 = ops.Slices(d_out.unsqueeze(0))
        dx2 = ops.rt_ln_bwd(dx3, None, xh3, fin.weight.detach(), rs3, _gacc(fin.weight), _gacc(fin.bias))
        # z2 = x1 + k2 * (W2 f + b2): dy2 flows to x1, dl2 = dy2 * k2 to the FFN output
        dy2, dl2 = ops.rt_ln_bwd(ops.Slices(dx2.unsqueeze(0)), None, xh2, layer.norm2.weight.detach(), rs2, _gacc(layer.norm2.weight),
                                 _gacc(layer.norm2.bias), want_masked=True, drop_p=pd, drop_seed=s2)
        # ---- FFN (f is saved masked: f = kf * gelu(u))
        _wgrad(dl2, f, _gacc(layer.linear2.weight), _gacc(layer.linear2.bias))
        dfs = _dgrad(dl2, layer.linear2.weight.detach())
        du = ops.rt_elem(dfs, 2, u=u, drop_p=pd, drop_seed=sf)
        _wgrad(du, x1, _gacc(layer.linear1.weight), _gacc(layer.linear1.bias))
        dx1s = _dgrad(du, layer.linear1.weight.detach())
        # ---- LN1 over (cls + k1 * attn): incoming = FFN path + residual path; d attn = dy1 * k1, d cls += colsum(dy1)
        dy1, dattn = ops.rt_ln_bwd(dx1s, dy2, xh1, layer.norm1.weight.detach(), rs1, _gacc(layer.norm1.weight), _gacc(layer.norm1.bias),
                                   want_masked=True, drop_p=pd, drop_seed=s1)
        d_x0 = torch.empty(1, D, device=dev, dtype=torch.float32)
        ops.colsum(dy1, D, B, D, d_x0)
        _wgrad(dattn, cx, _gacc(att.out_proj.weight), _gacc(att.out_proj.bias))
        dcx = ops.rt_elem(_dgrad(dattn, att.out_proj.weight.detach()), 0)
        # ---- value projection: Wv_h += dcx_h^T m_h, bv_h += sum_b dcx_h psum, dm_h = dcx_h Wv_h
        ops.rt_gemm(dcx, m, dh, D, B, a_kmajor=True, b_kmajor=True, lda=D, ldb=H * D, nbatch=H, a_z=dh, b_z=D, out=gWi[2 * D:], ldc=D,
                    c_z=dh * D, beta=1.0, gb=gbi[2 * D:] if psum is None else None, gb_z=dh)
        # train mode: the bias path through sum_s(p mult) - bv's gradient and the extra term of the attention-weight gradient
        cbias = ops.rt_value_bias_bwd(dcx, att.in_proj_bias.detach()[2 * D:], psum, gbi[2 * D:], H) if psum is not None else None
        dm = torch.empty(B, H, D, device=dev, dtype=torch.float32)
        ops.rt_gemm(dcx, Wv, B, D, dh, nbatch=H, lda=D, a_z=dh, b_kmajor=True, ldb=D, b_z=dh * D, out=dm, ldc=H * D, c_z=D)
        # ---- attention pooling backward (two sweeps over X), gradient of the CLS slot and of a
        dp = ops.cls_scores(src, dm, True, B, R, D, H)
        dX, da_part = ops.cls_pool_bwd(src, p, dp, dm, a, lens, B, R, D, H, mult, cbias=cbias)    # (dp + cbias) * mult inside
        # The rest splits in two independent halves: the sweep over the 13 hidden states that needs dX (weighted-sum logits, unfrozen
        # layers) and ~10 small launches on parameter-sized tensors (CLS slot, a, q).  The small half goes to a second stream and
        # runs beside the sweep; the main stream waits for it before this node returns (autograd accumulates d_cls there).
        d_a = torch.empty(H, D, device=dev, dtype=torch.float32)
        dQm = torch.empty(H, D, device=dev, dtype=torch.float32)
        dq = torch.empty(1, D, device=dev, dtype=torch.float32)
        hd = ctx.handle
        if hd is not None:
            hd.check_fresh()
        main = torch.cuda.current_stream()
        side = _aux_stream(dev) if (hd is not None and ctx.needs_input_grad[1] and _AUX_ON) else None

        def small_half():
            ops.colsum(dX, R * D, B, D, d_x0, beta=1.0)             # row 0 of every utterance is the CLS token
            ops.colsum(da_part, H * D, B, H * D, d_a)
            # ---- a = s Qm Wk ;  q = Wq cls + bq   (bk receives exactly zero)
            s = dh ** -0.5
            ops.sgemm_ex(Qm, (1, D, 0), d_a, (1, D, 0), gWi[D: 2 * D], D, D, D, H, alpha=s, beta=1.0)
            ops.sgemm_ex(d_a, (D, 1, 0), Wk, (D, 1, 0), dQm, D, H, D, D, alpha=s)
            ops.headmask(dq, dQm, H, D, dh, gather=True)
            ops.sgemm_ex(dq, (1, 1, 0), x0, (1, 1, 0), gWi[:D], D, D, D, 1, beta=1.0)
            ops.colsum(dq, D, 1, D, gbi[:D], beta=1.0)
            ops.sgemm_ex(dq, (D, 1, 0), Wq, (1, D, 0), d_x0, D, 1, D, D, beta=1.0)

        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                small_half()
        else:
            small_half()
        d_cls = d_x0.reshape(1, 1, D)
        d_ws, d_feat = None, None
        if hd is not None and ctx.needs_input_grad[1]:
            d_ws = ops.wsum_bwd_logits(hd.hidden, dX, hd.w_soft, B, R, D, 1, normalize=hd.normalize, lazy=hd.lazy, seg=hd.seg)
        if hd is not None and hd.layers_bwd is not None:           # unfrozen HuBERT layers: continue the chain below the weighted sum
            hd.layers_bwd(dX, hd.w_soft)
        if hd is not None:
            hd.release()                     # nothing after this reads the encoder plan's resident states (speech_encoder._claim)
        if ctx.feat_meta is not None and ctx.needs_input_grad[2]:
            shape, dtype = ctx.feat_meta
            d_feat = dX[:, 1: 1 + shape[1]].to(dtype)
        if side is not None:
            main.wait_stream(side)
        return d_cls, d_ws, d_feat, None, None, None, None, None, None, None


class UnitRowsFn(torch.autograd.Function):
    """e = x / |x| over the last dimension of a [B, E] fp32 matrix (avssl/model/kwClip.py:857,913-915) on the row kernels: one launch
    forward (none when the producer already computed it, ParallelHeadFn), one backward - instead of norm / div and their autograd chain."""

    @staticmethod
    def forward(ctx, x):
        cached = getattr(x, "_sc_unit", None)
        if cached is not None:
            e, rn = cached
        else:
            _, e, rn = ops.rt_l2norm_fwd(ops.Slices(x.detach().float().contiguous().unsqueeze(0)), None, keep_x=False)
        ctx.save_for_backward(e, rn)
        return e

    @staticmethod
    def backward(ctx, g):
        e, rn = ctx.saved_tensors
        return ops.rt_l2norm_bwd(g, e, rn)


def unit_rows(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda or x.dim() != 2:
        raise RuntimeError("unit_rows: [B, E] device tensors only (speechclip_plus_amd has no CPU path)")
    return UnitRowsFn.apply(x)
