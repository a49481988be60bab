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
    """y[B,N] = x[B,K] W[N,K]^T + b"""
    Bn, K = x.shape
    N = W.shape[0]
    y = torch.empty(Bn, N, device=x.device, dtype=torch.float32)
    ops.sgemm_ex(x, (K, 1, 0), W, (K, 1, 0), y, N, Bn, N, K, bias=b)
    return y


def _lin_bwd(dy, x, W, gW, gb, need_dx=True):
    """gW += dy^T x ; gb += colsum(dy) ; returns dx = dy W"""
    Bn, N = dy.shape
    K = W.shape[1]
    ops.sgemm_ex(dy, (1, N, 0), x, (1, K, 0), gW, K, N, K, Bn, beta=1.0)
    if gb is not None:
        ops.colsum(dy, N, Bn, N, gb, beta=1.0)
    if not need_dx:
        return None
    dx = torch.empty(Bn, K, device=dy.device, dtype=torch.float32)
    ops.sgemm_ex(dy, (N, 1, 0), W, (1, K, 0), dx, K, Bn, K, N)
    return dx


class ParallelHeadFn(torch.autograd.Function):
    """inputs : cls [1,1,D] fp32 (parameter), ws_weights (weighted-sum logits or None), feat (generic path or None),
               then constants: the TransformerEncoder module, the projection nn.Linear (or None), the encoder handle,
               src [B,R,D] bf16 (row 0 = CLS slot), lens int32 [B], B, R
       output : [B, E] (or [B, D] without projection) fp32"""

    @staticmethod
    def forward(ctx, cls, ws_weights, feat, module, proj, handle, src, lens, B, R):
        layer = module.model.layers[0]
        D, H = module.d_model, module.nhead
        dh = D // H
        dev = src.device
        att = layer.self_attn
        Wi, bi = att.in_proj_weight.detach(), att.in_proj_bias.detach()
        Wq, Wk, Wv = Wi[:D], Wi[D: 2 * D], Wi[2 * D:]
        x0 = cls.detach().reshape(1, D).float().contiguous()
        src = src.detach()
        src[:, 0] = x0.to(torch.bfloat16)                       # CLS slot of the padded [CLS ; frames] buffer
        # ---- CLS query folded into the key projection: a_h = dh^-1/2 Wk_h^T q_h (bk shifts all scores alike: cancels)
        q = _lin(x0, Wq, bi[:D])                                  # [1, D]
        Qm = torch.empty(H, D, device=dev, dtype=torch.float32)
        ops.headmask(q, Qm, H, D, dh, gather=False)
        a = torch.empty(H, D, device=dev, dtype=torch.float32)
        ops.sgemm_ex(Qm, (D, 1, 0), Wk, (1, D, 0), a, D, H, D, D, alpha=dh ** -0.5)
        scores = ops.cls_scores(src, a, False, B, R, D, H)
        pd = float(module.dropout) if module.training else 0.0
        mk = (lambda *shape: ops.dropout_mult(shape, pd, dev)) if pd > 0 else None      # one launch per mask (was rand / compare / cast / scale)
        mult = mk(B, H, R) if mk else None
        p, m = ops.cls_pool_fwd(src, scores, lens, B, R, D, H, mult)   # m [B, H, D]
        # ---- value projection per head (softmax sums to 1 => + bv), out_proj, post-LN layer on B rows
        cx = torch.empty(B, D, device=dev, dtype=torch.float32)
        if mult is None:
            ops.sgemm_ex(m, (H * D, 1, D), Wv, (D, 1, dh * D), cx, D, B, dh, D, nbatch=H, scz=dh, bias=bi[2 * D:], sbiasz=dh)
            psum = None
        else:       # dropped attention weights no longer sum to 1: ctx_h = Wv_h m_h + bv_h * sum_s(p mult)
            ops.sgemm_ex(m, (H * D, 1, D), Wv, (D, 1, dh * D), cx, D, B, dh, D, nbatch=H, scz=dh)
            psum = (p * mult).sum(-1)                               # [B, H]
            cx.view(B, H, dh).addcmul_(psum[:, :, None], bi[2 * D:].view(1, H, dh))
        attn = _lin(cx, att.out_proj.weight.detach(), att.out_proj.bias.detach())
        k1, kf, k2 = (mk(B, D), mk(B, layer.linear1.out_features), mk(B, D)) if mk else (None, None, None)
        if mk:
            attn *= k1
        x1, xh1, rs1 = ops.rowln_fwd(attn, x0, 0, layer.norm1.weight.detach(), layer.norm1.bias.detach(), module.layer_norm_eps)
        u = _lin(x1, layer.linear1.weight.detach(), layer.linear1.bias.detach())
        f = ops.gelu_f32(u)
        if mk:
            f *= kf
        y2 = _lin(f, layer.linear2.weight.detach(), layer.linear2.bias.detach())
        if mk:
            y2 *= k2
        x2, xh2, rs2 = ops.rowln_fwd(y2, x1, D, layer.norm2.weight.detach(), layer.norm2.bias.detach(), module.layer_norm_eps)
        fin = module.model.norm
        x3, xh3, rs3 = ops.rowln_fwd(x2, None, 0, fin.weight.detach(), fin.bias.detach(), fin.eps)
        out = _lin(x3, proj.weight.detach(), proj.bias.detach()) if proj is not None else x3
        ctx.mod, ctx.proj, ctx.handle, ctx.dims = module, proj, handle, (B, R, D, H)
        ctx.feat_meta = None if feat is None else (feat.shape, feat.dtype)
        ctx.masks = (mult, k1, kf, k2, psum)
        ctx.save_for_backward(src, lens, x0, q, Qm, a, p, m, cx, xh1, rs1, x1, u, f, xh2, rs2, xh3, rs3, x3)
        return out

    @staticmethod
    def backward(ctx, d_out):
        src, lens, x0, q, Qm, a, p, m, cx, xh1, rs1, x1, u, f, xh2, rs2, xh3, rs3, x3 = ctx.saved_tensors
        module, proj = ctx.mod, ctx.proj
        B, R, D, H = ctx.dims
        mult, k1, kf, k2, psum = ctx.masks
        dh = D // H
        dev = src.device
        layer, fin = module.model.layers[0], module.model.norm
        att = layer.self_attn
        Wi = att.in_proj_weight.detach()
        Wq, Wk, Wv = Wi[:D], Wi[D: 2 * D], Wi[2 * D:]
        gWi, gbi = _gacc(att.in_proj_weight), _gacc(att.in_proj_bias)
        d_out = d_out.float().contiguous()
        # ---- projection, final LN, LN2
        dx3 = _lin_bwd(d_out, x3, proj.weight.detach(), _gacc(proj.weight), _gacc(proj.bias)) if proj is not None else d_out
        dx2 = ops.rowln_bwd(dx3, xh3, fin.weight.detach(), rs3, _gacc(fin.weight), _gacc(fin.bias))
        dy2 = ops.rowln_bwd(dx2, xh2, layer.norm2.weight.detach(), rs2, _gacc(layer.norm2.weight), _gacc(layer.norm2.bias))
        # ---- FFN:  y2 = x1 + k2 * (W2 (kf * gelu(W1 x1 + b1)) + b2)      (k* = dropout multipliers, 1 in eval; f is saved masked)
        dl2 = dy2 * k2 if k2 is not None else dy2
        df = _lin_bwd(dl2, f, layer.linear2.weight.detach(), _gacc(layer.linear2.weight), _gacc(layer.linear2.bias))
        if kf is not None:
            df *= kf
        du = ops.gelu_f32(u, df)
        dx1 = _lin_bwd(du, x1, layer.linear1.weight.detach(), _gacc(layer.linear1.weight), _gacc(layer.linear1.bias))
        dx1 += dy2
        # ---- LN1 over (cls + attn): d attn = dy1, d cls += colsum(dy1)
        dy1 = ops.rowln_bwd(dx1, xh1, layer.norm1.weight.detach(), rs1, _gacc(layer.norm1.weight), _gacc(layer.norm1.bias))
        d_x0 = torch.empty(1, D, device=dev, dtype=torch.float32)
        ops.colsum(dy1, D, B, D, d_x0)
        if k1 is not None:
            dy1 = dy1 * k1
        dcx = _lin_bwd(dy1, cx, att.out_proj.weight.detach(), _gacc(att.out_proj.weight), _gacc(att.out_proj.bias))
        # ---- value projection: bv, Wv_h += dcx_h^T m_h, dm_h = dcx_h Wv_h
        if psum is None:
            ops.colsum(dcx, D, B, D, gbi[2 * D:], beta=1.0)
        else:
            gbi[2 * D:].view(H, dh).add_((dcx.view(B, H, dh) * psum[:, :, None]).sum(0))
        ops.sgemm_ex(dcx, (1, D, dh), m, (1, H * D, D), gWi[2 * D:], D, dh, D, B, nbatch=H, scz=dh * D, beta=1.0)
        dm = torch.empty(B, H, D, device=dev, dtype=torch.float32)
        ops.sgemm_ex(dcx, (D, 1, dh), Wv, (1, D, dh * D), dm, H * D, B, D, dh, nbatch=H, scz=D)
        # ---- attention pooling backward (two sweeps over X), gradient of the CLS slot and of a
        dp = ops.cls_scores(src, dm, True, B, R, D, H)
        if mult is not None:                                        # + the bias path through sum_s(p mult), then the mask itself
            dp += (dcx.view(B, H, dh) * att.in_proj_bias.detach()[2 * D:].view(1, H, dh)).sum(-1)[:, :, None]
            dp *= mult
        dX, da_part = ops.cls_pool_bwd(src, p, dp, dm, a, lens, B, R, D, H, mult)
        ops.colsum(dX, R * D, B, D, d_x0, beta=1.0)              # row 0 of every utterance is the CLS token
        d_a = torch.empty(H, D, device=dev, dtype=torch.float32)
        ops.colsum(da_part, H * D, B, H * D, d_a)
        # ---- a = s Qm Wk ;  q = Wq cls + bq   (bk receives exactly zero)
        s = dh ** -0.5
        ops.sgemm_ex(Qm, (1, D, 0), d_a, (1, D, 0), gWi[D: 2 * D], D, D, D, H, alpha=s, beta=1.0)
        dQm = torch.empty(H, D, device=dev, dtype=torch.float32)
        ops.sgemm_ex(d_a, (D, 1, 0), Wk, (D, 1, 0), dQm, D, H, D, D, alpha=s)
        dq = torch.empty(1, D, device=dev, dtype=torch.float32)
        ops.headmask(dq, dQm, H, D, dh, gather=True)
        ops.sgemm_ex(dq, (1, 1, 0), x0, (1, 1, 0), gWi[:D], D, D, D, 1, beta=1.0)
        ops.colsum(dq, D, 1, D, gbi[:D], beta=1.0)
        ops.sgemm_ex(dq, (D, 1, 0), Wq, (1, D, 0), d_x0, D, 1, D, D, beta=1.0)
        d_cls = d_x0.reshape(1, 1, D)
        d_ws, d_feat = None, None
        hd = ctx.handle
        if hd is not None:
            hd.check_fresh()
        if hd is not None and ctx.needs_input_grad[1]:
            d_soft = ops.wsum_bwd(hd.hidden, dX, B, R, D, 1, normalize=hd.normalize, lazy=hd.lazy)
            d_ws = hd.w_soft * (d_soft - (hd.w_soft * d_soft).sum())
        if hd is not None and hd.layers_bwd is not None:           # unfrozen HuBERT layers: continue the chain below the weighted sum
            hd.layers_bwd(dX, hd.w_soft)
        if ctx.feat_meta is not None and ctx.needs_input_grad[2]:
            shape, dtype = ctx.feat_meta
            d_feat = dX[:, 1: 1 + shape[1]].to(dtype)
        return d_cls, d_ws, d_feat, None, None, None, None, None, None, None
