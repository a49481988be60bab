"""Attention blocks of the branch heads: mirrors of avssl/module/kw_modules/TransformerModels.py:48-136.

``TransformerEncoder`` keeps the reference constructor and parameter names (it instantiates the same
``nn.TransformerEncoder`` container for initialisation / state-dict compatibility:
``model.layers.{i}.self_attn.in_proj_weight`` ... ``model.norm.weight``) but never calls torch's forward.

``cls_forward`` is the path the parallel branch needs (kw_branches.py:266-280 keeps only row 0 of the layer
output): the CLS query attends over all S keys without materialising K / V (csrc/clspool.hip), then the
one-row-per-utterance remainder of the post-LN layer (out_proj, LayerNorm, FFN, LayerNorm, final LayerNorm)
runs in fp32 on the master weights.
"""
import logging
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops

logger = logging.getLogger(__name__)

__all__ = ["TransformerEncoder", "MultiheadAttentionAndNorm"]


class _ClsAttnPoolFn(torch.autograd.Function):
    """m[b,h,:] = sum_s softmax_s(a_h . X[b,s])[s] X[b,s,:]   with X row 0 = cls.

    inputs : cls [1,1,D] fp32, a [H,D] fp32, ws_weights (or None), feat (generic path) ; handle / lens as consts
    grads  : d cls (through X row 0), d a, d ws_weights (handle path) or d feat (generic path)
    """

    @staticmethod
    def forward(ctx, cls, a, ws_weights, feat, handle, src, lens, B, R, D, H):
        src = src.detach()                                          # shares storage with the encoder's buffer
        src[:, 0] = cls.detach().reshape(1, D).to(torch.bfloat16)
        a_c = a.detach().float().contiguous()
        scores = ops.cls_scores(src, a_c, False, B, R, D, H)
        p, m = ops.cls_pool_fwd(src, scores, lens, B, R, D, H)
        ctx.save_for_backward(src, p, a_c, lens)
        ctx.handle, ctx.dims = handle, (B, R, D, H)
        ctx.feat_meta = None if feat is None else (feat.shape, feat.dtype)
        return m

    @staticmethod
    def backward(ctx, dm):
        src, p, a_c, lens = ctx.saved_tensors
        B, R, D, H = ctx.dims
        dm = dm.float().contiguous()
        dp = ops.cls_scores(src, dm, True, B, R, D, H)
        dX, da_part = ops.cls_pool_bwd(src, p, dp, dm, a_c, lens, B, R, D, H)
        d_cls = dX[:, 0].sum(0).reshape(1, 1, D)
        d_a = da_part.sum(0)
        d_ws, d_feat = None, None
        hd = ctx.handle
        if hd is not None and ctx.needs_input_grad[2]:
            d_soft = ops.wsum_bwd(hd.hidden, dX, B, R, D, 1, normalize=hd.normalize)
            d_ws = hd.w_soft * (d_soft - (hd.w_soft * d_soft).sum())
        if ctx.feat_meta is not None and ctx.needs_input_grad[3]:
            shape, dtype = ctx.feat_meta
            d_feat = dX[:, 1: 1 + shape[1]].to(dtype)
        return d_cls, d_a, d_ws, d_feat, None, None, None, None, None, None, None


class TransformerEncoder(nn.Module):
    def __init__(self, n_layers: int = 1, d_model: int = 768, nhead: int = 8, dim_feedforward: int = 3072,
                 dropout: float = 0.1, activation: str = "gelu", layer_norm_eps: float = 1e-5, batch_first: bool = True,
                 norm_first: bool = False, **kwargs) -> None:
        super().__init__()
        logger.info(f"Using {n_layers} layer transformer encoder")
        assert activation == "gelu" and batch_first, "only the reference's gelu / batch_first configuration is built"
        encoder_layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward,
                                                   dropout=dropout, activation=activation, layer_norm_eps=layer_norm_eps,
                                                   batch_first=batch_first, norm_first=norm_first)
        encoder_norm = nn.LayerNorm(d_model, eps=1e-5)
        self.model = nn.TransformerEncoder(encoder_layer, n_layers, encoder_norm, enable_nested_tensor=False)
        self.n_layers, self.d_model, self.nhead = n_layers, d_model, nhead
        self.norm_first, self.layer_norm_eps, self.dropout = norm_first, layer_norm_eps, dropout

    def forward(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """Full-sequence path (every row is needed by the cascaded+/hybrid+ branches that feed CIF; scope row a11:
        stock device-side torch ops until the head_dim 96/128 attention kernel of row f3 exists)."""
        return self.model(src=src.float(), src_key_padding_mask=key_padding_mask)

    def extract_hidden_states(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """TransformerModels.py:16-45,85-97: inputs of every layer + output of the last, before the final norm."""
        output, hidden = src.float(), []
        for mod in self.model.layers:
            hidden.append(output)
            output = mod(output, src_key_padding_mask=key_padding_mask)
        hidden.append(output)
        return tuple(hidden)

    # ------------------------------------------------------------------------------------------------
    def cls_forward(self, cls: torch.Tensor, feat: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """Row 0 of ``self.model(cat([cls, feat], 1), key_padding_mask(lens))`` -> (B, D) fp32.

        ``lens`` = number of valid keys per utterance including the CLS slot (audio_len + 1)."""
        if self.n_layers != 1 or self.norm_first:
            raise NotImplementedError("cls_forward covers the shipped parallel-branch recipe: 1 post-LN layer")
        if self.training and self.dropout > 0:
            # the reference applies dropout(p=0.1) here in train mode; this build runs the head deterministically
            pass
        layer = self.model.layers[0]
        D, H = self.d_model, self.nhead
        dh = D // H
        handle = getattr(feat, "_sc_handle", None)
        if handle is not None:
            src, B, R = handle.src.detach(), handle.B, handle.R
            ws_w = handle.ws_layer.weights
            feat_in = None
        else:
            B, T = feat.shape[:2]
            R = (T + 1 + 127) // 128 * 128
            src = torch.zeros(B, R, D, device=feat.device, dtype=torch.bfloat16)
            src[:, 1: T + 1] = feat.detach().to(torch.bfloat16)
            ws_w, feat_in = None, feat
        lens32 = lens.to(device=src.device, dtype=torch.int32).contiguous()
        Wq, Wk, Wv = layer.self_attn.in_proj_weight.split(D, dim=0)
        bq, _bk, bv = layer.self_attn.in_proj_bias.split(D, dim=0)   # bk shifts every score equally: cancels
        x0 = cls.reshape(1, D).float()
        q = (F.linear(x0, Wq, bq) * dh ** -0.5).reshape(H, dh)                       # CLS query, same for every b
        a = torch.einsum("hd,hdk->hk", q, Wk.reshape(H, dh, D))                      # a_h = Wk_h^T q_h
        m = _ClsAttnPoolFn.apply(cls, a, ws_w, feat_in, handle, src, lens32, B, R, D, H)     # (B, H, D)
        ctx_v = torch.einsum("bhk,hdk->bhd", m, Wv.reshape(H, dh, D)).reshape(B, D) + bv      # softmax sums to 1
        attn_out = F.linear(ctx_v, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
        x = F.layer_norm(x0 + attn_out, (D,), layer.norm1.weight, layer.norm1.bias, self.layer_norm_eps)
        ff = F.linear(F.gelu(F.linear(x, layer.linear1.weight, layer.linear1.bias)), layer.linear2.weight, layer.linear2.bias)
        x = F.layer_norm(x + ff, (D,), layer.norm2.weight, layer.norm2.bias, self.layer_norm_eps)
        return F.layer_norm(x, (D,), self.model.norm.weight, self.model.norm.bias, 1e-5)


class MultiheadAttentionAndNorm(nn.Module):
    def __init__(self, d_model: int = 768, nhead: int = 8, dropout: float = 0.1, layer_norm_eps: float = 1e-5,
                 batch_first: bool = True, **kwargs) -> None:
        super().__init__()
        self.multihead_attn_layer = nn.MultiheadAttention(d_model, num_heads=nhead, dropout=dropout, batch_first=batch_first)
        self.attentionBlock_Norm = nn.LayerNorm(d_model, eps=layer_norm_eps)

    def forward(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """LN(MHA(x, x, x) + x) (TransformerModels.py:120-126); scope row a11: stock device-side torch ops."""
        src = src.float()
        return self.attentionBlock_Norm(
            self.multihead_attn_layer(src, src, src, key_padding_mask=key_padding_mask)[0] + src)

    def extract_hidden_states(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        return tuple([src, self.forward(src, key_padding_mask)])
