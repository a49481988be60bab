"""Attention blocks of the branch heads: mirrors of avssl/module/kw_modules/TransformerModels.py:48-136.

``TransformerEncoder`` keeps the reference constructor and parameter names (it instantiates the same
``nn.TransformerEncoder`` container for initialisation / state-dict compatibility:
``model.layers.{i}.self_attn.in_proj_weight`` ... ``model.norm.weight``) but never calls torch's forward.

``cls_forward`` is the path the parallel branch needs (kw_branches.py:266-280 keeps only row 0 of the layer
output): the CLS query attends over all S keys without materialising K / V (csrc/clspool.hip), then the
one-row-per-utterance remainder of the post-LN layer (out_proj, LayerNorm, FFN, LayerNorm, final LayerNorm,
projection) runs in fp32 on the master weights in csrc/headtail.hip - see head_tail.py.
"""
import logging
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .head_tail import ParallelHeadFn

logger = logging.getLogger(__name__)

__all__ = ["TransformerEncoder", "MultiheadAttentionAndNorm"]


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
    def cls_forward(self, cls: torch.Tensor, feat: torch.Tensor, lens: torch.Tensor,
                    proj: Optional[nn.Linear] = None) -> torch.Tensor:
        """Row 0 of ``self.model(cat([cls, feat], 1), key_padding_mask(lens))``, optionally followed by ``proj`` (the branch's
        ``linear_proj``) -> (B, D) or (B, E) fp32.  One autograd node on the library's kernels (head_tail.ParallelHeadFn).

        ``lens`` = number of valid keys per utterance including the CLS slot (audio_len + 1)."""
        if self.n_layers != 1 or self.norm_first:
            raise NotImplementedError("cls_forward covers the shipped parallel-branch recipe: 1 post-LN layer")
        # train mode: the layer's four dropout sites (p = self.dropout) are applied inside ParallelHeadFn
        D = self.d_model
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
        return ParallelHeadFn.apply(cls, ws_w, feat_in, self, proj, handle, src, lens32, B, R)


class MultiheadAttentionAndNorm(nn.Module):
    def __init__(self, d_model: int = 768, nhead: int = 8, dropout: float = 0.1, layer_norm_eps: float = 1e-5,
                 batch_first: bool = True, **kwargs) -> None:
        super().__init__()
        self.multihead_attn_layer = nn.MultiheadAttention(d_model, num_heads=nhead, dropout=dropout, batch_first=batch_first)
        self.attentionBlock_Norm = nn.LayerNorm(d_model, eps=layer_norm_eps)

    def forward(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """LN(MHA(x, x, x) + x) (TransformerModels.py:120-126).  On a GPU the whole block - projections, S x S core as batched bf16
        GEMMs (any head_dim that is a multiple of 64: 768 / 128 in the shipped recipes), residual, LayerNorm, and their backward -
        runs on the library's kernels (mha_block.MhaNormFn).  Train mode applies nn.MultiheadAttention's dropout to the attention
        probabilities."""
        src = src.float()
        mha = self.multihead_attn_layer
        if not src.is_cuda:
            return self.attentionBlock_Norm(mha(src, src, src, key_padding_mask=key_padding_mask)[0] + src)
        D, H = src.shape[-1], mha.num_heads
        if D % 64 == 0 and D % H == 0 and (D // H) % 64 == 0:
            from .mha_block import mha_norm                  # whole block (projections, S x S core, residual, LayerNorm) on own kernels
            return mha_norm(src, mha, self.attentionBlock_Norm, key_padding_mask, self.training)
        return self._forward_stock_core(src, key_padding_mask)

    def _forward_stock_core(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """head_dim not a multiple of 64: projections on the library's GEMM, fp32 torch bmm core."""
        mha = self.multihead_attn_layer
        from .linear_fn import linear_bf16_autograd
        B, S, D = src.shape
        H = mha.num_heads
        dh = D // H
        qkv = linear_bf16_autograd(src, mha.in_proj_weight, mha.in_proj_bias)                  # (B, S, 3D)
        q, k, v = (t.reshape(B, S, H, dh).transpose(1, 2).reshape(B * H, S, dh).contiguous() for t in qkv.split(D, dim=-1))
        scores = torch.bmm(q, k.transpose(1, 2)).view(B, H, S, S) * dh ** -0.5
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        p = torch.softmax(scores, dim=-1).view(B * H, S, S)
        if self.training and mha.dropout > 0:
            p = torch.nn.functional.dropout(p, mha.dropout)
        ctx = torch.bmm(p, v).view(B, H, S, dh).transpose(1, 2).reshape(B, S, D)
        out = linear_bf16_autograd(ctx, mha.out_proj.weight, mha.out_proj.bias)
        return self.attentionBlock_Norm(out + src)

    def extract_hidden_states(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        return tuple([src, self.forward(src, key_padding_mask)])
