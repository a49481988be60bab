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
from typing import Optional, Tuple

import torch
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

    def _layer(self, mod: nn.TransformerEncoderLayer, x: torch.Tensor, key_padding_mask: torch.Tensor) -> torch.Tensor:
        """One post-LN nn.TransformerEncoderLayer on the library's kernels: norm1(x + drop(MHA(x))) -> norm2(. + drop(FFN(.)))
        (mha_block.MhaNormFn / FfnNormFn; any head_dim).  Train mode runs the layer's three dropout sites."""
        from .mha_block import FfnNormFn, _next_seed, mha_norm
        p = float(self.dropout) if self.training else 0.0
        x = mha_norm(x, mod.self_attn, mod.norm1, key_padding_mask, self.training, p_res=p)
        return FfnNormFn.apply(x, mod.linear1.weight, mod.linear1.bias, mod.linear2.weight, mod.linear2.bias, mod.norm2.weight,
                               mod.norm2.bias, mod.norm2.eps, p, _next_seed() if p > 0 else 0, _next_seed() if p > 0 else 0)

    def _check(self, src: torch.Tensor) -> None:
        if not src.is_cuda:
            raise RuntimeError("TransformerEncoder runs on the HIP kernels: device tensors only (CPU restatement: oracle/head_ref.py)")
        if self.norm_first:
            raise NotImplementedError("norm_first = True is not used by any shipped config")

    def forward(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """Full-sequence path (TransformerModels.py:73-83): every layer, then the final LayerNorm."""
        from .mha_block import LayerNormFn
        self._check(src)
        output = src.float()
        for mod in self.model.layers:
            output = self._layer(mod, output, key_padding_mask)
        norm = self.model.norm
        return LayerNormFn.apply(output, norm.weight, norm.bias, norm.eps)

    def extract_hidden_states(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        """TransformerModels.py:16-45,85-97: inputs of every layer + output of the last, before the final norm."""
        self._check(src)
        output, hidden = src.float(), []
        for mod in self.model.layers:
            hidden.append(output)
            output = self._layer(mod, output, key_padding_mask)
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
        """LN(MHA(x, x, x) + x) (TransformerModels.py:120-126): the whole block - projections, S x S core as batched bf16 GEMMs (any
        head_dim: 768 / 128 / 96 in the shipped recipes), residual, LayerNorm, and their backward - runs on the library's
        kernels (mha_block.MhaNormFn).  Train mode applies nn.MultiheadAttention's dropout to the attention probabilities."""
        if not src.is_cuda:
            raise RuntimeError("MultiheadAttentionAndNorm runs on the HIP kernels: device tensors only (CPU restatement: "
                               "oracle/head_ref.py)")
        from .mha_block import mha_norm
        # (no cast of the input: the block rounds its operands to bf16 itself and reads a view of the encoder's output rows in place;
        # the result leaves in fp32 like the reference's)
        return mha_norm(src, self.multihead_attn_layer, self.attentionBlock_Norm, key_padding_mask, self.training,
                        out_dtype=torch.float32)

    def forward_rows(self, src: torch.Tensor, key_padding_mask: torch.Tensor, n_cls: int = 0):
        """The same block for a consumer that reads its output rows in place (the CIF module of the cascaded+/hybrid+ branches):
        -> (mha_block.BranchRows: the bf16 output rows, frames of every utterance behind its ``n_cls`` leading rows; the CLS rows as
        fp32 [B, D] or None).  Training path only: forward() is the module's public form."""
        from .mha_block import mha_norm
        return mha_norm(src, self.multihead_attn_layer, self.attentionBlock_Norm, key_padding_mask, self.training, rows_out=int(n_cls))

    def extract_hidden_states(self, src: torch.Tensor, key_padding_mask: torch.Tensor):
        return tuple([src, self.forward(src, key_padding_mask)])
