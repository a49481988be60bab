"""fp32 torch-CPU restatement of the weighted sum and the attention-pooling heads.

ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* weighted_sum                 avssl/module/weighted_sum.py:26-45
* transformer_encoder_forward  avssl/module/kw_modules/TransformerModels.py:48-97
                               (nn.TransformerEncoderLayer x n, post-/pre-LN, GELU(erf), then LayerNorm(1e-5))
* mha_and_norm_forward         avssl/module/kw_modules/TransformerModels.py:100-136
* parallel_branch_forward      avssl/model/kw_branches.py:251-282 (ctor :207-221; the committed ctor
                               assigns None to self_att - SURVEY F7 - the intended computation at
                               :266-280 is what is restated)

State-dict key names are those of the reference modules (``cls``, ``self_att.model.layers.0...``,
``self_att.model.norm``, ``linear_proj``).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .lengths import get_keypadding_mask


def weighted_sum(weights: torch.Tensor, hidden_states: Sequence[torch.Tensor], normalize_features: bool = False) -> torch.Tensor:
    assert len(hidden_states) == weights.numel()
    w = torch.softmax(weights, dim=0)
    x = torch.stack(list(hidden_states), dim=0)
    if normalize_features:
        x = F.layer_norm(x, (x.shape[-1],))
    return (w.view(-1, *([1] * hidden_states[0].dim())) * x).sum(0)


def _mha(W, p: str, x: torch.Tensor, kpm: Optional[torch.Tensor], nhead: int, drop=None) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True) self-attention; ``drop`` (train mode) acts on the attention probabilities."""
    B, S, D = x.shape
    dh = D // nhead
    qkv = F.linear(x, W[p + "in_proj_weight"], W[p + "in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, S, nhead, dh).transpose(1, 2) * dh ** -0.5
    k = k.view(B, S, nhead, dh).transpose(1, 2)
    v = v.view(B, S, nhead, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    if drop is not None:
        a = drop(a)
    o = (a @ v).transpose(1, 2).reshape(B, S, D)
    return F.linear(o, W[p + "out_proj.weight"], W[p + "out_proj.bias"])


def transformer_encoder_forward(W, prefix: str, src: torch.Tensor, key_padding_mask: torch.Tensor,
                                n_layers: int, nhead: int, norm_first: bool = False,
                                layer_norm_eps: float = 1e-5, return_hidden: bool = False, drop=None):
    """nn.TransformerEncoder of nn.TransformerEncoderLayer(activation=gelu) + final LayerNorm.  ``drop`` = None: eval mode.
    Train mode: ``drop(site, layer, tensor)`` at the layer's four dropout sites - "attn" (attention probabilities (B,H,S,S)),
    "dropout1" (attention block output), "dropout" (after the activation), "dropout2" (FFN output); masks are the caller's."""
    D = src.shape[-1]
    x = src
    hidden = []
    for i in range(n_layers):
        hidden.append(x)
        p = f"{prefix}model.layers.{i}."

        def ln(name, t):
            return F.layer_norm(t, (D,), W[p + name + ".weight"], W[p + name + ".bias"], layer_norm_eps)

        dr = (lambda site: (lambda t: drop(site, i, t))) if drop is not None else (lambda site: (lambda t: t))

        def ff(t):
            return dr("dropout2")(F.linear(dr("dropout")(F.gelu(F.linear(t, W[p + "linear1.weight"], W[p + "linear1.bias"]))),
                                           W[p + "linear2.weight"], W[p + "linear2.bias"]))

        d_att = dr("attn") if drop is not None else None
        if norm_first:
            x = x + dr("dropout1")(_mha(W, p + "self_attn.", ln("norm1", x), key_padding_mask, nhead, d_att))
            x = x + ff(ln("norm2", x))
        else:
            x = ln("norm1", x + dr("dropout1")(_mha(W, p + "self_attn.", x, key_padding_mask, nhead, d_att)))
            x = ln("norm2", x + ff(x))
    hidden.append(x)
    out = F.layer_norm(x, (D,), W[prefix + "model.norm.weight"], W[prefix + "model.norm.bias"], 1e-5)
    if return_hidden:
        return out, tuple(hidden)
    return out


def mha_and_norm_forward(W, prefix: str, src: torch.Tensor, key_padding_mask: torch.Tensor, nhead: int,
                         layer_norm_eps: float = 1e-5) -> torch.Tensor:
    D = src.shape[-1]
    y = _mha(W, prefix + "multihead_attn_layer.", src, key_padding_mask, nhead) + src
    return F.layer_norm(y, (D,), W[prefix + "attentionBlock_Norm.weight"], W[prefix + "attentionBlock_Norm.bias"], layer_norm_eps)


def init_parallel_branch_weights(d_model: int = 768, ffn: int = 3072, out_dim: int = 512, n_layers: int = 1,
                                 seed: int = 7123) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*shape, s):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    W: Dict[str, torch.Tensor] = {"cls": randn(1, 1, d_model, s=1.0)}
    for i in range(n_layers):
        p = f"self_att.model.layers.{i}."
        W[p + "self_attn.in_proj_weight"] = randn(3 * d_model, d_model, s=d_model ** -0.5)
        W[p + "self_attn.in_proj_bias"] = randn(3 * d_model, s=0.05)
        W[p + "self_attn.out_proj.weight"] = randn(d_model, d_model, s=d_model ** -0.5)
        W[p + "self_attn.out_proj.bias"] = randn(d_model, s=0.05)
        W[p + "linear1.weight"] = randn(ffn, d_model, s=d_model ** -0.5)
        W[p + "linear1.bias"] = randn(ffn, s=0.05)
        W[p + "linear2.weight"] = randn(d_model, ffn, s=ffn ** -0.5)
        W[p + "linear2.bias"] = randn(d_model, s=0.05)
        for n in ("norm1", "norm2"):
            W[p + n + ".weight"] = 1.0 + randn(d_model, s=0.1)
            W[p + n + ".bias"] = randn(d_model, s=0.1)
    W["self_att.model.norm.weight"] = 1.0 + randn(d_model, s=0.1)
    W["self_att.model.norm.bias"] = randn(d_model, s=0.1)
    W["linear_proj.weight"] = randn(out_dim, d_model, s=d_model ** -0.5)
    W["linear_proj.bias"] = randn(out_dim, s=0.05)
    return W


def parallel_branch_forward(W, audio_feat: torch.Tensor, audio_len: torch.Tensor, nhead: int = 8,
                            n_layers: int = 1, need_projection: bool = True, drop=None) -> torch.Tensor:
    """kw_branches.py:266-280 -> (B, E) un-normalised parallel_audio_feat (``drop``: see transformer_encoder_forward)."""
    bsz, T = audio_feat.shape[:2]
    cls = torch.cat([W["cls"]] * bsz, dim=0)
    src = torch.cat([cls, audio_feat], dim=1)
    kpm = get_keypadding_mask(T + 1, audio_len + 1)
    out = transformer_encoder_forward(W, "self_att.", src, kpm, n_layers, nhead, drop=drop)
    out = out[:, :1].reshape(-1, audio_feat.shape[-1])
    if need_projection:
        out = F.linear(out, W["linear_proj.weight"], W["linear_proj.bias"])
    return out
