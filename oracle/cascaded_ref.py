"""fp32 torch-CPU restatement of the cascaded+ / hybrid+ branch tails.

ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional style over a weight dict whose keys are the
reference's state-dict names (``self_att.multihead_attn_layer.in_proj_weight``, ``downsampling.conv.0.weight``,
``linear_proj.weight``, ``bn_layer.bn_layer.weight``, ``clip.model.token_embedding.weight`` ...).

* cif_forward / integrate_and_fire   avssl/module/cif.py:97-311          (pinned: tests/golden/cif_d32.npz)
* vq_forward                         my_vector_quantizer.py:64-165        (pinned: vq_v50.npz)
* batchnorm over keywords            kw_bn.py:167-228                     (pinned: kwbn_e16.npz)
* keyword cosine / vq_audio_features kw_branches.py:143-197
* clip_encode_keywords               clip_official.py:222-279 over openai/CLIP's text transformer
                                     (third party, unpinned, absent offline: restated from its published architecture)
* cascaded_plus_forward / hybrid_plus_forward   kw_branches.py:701-753, 808-866
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .head_ref import mha_and_norm_forward
from .lengths import get_keypadding_mask

MAX_FEAT_LEN = 75


def integrate_and_fire(x: torch.Tensor, alpha: torch.Tensor, thr: float, target_given: bool, tail_thr: float = 0.5):
    B, S, C = x.shape
    lengths = (alpha.sum(1) / thr).floor().clip(min=1, max=MAX_FEAT_LEN).long()
    T = int(lengths.max())
    csum = alpha.cumsum(-1)
    right = (csum / thr).floor().long().clip(0, T).detach()
    left = right.roll(1, dims=1)
    left[:, 0] = 0
    fires = right - left
    extra = (fires - 1).clip(min=0)
    out = x.new_zeros(B, T + 1, C)
    rw = torch.where(fires > 0, csum - right.type_as(alpha) * thr, alpha.new_zeros(1))
    out = out.scatter_add(1, right.unsqueeze(-1).expand(-1, -1, C), rw.unsqueeze(-1) * x)
    lw = alpha - rw - extra.type_as(alpha) * thr
    out = out.scatter_add(1, left.unsqueeze(-1).expand(-1, -1, C), lw.unsqueeze(-1) * x)
    tgt = left
    for _ in range(int(extra.max())):
        tgt = (tgt + 1).clip(max=T)
        out = out.scatter_add(1, tgt.unsqueeze(-1).expand(-1, -1, C), x * thr * (extra > 0).unsqueeze(2))
        extra = extra - 1
    if target_given:
        out = out[:, :T]
    else:
        tail = torch.where(right == lengths.unsqueeze(1), rw, rw.new_zeros(1)).sum(-1)
        tail = tail + torch.where(left == lengths.unsqueeze(1), lw, lw.new_zeros(1)).sum(-1)
        extend = tail >= tail_thr
        if extend.any():
            factor = (thr / tail.masked_fill(~extend, thr)).view(B, 1, 1).expand(-1, -1, C)
            up = torch.ones_like(out).scatter(1, lengths.view(B, 1, 1).expand(-1, -1, C), factor).detach()
            out = out * up
            lengths = (lengths + extend.long()).clip(max=MAX_FEAT_LEN)
            T = int(lengths.max())
        out = out[:, :T]
        out = out.masked_fill((torch.arange(T).unsqueeze(0) >= lengths.unsqueeze(1)).unsqueeze(-1), 0)
    return out, lengths


def cif_forward(W: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, pad: torch.Tensor,
                target_len: Optional[torch.Tensor], thr: float = 1.0, apply_scaling: bool = True, eps: float = 1e-5):
    """eval-mode weight generator (the reference's p=0.5 dropouts are identity in eval)."""
    h = F.relu(F.conv1d(x.permute(0, 2, 1), W[prefix + "conv.0.weight"], W[prefix + "conv.0.bias"],
                        padding=W[prefix + "conv.0.weight"].shape[-1] // 2)).permute(0, 2, 1)
    alpha = torch.sigmoid(F.linear(h, W[prefix + "weight_proj.1.weight"], W[prefix + "weight_proj.1.bias"]))
    alpha = alpha.clip(0.0, 1.0).float().squeeze(-1).masked_fill(pad, 0.0)
    quantity = alpha.sum(1)
    if apply_scaling and target_len is not None:
        alpha = alpha * ((thr * target_len.type_as(alpha) + eps) / quantity).unsqueeze(1)
    feats, lengths = integrate_and_fire(x, alpha, thr, target_len is not None)
    return feats, lengths, quantity


def vq_forward(scores: torch.Tensor, temp: float, training: bool, prob_msk=(0, 2, 3), forced: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``forced`` (test aid, not in the reference): token indices to use instead of the argmax - lets a parity test share the
    DISCRETE choice between two implementations where the reference's own top-2 margin is a near-tie, and compare everything
    continuous downstream."""
    s = scores.clone()
    s[..., list(prob_msk)] = float("-inf")
    hard = F.one_hot(s.argmax(-1) if forced is None else forced, s.shape[-1]).type_as(s)
    if not training:
        return hard
    soft = torch.softmax(s / temp, dim=-1)
    return hard + soft - soft.detach()


def clip_text_transformer(W: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """openai/CLIP ``Transformer`` of the text tower (pre-LN residual blocks, causal mask, QuickGELU): (B, 77, W) -> (B, 77, W)."""
    B, S, Wd = x.shape
    mask = torch.full((S, S), float("-inf")).triu_(1)
    layers = 1 + max(int(k.split(".")[len(prefix.split(".")) + 1]) for k in W if k.startswith(prefix + "transformer.resblocks."))
    dh = Wd // heads
    for i in range(layers):
        p = f"{prefix}transformer.resblocks.{i}."
        y = F.layer_norm(x, (Wd,), W[p + "ln_1.weight"], W[p + "ln_1.bias"])
        q, k, v = F.linear(y, W[p + "attn.in_proj_weight"], W[p + "attn.in_proj_bias"]).split(Wd, dim=-1)
        q = q.view(B, S, heads, dh).transpose(1, 2) * dh ** -0.5
        k = k.view(B, S, heads, dh).transpose(1, 2)
        v = v.view(B, S, heads, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, S, Wd)
        x = x + F.linear(o, W[p + "attn.out_proj.weight"], W[p + "attn.out_proj.bias"])
        y = F.layer_norm(x, (Wd,), W[p + "ln_2.weight"], W[p + "ln_2.bias"])
        y = F.linear(y, W[p + "mlp.c_fc.weight"], W[p + "mlp.c_fc.bias"])
        x = x + F.linear(y * torch.sigmoid(1.702 * y), W[p + "mlp.c_proj.weight"], W[p + "mlp.c_proj.bias"])
    return x


def clip_encode_keywords(W: Dict[str, torch.Tensor], prefix: str, keywords: torch.Tensor, n_kw: torch.Tensor,
                         heads: int, sot: int, eot: int) -> torch.Tensor:
    B, N, Wd = keywords.shape
    emb = W[prefix + "token_embedding.weight"]
    x = emb[0].expand(B, 77, Wd).clone()                      # token id 0 everywhere ...
    x[:, 0] = emb[sot]
    for b in range(B):                                        # clip_official.py:261-263 per-row splice
        n = int(n_kw[b])
        x[b, 1: 1 + n] = keywords[b, :n]
        x[b, 1 + n] = emb[eot]
    x = x + W[prefix + "positional_embedding"]
    x = clip_text_transformer(W, prefix, x, heads)
    x = F.layer_norm(x, (Wd,), W[prefix + "ln_final.weight"], W[prefix + "ln_final.bias"])
    return x[torch.arange(B), n_kw + 1] @ W[prefix + "text_projection"]


def _keyword_tail(W, feats, pad, feat_len, training, target_len, nhead_clip, sot, eot, vq_temp, apply_scaling, forced_tokens=None,
                  aux: Optional[dict] = None):
    ds, ds_len, quantity = cif_forward(W, "downsampling.", feats, pad, target_len if training else None,
                                       apply_scaling=apply_scaling)
    kw = ds
    if "linear_proj.weight" in W:
        kw = F.linear(kw, W["linear_proj.weight"], W["linear_proj.bias"])
    else:                                                     # MLPLayers: Linear, ReLU, (Dropout), Linear
        i = 0
        while f"linear_proj.sequential.{i}.weight" in W:
            if i > 0:
                kw = F.relu(kw)
            kw = F.linear(kw, W[f"linear_proj.sequential.{i}.weight"], W[f"linear_proj.sequential.{i}.bias"])
            i += 3
    if "bn_layer.bn_layer.weight" in W:                       # BatchNorm1d over (B, N) positions
        g, b = W["bn_layer.bn_layer.weight"], W["bn_layer.bn_layer.bias"]
        if training:
            mu = kw.mean(dim=(0, 1))
            var = kw.var(dim=(0, 1), unbiased=False)
        else:
            mu, var = W["bn_layer.bn_layer.running_mean"], W["bn_layer.bn_layer.running_var"]
        kw = (kw - mu) / torch.sqrt(var + 1e-5) * g + b
    emb = W["clip.model.token_embedding.weight"]
    cos = F.normalize(kw, dim=-1, eps=1e-8) @ F.normalize(emb, dim=-1, eps=1e-8).t()
    prob = vq_forward(cos, vq_temp, training, forced=forced_tokens)
    if aux is not None:
        masked = cos.detach().clone()
        masked[..., [0, 2, 3]] = float("-inf")
        aux.update(cos=masked, tokens=masked.argmax(-1), kw_projected=kw.detach())
    keywords = prob @ emb
    out = clip_encode_keywords(W, "clip.model.", keywords, ds_len, nhead_clip, sot, eot)
    return out, keywords, ds_len, quantity


def cascaded_plus_forward(W, audio_feat, audio_len, nhead, training=False, target_len=None, nhead_clip=8, sot=49406,
                          eot=49407, vq_temp=0.1, apply_scaling=True, forced_tokens=None, aux=None):
    """kw_branches.py:701-753 -> (cascaded_audio_feat, keywords, dsample_len, quantity_out)."""
    pad = get_keypadding_mask(audio_feat.shape[1], audio_len)
    feats = mha_and_norm_forward(W, "self_att.", audio_feat, pad, nhead)
    return _keyword_tail(W, feats, pad, audio_len, training, target_len, nhead_clip, sot, eot, vq_temp, apply_scaling, forced_tokens,
                         aux)


def hybrid_plus_forward(W, audio_feat, audio_len, nhead, training=False, target_len=None, nhead_clip=8, sot=49406,
                        eot=49407, vq_temp=0.1, apply_scaling=True, forced_tokens=None, aux=None):
    """kw_branches.py:808-866 -> (parallel_audio_feat, cascaded_audio_feat, keywords, dsample_len, quantity_out)."""
    B, T, D = audio_feat.shape
    pad = get_keypadding_mask(T + 1, audio_len + 1)
    src = torch.cat([W["cls"].expand(B, -1, -1), audio_feat], dim=1)
    post = mha_and_norm_forward(W, "self_att.", src, pad, nhead)
    par = F.linear(post[:, 0], W["parallel_proj.weight"], W["parallel_proj.bias"])
    rest = _keyword_tail(W, post[:, 1:], pad[:, 1:], audio_len, training, target_len, nhead_clip, sot, eot, vq_temp,
                         apply_scaling, forced_tokens, aux)
    return (par,) + rest
