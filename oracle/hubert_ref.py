"""fp32 torch-CPU restatement of the HuBERT speech encoder as the reference runs it.

ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, in structure, the two monkey-patched functions of the reference
(avssl/module/speech_encoder_plus.py:29-64 ``extract_features`` and :67-107
``customHubertForward``) and the wrapper forward (:506-518, :520-634).  The arithmetic below
those call sites lives in fairseq @ b5a039c292facba9c73f59ff34621ec131d82341
(requirements.txt:6), which is NOT in /root/reference nor installed in this image; it is
restated from its published algorithm:

* ConvFeatureExtractionModel: 7x Conv1d(512, k, s) (k=(10,3,3,3,3,2,2), s=(5,2,2,2,2,2,2)),
  mode "default": GroupNorm(512 groups, 512 ch, eps 1e-5, affine) after conv 0 only, no conv bias;
  mode "layer_norm": LayerNorm(512) after every conv, conv bias.  GELU(erf) after each.
* HubertModel: LayerNorm(512) -> forward_padding_mask (chunk .all()) -> Linear 512->D.
* TransformerEncoder: zero padded frames; pos_conv = Conv1d(D, D, 128, padding 64, groups 16)
  (weight-norm folded into a plain weight), SamePad drops the last frame, GELU; x += pos;
  LayerNorm (post-LN variant only); NL x TransformerSentenceEncoderLayer.
* MultiheadAttention: q = (x Wq + bq) * dh^-0.5 ; k, v ; softmax in fp32 with -inf key mask.

State-dict key names are fairseq's (with ``encoder.pos_conv.0.weight`` already folded).

``store`` (default None = the fp32 reference arithmetic): a CONTROL for the recall-parity question.  When a
callable is given (``bf16_store``), it is applied at exactly the tensors the HIP path keeps in bf16 - every conv output, the
feature LayerNorm, the projection, the pos_conv sum, every LayerNorm output, Q / K / V, the attention probabilities in front of
P.V (their row sum stays fp32), the context, both residual sums and the FFN activation - and ``bf16_weights`` rounds the GEMM
weights the kernels hold in bf16 (biases, norm affines and conv layer 0 stay fp32 there too).  Everything else (accumulation,
softmax, GELU, statistics) stays fp32.  That separates "what bf16 storage does to the result" from "what a kernel defect does".
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .lengths import fairseq_valid_frames, feat_len_rule


@dataclass
class HubertArch:
    embed_dim: int = 768
    ffn_dim: int = 3072
    layers: int = 12
    heads: int = 12
    conv_dim: int = 512
    conv_kernels: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_strides: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    extractor_mode: str = "default"      # "default" (base) | "layer_norm" (large)
    conv_bias: bool = False
    layer_norm_first: bool = False       # False = post-LN (base), True = pre-LN (large)
    pos_conv_kernel: int = 128
    pos_conv_groups: int = 16
    normalize_wav: bool = False          # fairseq task cfg.normalize (False base, True large)
    downsample_rate: int = 320
    feature_grad_mult: float = 1.0       # fairseq HubertConfig.feature_grad_mult (0.1 in hubert_base_librispeech.yaml); only acts
                                         # on the backward of a TRAINABLE encoder

    @staticmethod
    def base() -> "HubertArch":
        return HubertArch()

    @staticmethod
    def large() -> "HubertArch":
        return HubertArch(embed_dim=1024, ffn_dim=4096, layers=24, heads=16,
                          extractor_mode="layer_norm", conv_bias=True,
                          layer_norm_first=True, normalize_wav=True)


def init_hubert_weights(arch: HubertArch, seed: int = 7122, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no checkpoints exist offline).  Conv / linear ~ N(0, s),
    norms ~ (1 + N(0, .1), N(0, .1)) so that every affine term is exercised."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def randn(*shape, s=std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    def norm(prefix, n):
        W[prefix + ".weight"] = 1.0 + randn(n, s=0.1)
        W[prefix + ".bias"] = randn(n, s=0.1)

    cin = 1
    for i, k in enumerate(arch.conv_kernels):
        fan_in = cin * k
        W[f"feature_extractor.conv_layers.{i}.0.weight"] = randn(arch.conv_dim, cin, k, s=(2.0 / fan_in) ** 0.5)
        if arch.conv_bias:
            W[f"feature_extractor.conv_layers.{i}.0.bias"] = randn(arch.conv_dim, s=0.05)
        if arch.extractor_mode == "default" and i == 0:
            norm(f"feature_extractor.conv_layers.{i}.2", arch.conv_dim)
        if arch.extractor_mode == "layer_norm":
            norm(f"feature_extractor.conv_layers.{i}.2.1", arch.conv_dim)
        cin = arch.conv_dim
    norm("layer_norm", arch.conv_dim)
    D = arch.embed_dim
    W["post_extract_proj.weight"] = randn(D, arch.conv_dim, s=arch.conv_dim ** -0.5)
    W["post_extract_proj.bias"] = randn(D, s=0.05)
    gsz = D // arch.pos_conv_groups
    W["encoder.pos_conv.0.weight"] = randn(D, gsz, arch.pos_conv_kernel, s=(gsz * arch.pos_conv_kernel) ** -0.5)
    W["encoder.pos_conv.0.bias"] = randn(D, s=0.05)
    norm("encoder.layer_norm", D)
    for i in range(arch.layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + f"self_attn.{n}.weight"] = randn(D, D, s=D ** -0.5)
            W[p + f"self_attn.{n}.bias"] = randn(D, s=0.05)
        norm(p + "self_attn_layer_norm", D)
        W[p + "fc1.weight"] = randn(arch.ffn_dim, D, s=D ** -0.5)
        W[p + "fc1.bias"] = randn(arch.ffn_dim, s=0.05)
        W[p + "fc2.weight"] = randn(D, arch.ffn_dim, s=arch.ffn_dim ** -0.5)
        W[p + "fc2.bias"] = randn(D, s=0.05)
        norm(p + "final_layer_norm", D)
    return W


def bf16_store(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even, as the kernels' v_cvt_pk_bf16_f32) and back to fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_weights(W: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The weights as the HIP path holds them: GEMM operands in bf16 (conv layers 1-6, post_extract_proj, pos_conv, q/k/v/out
    projections, fc1, fc2); conv layer 0, every bias and every norm affine in fp32 (speech_encoder.py:_load_weights)."""
    out = {}
    for k, v in W.items():
        gemm = k.endswith(".weight") and v.dim() >= 2 and not k.startswith("feature_extractor.conv_layers.0.")
        out[k] = bf16_store(v) if gemm else v
    return out


def _keep(t):
    return t


def _site(store, name: str):
    """The rounding applied at storage site class ``name``: identity without a store; the store itself for a plain callable
    (bf16_store: every site); for a ``SitedStore`` only the classes it lists (tools/storage_ablation.py: which site moves recall)."""
    if store is None:
        return _keep
    if isinstance(store, SitedStore):
        return store.fn if name in store.sites else _keep
    return store


class SitedStore:
    """bf16 rounding at a SUBSET of the storage site classes: "conv" (conv stack outputs), "ln" (every LayerNorm output incl. the feature
    and encoder LayerNorms), "proj" (post_extract_proj output), "residual" (the pos_conv sum and both residual sums of every layer), "qkv",
    "p" (un-normalised attention probabilities in front of P.V), "ctx" (attention context), "ffn_act" (GELU(fc1))."""
    ALL = ("conv", "ln", "proj", "residual", "qkv", "p", "ctx", "ffn_act")

    def __init__(self, sites, fn=bf16_store):
        assert set(sites) <= set(self.ALL), sites
        self.sites, self.fn = frozenset(sites), fn


def fold_weight_norm(weight_g: torch.Tensor, weight_v: torch.Tensor) -> torch.Tensor:
    """nn.utils.weight_norm(conv, dim=2): w = g * v / ||v|| with the norm over dims (0, 1)."""
    n = weight_v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return weight_g * weight_v / n


def preprocess_input(wavs: Sequence[torch.Tensor], normalize: bool):
    """speech_encoder_plus.py:506-518."""
    if normalize:
        wavs = [F.layer_norm(w, w.shape) for w in wavs]
    lens = torch.tensor([len(w) for w in wavs], dtype=torch.long)
    L = int(lens.max())
    padded = torch.zeros(len(wavs), L, dtype=torch.float32)
    for b, w in enumerate(wavs):
        padded[b, : len(w)] = w
    mask = torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)
    return padded, mask


def conv_feature_extractor(W, arch: HubertArch, x: torch.Tensor, collect: Optional[list] = None, store=None) -> torch.Tensor:
    """fairseq ConvFeatureExtractionModel.forward: (B, L) -> (B, C, T)."""
    st = _site(store, "conv")
    x = x.unsqueeze(1)
    for i, (k, s) in enumerate(zip(arch.conv_kernels, arch.conv_strides)):
        w = W[f"feature_extractor.conv_layers.{i}.0.weight"]
        b = W.get(f"feature_extractor.conv_layers.{i}.0.bias")
        x = F.conv1d(x, w, b, stride=s)
        if arch.extractor_mode == "default" and i == 0:
            x = F.group_norm(x.float(), arch.conv_dim,
                             W[f"feature_extractor.conv_layers.{i}.2.weight"],
                             W[f"feature_extractor.conv_layers.{i}.2.bias"], 1e-5)
        elif arch.extractor_mode == "layer_norm":
            if i > 0:
                x = st(x)               # the conv GEMM's bf16 output in front of the LayerNorm (layer 0 fuses conv + LN + GELU)
            x = F.layer_norm(x.transpose(1, 2), (arch.conv_dim,),
                             W[f"feature_extractor.conv_layers.{i}.2.1.weight"],
                             W[f"feature_extractor.conv_layers.{i}.2.1.bias"], 1e-5).transpose(1, 2)
        x = st(F.gelu(x))
        if collect is not None:
            collect.append(x)
    return x


def forward_padding_mask(T: int, padding_mask: torch.Tensor) -> torch.Tensor:
    """fairseq HubertModel.forward_padding_mask."""
    extra = padding_mask.size(1) % T
    if extra > 0:
        padding_mask = padding_mask[:, :-extra]
    padding_mask = padding_mask.view(padding_mask.size(0), T, -1)
    return padding_mask.all(-1)


def self_attention(W, p: str, x: torch.Tensor, key_padding_mask: Optional[torch.Tensor], heads: int, drop=None, store=None) -> torch.Tensor:
    """fairseq MultiheadAttention (self-attention).  x: (B, T, D).  ``drop`` (train mode): applied to the attention
    probabilities (B, H, T, T), fairseq's dropout_module(attn_weights)."""
    B, T, D = x.shape
    dh = D // heads
    st, st_p, st_ctx = _site(store, "qkv"), _site(store, "p"), _site(store, "ctx")
    # fairseq scales q before the product; dh^-0.5 = 1/8 is a power of two, so scaling the stored (bf16) q commutes with the rounding
    q = st(F.linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"])) * dh ** -0.5
    k = st(F.linear(x, W[p + "k_proj.weight"], W[p + "k_proj.bias"]))
    v = st(F.linear(x, W[p + "v_proj.weight"], W[p + "v_proj.bias"]))
    q = q.view(B, T, heads, dh).transpose(1, 2)
    k = k.view(B, T, heads, dh).transpose(1, 2)
    v = v.view(B, T, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    if store is None:
        a = torch.softmax(s.float(), dim=-1)
        if drop is not None:
            a = drop(a)
        o = a @ v
    else:               # flash form: un-normalised probabilities rounded in front of P.V, fp32 row sum, one division at the end
        assert drop is None
        e = torch.exp(s.float() - s.float().amax(dim=-1, keepdim=True))
        o = (st_p(e) @ v) / e.sum(dim=-1, keepdim=True)
    o = st_ctx(o.transpose(1, 2).reshape(B, T, D))
    return F.linear(o, W[p + "out_proj.weight"], W[p + "out_proj.bias"])


def encoder_layer(W, arch: HubertArch, i: int, x: torch.Tensor, kpm: Optional[torch.Tensor], drop=None, store=None) -> torch.Tensor:
    """fairseq TransformerSentenceEncoderLayer.forward.  ``drop`` = None: eval mode.  Train mode (the reference's training
    step runs the frozen HuBERT in train mode, see hubert_forward): ``drop(site, layer, tensor)`` is called at the layer's
    dropout sites - "attn" (attention probabilities), "dropout1" (after out_proj, before the residual), "dropout3" (after
    fc2, before the residual); dropout2 (activation_dropout) is 0 in the released HuBERT configs."""
    p = f"encoder.layers.{i}."
    D = arch.embed_dim
    d_att = (lambda t: drop("attn", i, t)) if drop is not None else None
    d1 = (lambda t: drop("dropout1", i, t)) if drop is not None else (lambda t: t)
    d3 = (lambda t: drop("dropout3", i, t)) if drop is not None else (lambda t: t)

    st, st_ln, st_act = _site(store, "residual"), _site(store, "ln"), _site(store, "ffn_act")

    def ln(name, t):
        return st_ln(F.layer_norm(t, (D,), W[p + name + ".weight"], W[p + name + ".bias"], 1e-5))

    def ffn(t):
        return F.linear(st_act(F.gelu(F.linear(t, W[p + "fc1.weight"], W[p + "fc1.bias"]))), W[p + "fc2.weight"], W[p + "fc2.bias"])

    if not arch.layer_norm_first:
        x = ln("self_attn_layer_norm", st(x + d1(self_attention(W, p + "self_attn.", x, kpm, arch.heads, d_att, store))))
        x = ln("final_layer_norm", st(x + d3(ffn(x))))
    else:
        x = st(x + d1(self_attention(W, p + "self_attn.", ln("self_attn_layer_norm", x), kpm, arch.heads, d_att, store)))
        x = st(x + d3(ffn(ln("final_layer_norm", x))))
    return x


def hubert_forward(W, arch: HubertArch, padded_wav: torch.Tensor, wav_padding_mask: Optional[torch.Tensor],
                   debug: Optional[dict] = None, drop=None, store=None) -> List[torch.Tensor]:
    """customHubertForward (speech_encoder_plus.py:67-107) + patched extract_features (:29-64).

    ``drop`` = None is eval mode.  In the reference's TRAINING step the frozen HuBERT is in train mode (the constructor's
    eval() at :402 is undone by Lightning's model.train(); nothing overrides train()), so its dropout sites are live:
    ``drop(site, layer, tensor)`` is called with site "input" (dropout_input, :87), "encoder" (F.dropout after the encoder
    LayerNorm, :42) and the per-layer sites of encoder_layer; the caller supplies the masks (the oracle has no RNG of its own).

    Returns layer_results = [encoder input, out_1 .. out_NL], each (B, T, D)."""
    st, st_ln, st_proj = _site(store, "residual"), _site(store, "ln"), _site(store, "proj")
    conv_outs = [] if debug is not None else None
    feats = conv_feature_extractor(W, arch, padded_wav, conv_outs, store)   # :75
    fgm = getattr(arch, "feature_grad_mult", 1.0)
    if fgm != 1.0 and feats.requires_grad:                                   # fairseq HubertModel.forward: GradMultiply.apply(features,
        feats = feats * fgm + feats.detach() * (1.0 - fgm)                   # feature_grad_mult): identity forward, scaled backward
    feats = feats.transpose(1, 2)                                            # :77
    feats = st_ln(F.layer_norm(feats, (arch.conv_dim,), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5))  # :78
    T = feats.shape[1]
    pm = forward_padding_mask(T, wav_padding_mask) if wav_padding_mask is not None else None        # :81-82
    x = st_proj(F.linear(feats, W["post_extract_proj.weight"], W["post_extract_proj.bias"]))        # :84-85
    if debug is not None:
        debug["conv"] = conv_outs
        debug["proj"] = x.clone()
        debug["padding_mask"] = pm
    if drop is not None:
        x = drop("input", -1, x)                                             # :87 dropout_input
    # mask=None is falsy -> no time masking (:90-94)
    if pm is not None:
        x = x.masked_fill(pm.unsqueeze(-1), 0.0)                             # :32-33 index_put(x, mask, 0)
    k = arch.pos_conv_kernel
    xc = F.conv1d(x.transpose(1, 2), W["encoder.pos_conv.0.weight"], W["encoder.pos_conv.0.bias"],
                  padding=k // 2, groups=arch.pos_conv_groups)
    if k % 2 == 0:
        xc = xc[:, :, :-1]                                                   # SamePad
    xc = F.gelu(xc).transpose(1, 2)                                          # :35-36
    x = st(x + xc)                                                           # :37
    if not arch.layer_norm_first:
        x = st_ln(F.layer_norm(x, (arch.embed_dim,), W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"], 1e-5))  # :39-40
    if drop is not None:
        x = drop("encoder", -1, x)                                           # :42
    layer_results = [x]                                                      # :47
    for i in range(arch.layers):                                             # :49-53 (layerdrop 0)
        x = encoder_layer(W, arch, i, x, pm, drop, store)
        layer_results.append(x)
    return layer_results


def speech_encoder_forward(W, arch: HubertArch, wavs: Sequence[torch.Tensor], drop=None, store=None):
    """FairseqSpeechEncoder_Hubert.forward (no crop; eval unless ``drop`` is given, see hubert_forward): returns
    (hidden_states tuple, feat_len).  speech_encoder_plus.py:554-611."""
    padded, mask = preprocess_input(wavs, arch.normalize_wav)
    hs = hubert_forward(W, arch, padded, mask, drop=drop, store=store)
    T = hs[-1].shape[1]
    feat_len = torch.tensor(feat_len_rule([len(w) for w in wavs], T, arch.downsample_rate), dtype=torch.long)
    return tuple(hs), feat_len
