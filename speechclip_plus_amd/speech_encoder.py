"""HuBERT speech encoder on MI355X: host-side mirror of the reference wrapper
``FairseqSpeechEncoder_Hubert`` (avssl/module/speech_encoder_plus.py:319-634), driving the HIP kernels of
libspeechclip_hip.so.  Same constructor keywords, ``forward(wav, wav_len, feat_select_idx,
return_hidden_states) -> (feat, feat_len[, hidden_states])``, ``out_dim``, ``downsample_rate``,
``trainable_params()``.

Data layout in HBM: channels-last bf16 activations in a row layout with a pitch PER UTTERANCE (round 4): utterance b /
frame t -> row row0[b] + t, pitch_b = row0[b + 1] - row0[b] = roundup(n_b + 1, 8) with n_b the frames anything downstream reads
(``_Plan.bind``), and 2^(6-l) times that table down the conv stack (the waveform: 320 samples per row).  With that layout every
strided Conv1d of the feature extractor is ONE flat GEMM whose A rows overlap (lda = stride*C, K = k*C), the transformer GEMMs see
M = sum_b pitch_b rows - the work follows the real lengths of a ragged batch instead of its padded length - and attention /
pos_conv / the weighted sum find their utterance through ``ops.RowSegments``.  Rows t >= n_b of an utterance are scratch: finite,
never read as keys, never returned.  Unfrozen layers (hubert_train.py) keep the uniform pitch R = roundup(T + 2, 128) of rounds 1-3.

There is no pretrained checkpoint offline (the reference downloads hubert_base_ls960.pt,
speech_encoder_plus.py:327-331,382): weights come from ``state_dict`` (fairseq key names, pos_conv weight-norm
already folded or given as weight_g / weight_v) or are seeded random.
"""
import logging
import os
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import nn

from . import ops
from .weighted_sum import WeightedSumLayer

logger = logging.getLogger(__name__)

FEAT_SELECT_IDX_WEIGHTED_SUM_MODE = "weighted_sum"


@dataclass
class HubertArch:
    embed_dim: int = 768
    ffn_dim: int = 3072
    layers: int = 12
    heads: int = 12
    conv_dim: int = 512
    conv_kernels: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_strides: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    extractor_mode: str = "default"
    conv_bias: bool = False
    layer_norm_first: bool = False
    pos_conv_kernel: int = 128
    pos_conv_groups: int = 16
    normalize_wav: bool = False
    downsample_rate: int = 320
    # fairseq HubertConfig dropouts of the released checkpoints (hubert_base_librispeech.yaml; hubert_large_librivox.yaml has
    # all of them 0).  They act in the reference's TRAIN step although HuBERT is frozen: Lightning's model.train() flips the
    # eval() of speech_encoder_plus.py:402 back.  encoder_layerdrop is overwritten by layer_drop (:404-411, 0.0 in every recipe).
    dropout: float = 0.1               # after the encoder LayerNorm, after out_proj and after fc2
    attention_dropout: float = 0.1     # on the attention probabilities
    activation_dropout: float = 0.0    # after the FFN activation
    dropout_input: float = 0.1         # on post_extract_proj's output
    feature_grad_mult: float = 0.1     # fairseq GradMultiply on the conv extractor's output (hubert_base_librispeech.yaml: 0.1;
                                       # hubert_large_librivox.yaml: 1.0); acts only on the backward of a fully trainable encoder


ARCHS = {
    "hubert": HubertArch(),
    "hubert_base": HubertArch(),
    "hubert_large_ll60k": HubertArch(embed_dim=1024, ffn_dim=4096, layers=24, heads=16, extractor_mode="layer_norm",
                                     conv_bias=True, layer_norm_first=True, normalize_wav=True, dropout=0.0,
                                     attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0,
                                     feature_grad_mult=1.0),
}


def conv_out_lengths(L: int, arch: HubertArch) -> List[int]:
    out, t = [], int(L)
    for k, s in zip(arch.conv_kernels, arch.conv_strides):
        t = (t - k) // s + 1
        out.append(t)
    return out


def random_crop_max_length(audio: torch.Tensor, max_len: int, orig_len: int = 1000000000) -> torch.Tensor:
    """avssl/data/audio_transforms.py:5-23 (called at speech_encoder_plus.py:548-552)."""
    audio_len = min(len(audio), orig_len)
    if audio_len <= max_len or max_len < 0:
        return audio[:audio_len]
    offset = np.random.randint(audio_len - max_len)
    return audio[offset: offset + max_len]


def crop_windows(lens: Sequence[int], max_len: int) -> Tuple[List[int], List[int]]:
    """The training crop of speech_encoder_plus.py:548-552 as (offset, cropped length) per utterance: the same draws from numpy's global
    generator, in the same order, as the reference's loop over ``random_crop_max_length`` (audio_transforms.py:5-23) - one
    ``np.random.randint(len - max_len)`` per utterance longer than ``max_len``, none for the others - so a seeded run crops the same
    windows.  The samples themselves are never touched on the host: the kernels that read the caller's batch take the offsets."""
    offs, out = [], []
    for l in lens:
        l = int(l)
        if l <= max_len or max_len < 0:
            offs.append(0)
            out.append(l)
        else:
            offs.append(int(np.random.randint(l - max_len)))
            out.append(int(max_len))
    return offs, out


def random_hubert_state_dict(arch: HubertArch, seed: int = 7122) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights with fairseq key names (fan-in scaled so activations stay O(1))."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def randn(*shape, s):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    W: Dict[str, torch.Tensor] = {}

    def norm(prefix, n):
        W[prefix + ".weight"] = 1.0 + randn(n, s=0.1)
        W[prefix + ".bias"] = randn(n, s=0.1)

    cin = 1
    for i, k in enumerate(arch.conv_kernels):
        W[f"feature_extractor.conv_layers.{i}.0.weight"] = randn(arch.conv_dim, cin, k, s=(2.0 / (cin * k)) ** 0.5)
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


_USE_GRAPH = os.environ.get("SC_ENCODER_GRAPH", "0") == "1"      # opt-in: measured +-0 on one GPU
# Round 3, opt-in (SC_FUSED_LN=1): the frozen post-LN encoder WITHOUT LayerNorm launches - the two LayerNorms of a layer folded into
# their neighbour GEMMs, residual stream / hidden states kept as raw rows + row statistics (csrc/gemm256_bf16.hip "LN").  Correct
# (slightly closer to the oracle than the LayerNorm kernels: 8.8e-3 vs 9.7e-3 worst hidden state) but NOT faster: same-box A/B at
# B = 64 x 10 s: forward 11.75-11.77 ms folded vs 11.70-11.77 ms with the 25 LayerNorm launches, train step 13.6-14.0 vs 13.4 ms
# (the weighted sum and its backward normalise 12 states on the fly).  The 24 launches and 2.4 GB of LayerNorm traffic it removes
# (0.6 ms) come back as VALU work in GEMM epilogues that are already issue-bound (+12 us fc1, +19 fc2, +6 QKV, +7 out_proj per
# launch with all operands prefetched into LDS; +25..45 before that).  DESIGN.md section 7.
_FUSED_LN = os.environ.get("SC_FUSED_LN", "0") == "1"
# The frozen encoder of train step N + 1 on a stream of its own, under the branch / head / loss / backward kernels of step N (two
# alternating sets of resident buffers): see _encode_overlapped.  SC_ENC_OVERLAP=0: everything on the caller's stream.
_ENC_OVERLAP = os.environ.get("SC_ENC_OVERLAP", "1") == "1"


def _mix32(x: int) -> int:
    """lowbias32 (the kernels' sc_hash32) on the host: decorrelates the per-site dropout seeds."""
    x &= 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    return x ^ (x >> 16)


def _roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _Plan:
    """Shapes + resident workspaces for one (B, L) batch geometry.

    ``seg_mode`` (the frozen encoder, round 4): buffers are sized for the uniform worst case (every utterance L samples long,
    pitch ``R = roundup(T + 1, 8)``) and ``bind`` lays the CURRENT batch out in them by its real lengths - per-utterance pitches,
    ``ops.RowSegments`` tables uploaded with the batch's other integers - so ``M``, ``hidden`` and the row buffers below are views
    that change from forward to forward.  Otherwise (unfrozen layers: their backward kernels index utterances by ``b * R``) the
    uniform layout of rounds 1-3 with ``R = roundup(T + 2, 128)``."""

    SLACK = 64           # rows behind the last utterance that a 64-key attention tile / a conv window may read (never written)

    out_align, branch_rows = 8, 0          # set per call by the encoder (_plan): output pitch granularity, zero rows behind the output

    def __init__(self, arch: HubertArch, B: int, L: int, device, seg_mode: bool = False):
        # seg_mode: L is the plan's CAPACITY (the longest padded batch it can hold, speech_encoder._plan buckets the batch length);
        # the geometry of the batch in flight (L, T_l, T) is set per forward by set_length - real data changes its longest
        # utterance from batch to batch, and a fresh 7 GB of zeroed workspaces per distinct length would cost more than the step
        self.arch = arch
        self.B, self.L = B, L
        self.L_cap = L
        self.T_l = conv_out_lengths(L, arch)
        self.T = self.T_l[-1]
        if self.T < 1:
            raise ValueError(f"waveform too short for the conv stack: L={L}")
        self.seg_mode = seg_mode
        nl = len(arch.conv_kernels)
        self.spr = 1                                    # waveform samples per row of the last conv layer (320)
        for s_ in arch.conv_strides:
            self.spr *= s_
        # seg_mode: the frames < n of an utterance need conv rows < 2^(6-l) n + c_l (c_0 = 15) of layer l and samples < 320 n + 80,
        # all inside a pitch of n + 1 rows (tests/test_host_cpu.py::test_segment_geometry); the legacy layout stores every padded row
        self.R = _roundup(self.T + 1, ops.RowSegments.GRAN) if seg_mode else _roundup(self.T + 2, 128)
        self.R_l = [self.R * (2 ** (nl - 1 - i)) for i in range(nl)]
        if not seg_mode:
            assert all(r >= t for r, t in zip(self.R_l, self.T_l))
        self.ldw = _roundup(max(L, arch.conv_strides[0] * self.R_l[0] + 16), 64)
        C, D, F = arch.conv_dim, arch.embed_dim, arch.ffn_dim
        bf, dev = torch.bfloat16, device
        z = lambda *s, dtype=bf: torch.zeros(*s, device=dev, dtype=dtype)
        M = B * self.R
        self.M = M
        S = self.SLACK if seg_mode else 0
        G, halo = arch.pos_conv_groups, arch.pos_conv_kernel // 2
        self.halo = halo
        if seg_mode:
            self.wav_pad = z(self.spr * M + 64, dtype=torch.float32)           # ONE flat waveform, utterance b at sample 320 row0[b]
            self.xg = z(G * (M + 2 * halo * B) * (D // G))                     # flat slab buffer (sc_posconv_prep_seg)
            self.vt = z(D * (M + S))
            self._seg_cache = (None, None)
        else:
            self.wav_pad = z(B, self.ldw, dtype=torch.float32)
            self.xg = z(G, B, self.R + 2 * halo, D // G)
            self.vt = z(B, arch.heads, D // arch.heads, self.R)
        # conv activations, channels-last, +8 slack rows (the last GEMM row's window runs one row past the end)
        self._conv = [z(B * r + 8 + S, C) for r in self.R_l]
        self._rows = dict(feat_ln=z(M + S, C), x_proj=z(M + S, D), xz=z(M + S, D), pre=z(M + S, D), x1=z(M + S, D),
                          qk=z(M + S, 2 * D), ctx=z(M + S, D), ffn=z(M + S, F))
        self._hidden = z((arch.layers + 1) * M * D + S * D)
        self.NL, self.D = arch.layers + 1, D
        # LayerNorm-free layers: hidden[n >= 1] then hold the rows in FRONT of layer n's final LayerNorm, stats[n] their (sum, sum of
        # squares) strips, stats1 the scratch for the rows in front of LN1; lazy = how consumers read such states (ops.LazyStates)
        self.stats = z(arch.layers + 1, M, 8, 2, dtype=torch.float32) if (_FUSED_LN and not seg_mode) else None
        self.stats1 = z(M, 8, 2, dtype=torch.float32) if (_FUSED_LN and not seg_mode) else None
        self.lazy = None
        self.valid = torch.zeros(B, device=dev, dtype=torch.int32)
        self.len_dev = torch.zeros(B, device=dev, dtype=torch.int64)
        self.wav_in = torch.zeros(B, L, device=dev, dtype=torch.float32) if _USE_GRAPH and torch.device(dev).type == "cuda" else None
        self.graph, self.graph_warm = None, 0        # hipGraph of the frozen encoder forward for this geometry
        self.train = {}              # per unfrozen layer: activations kept for the backward (hubert_train.TrainableLayers)
        self.generation = 0          # forwards run on this plan (weighted_sum.PaddedFeatHandle.check_fresh)
        # overlapped schedule (speech_encoder._claim): a forward that will be differentiated leaves ``grad_pending`` set until its
        # backward has ENQUEUED its last read of this plan's buffers (weighted-sum / unfrozen-layer backward) and recorded
        # ``readers_done`` behind it on the stream it ran on
        self.grad_pending = False
        self.readers_done = None
        self.wav_off, self.src_L = None, L
        self.seg = None
        self.bind(None)

    def set_length(self, L: int) -> None:
        """geometry of the batch in flight: padded length L <= the plan's capacity (segment layout only)"""
        if L == self.L:
            return
        assert self.seg_mode and L <= self.L_cap, (L, self.L_cap)
        T_l = conv_out_lengths(L, self.arch)
        if T_l[-1] < 1:
            raise ValueError(f"waveform too short for the conv stack: L={L}")
        self.L, self.T_l, self.T = L, T_l, T_l[-1]

    def segments(self, pitch, keys, device, keys_known: bool = True):
        """ops.RowSegments of a batch; the tables of the previous batch are re-used when the layout and the work order are the same
        (read-only on the device: a fresh upload per change, never an overwrite of tables a kernel in flight may read)"""
        key = (tuple(pitch), tuple(keys), keys_known)
        if self._seg_cache[0] != key:
            self._seg_cache = (key, ops.RowSegments(pitch, keys, device, keys_known=keys_known))
        return self._seg_cache[1]

    def release(self) -> None:
        """Called by the backward that read this plan's resident buffers last (weighted_sum / head_tail), on the stream it ran on:
        from the recorded event on, the encoder stream may overwrite them."""
        if self.hidden.is_cuda:          # (every reader records: with two readers in one backward - hybrid+ - the later event counts)
            ev = torch.cuda.Event()
            ev.record()
            self.readers_done = ev
        self.grad_pending = False

    def bind(self, seg) -> None:
        """Lay the current batch out: ``seg`` = its ops.RowSegments (seg_mode) or None = the uniform B x R rows."""
        self.seg = seg
        M = seg.rows if seg is not None else self.B * self.R
        self.M = M
        nl = len(self.R_l)
        self.conv = self._conv if seg is None else [c[: M * (2 ** (nl - 1 - i)) + 8] for i, c in enumerate(self._conv)]
        for k, v in self._rows.items():
            setattr(self, k, v[:M])
        self.hidden = self._hidden[: self.NL * M * self.D].view(self.NL, M, self.D)
        # uniform pitch of what leaves the encoder ([B, Rout, D]); a multiple of 64 when a cascaded+/hybrid+ attention block reads it in place
        self.Rout = -(-seg.max_pitch // self.out_align) * self.out_align if seg is not None else self.R


class FairseqSpeechEncoder_Hubert(nn.Module):
    MODEL_DOWNSAMPLE_RATE = {"hubert": 320, "hubert_base": 320, "hubert_large_ll60k": 320}

    def __init__(self, name: str = "hubert", pretrained: bool = False, trainable: bool = False, device: str = "cuda",
                 feat_select_idx: Union[str, list] = "all", layer_drop: Union[str, float] = 0.0, max_audio_len: int = -1,
                 reinit_layers: List[int] = [], unfreeze_layers: List[int] = [], normalize_hiddenstates: bool = False,
                 normalize_type: str = "s3prl", state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 arch: Optional[HubertArch] = None, seed: int = 7122, **kwargs):
        super().__init__()
        assert name in ARCHS, "Model name({}) should be in {}".format(name, ARCHS.keys())
        self.name = name
        self.arch = arch if arch is not None else ARCHS[name]
        self.pretrained = pretrained
        self.trainable = trainable
        assert not (len(reinit_layers) > 0 and len(unfreeze_layers) > 0)               # speech_encoder_plus.py:416
        train_ids = sorted(set(int(i) for i in (list(reinit_layers) or list(unfreeze_layers))))
        self._train_all = False
        if train_ids:
            assert trainable, "reinit_layers / unfreeze_layers need trainable: true (speech_encoder_plus.py:419,434)"
        elif trainable:
            # speech_encoder_plus.py:556-562: every encoder parameter trains (conv extractor, projection, pos_conv, all layers)
            self._train_all = True
            train_ids = list(range(self.arch.layers))
        self._train_ids = train_ids
        assert self.arch.extractor_mode in ("default", "layer_norm"), self.arch.extractor_mode
        assert self.arch.embed_dim == self.arch.heads * 64, "the attention kernel is built for head_dim 64"
        if not (isinstance(layer_drop, float) and layer_drop == 0.0) and layer_drop != "original":
            raise ValueError(f"layer_drop = {layer_drop} is not supported.")
        self.feat_select_idx = feat_select_idx
        self.hubert_dropout = True              # False: keep the frozen encoder deterministic in train mode (not the reference)
        # Round 4: rows follow the real lengths.  ``ragged`` False = every utterance at the batch's padded length (the reference's
        # amount of work; also what ``return_hidden_states`` uses: those tensors carry every padded row, like the reference's).
        # ``tail_rows``: frames behind feat_len that a consumer still reads - the CIF weight conv of the cascaded+/hybrid+ branches
        # (conv_cif_width 3 or 5, avssl/module/cif.py:44-52) looks one / two frames past the last valid one.
        self._section_ev = None                 # bench.py: a dict that receives event records (start / conv stack done) of one forward
        self.ragged = os.environ.get("SC_RAGGED", "1") == "1"
        self.tail_rows = 2
        self.branch_inplace = False             # model.py: True when a cascaded+/hybrid+ branch consumes the output rows (see _plan)
        self._drop_calls = 0
        self.before_trainable = None            # optional callable invoked right before the first trainable module (train.py)
        self.max_audio_len = max_audio_len
        self.reinit_layers = reinit_layers
        self.unfreeze_layers = unfreeze_layers
        self.normalize_hiddenstates = normalize_hiddenstates
        assert normalize_type in ["s3prl", "method1", "method2"], normalize_type
        if normalize_hiddenstates and normalize_type != "s3prl":
            raise NotImplementedError("normalize_type method1 / method2 is not used by any shipped config")
        self.normalize_type = normalize_type
        self.downsample_rate = self.MODEL_DOWNSAMPLE_RATE[name]
        self.out_dim = self.arch.embed_dim
        self.upstream_model_hiddenstates_len = self.arch.layers + 1
        self._dev = torch.device(device)
        if state_dict is None:
            if pretrained:
                logger.warning("no checkpoint available offline: HuBERT weights are seeded random (seed %d)", seed)
            state_dict = random_hubert_state_dict(self.arch, seed)
        self._load_weights(state_dict)
        self.train_layers = None
        if train_ids:
            from .hubert_train import TrainableLayers
            self.train_layers = TrainableLayers(self.arch, state_dict, train_ids, self._dev, reinit=len(reinit_layers) > 0, seed=seed)
        self.frontend = None
        if self._train_all:
            from .hubert_frontend_train import TrainableFrontend
            self.frontend = TrainableFrontend(self.arch, state_dict, self._dev)
            self.train_layers.frontend = self.frontend
        self._plans: Dict[Tuple[int, int, bool, int], _Plan] = {}
        self.enc_overlap = _ENC_OVERLAP
        self._enc_stream = None           # _encode_overlapped
        self._prev_entry = None
        self._parity = 0
        self._consumer_stream = None
        self._retired = []                # (forward count at eviction, plan): see _plan
        self._switch = None               # _encode_overlapped -> _first_trainable
        self._ov = None                   # (encoder stream, entry event) while _encode_overlapped runs: see _claim
        self._forwards = 0
        if self.feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            self.weightedsum_layer = WeightedSumLayer(
                n_weights=self.upstream_model_hiddenstates_len,
                normalize_features=self.normalize_hiddenstates and self.normalize_type == "s3prl").to(self._dev)

    # ------------------------------------------------------------------------------------------ weights
    def _load_weights(self, sd: Dict[str, torch.Tensor]) -> None:
        a, dev = self.arch, self._dev
        bf = lambda t: t.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        if "encoder.pos_conv.0.weight" not in sd:
            g, v = sd["encoder.pos_conv.0.weight_g"], sd["encoder.pos_conv.0.weight_v"]
            pos_w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()     # weight_norm(dim=2)
        else:
            pos_w = sd["encoder.pos_conv.0.weight"]
        w = {}
        w["conv0_w"] = f32(sd["feature_extractor.conv_layers.0.0.weight"].reshape(a.conv_dim, a.conv_kernels[0]))
        ln_mode = a.extractor_mode == "layer_norm"
        if ln_mode:
            w["conv0_ln_g"] = f32(sd["feature_extractor.conv_layers.0.2.1.weight"])
            w["conv0_ln_b"] = f32(sd["feature_extractor.conv_layers.0.2.1.bias"])
        else:
            w["gn_g"] = f32(sd["feature_extractor.conv_layers.0.2.weight"])
            w["gn_b"] = f32(sd["feature_extractor.conv_layers.0.2.bias"])
        w["conv0_bias"] = f32(sd["feature_extractor.conv_layers.0.0.bias"]) if a.conv_bias else None
        for i in range(1, len(a.conv_kernels)):
            cw = sd[f"feature_extractor.conv_layers.{i}.0.weight"]              # [C_out, C_in, k]
            w[f"conv{i}_w"] = bf(cw.permute(0, 2, 1).reshape(cw.shape[0], -1))  # [C_out, k*C_in]  (tap-major)
            w[f"conv{i}_bias"] = f32(sd[f"feature_extractor.conv_layers.{i}.0.bias"]) if a.conv_bias else None
            if ln_mode:
                w[f"conv{i}_ln_g"] = f32(sd[f"feature_extractor.conv_layers.{i}.2.1.weight"])
                w[f"conv{i}_ln_b"] = f32(sd[f"feature_extractor.conv_layers.{i}.2.1.bias"])
        w["ln_feat_g"], w["ln_feat_b"] = f32(sd["layer_norm.weight"]), f32(sd["layer_norm.bias"])
        w["proj_w"], w["proj_b"] = bf(sd["post_extract_proj.weight"]), f32(sd["post_extract_proj.bias"])
        D, G, Kp = a.embed_dim, a.pos_conv_groups, a.pos_conv_kernel
        Dg = D // G
        # [D, Dg, K] -> per group [Dg_out, K*Dg_in] (tap-major), groups stacked
        w["pos_w"] = bf(pos_w.reshape(G, Dg, Dg, Kp).permute(0, 1, 3, 2).reshape(G, Dg, Kp * Dg))
        w["pos_b"] = f32(sd["encoder.pos_conv.0.bias"])
        w["ln_enc_g"], w["ln_enc_b"] = f32(sd["encoder.layer_norm.weight"]), f32(sd["encoder.layer_norm.bias"])
        for i in range(a.layers):
            p = f"encoder.layers.{i}."
            w[f"l{i}_qkv_w"] = bf(torch.cat([sd[p + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0))
            w[f"l{i}_qkv_b"] = f32(torch.cat([sd[p + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0))
            w[f"l{i}_o_w"], w[f"l{i}_o_b"] = bf(sd[p + "self_attn.out_proj.weight"]), f32(sd[p + "self_attn.out_proj.bias"])
            w[f"l{i}_ln1_g"], w[f"l{i}_ln1_b"] = f32(sd[p + "self_attn_layer_norm.weight"]), f32(sd[p + "self_attn_layer_norm.bias"])
            w[f"l{i}_fc1_w"], w[f"l{i}_fc1_b"] = bf(sd[p + "fc1.weight"]), f32(sd[p + "fc1.bias"])
            w[f"l{i}_fc2_w"], w[f"l{i}_fc2_b"] = bf(sd[p + "fc2.weight"]), f32(sd[p + "fc2.bias"])
            w[f"l{i}_ln2_g"], w[f"l{i}_ln2_b"] = f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"])
        if not a.layer_norm_first:
            # LayerNorm folded into the consumer GEMM (csrc/gemm256_bf16.hip "LN"): W' = W diag(gamma) in bf16, s_n = sum_k W'[n,k] (of
            # the bf16 values the kernel multiplies), c_n = sum_k beta_k W[n,k] + b_n; fc1 with the layer's own LN1, QKV of layer i >= 1
            # with layer i - 1's final LayerNorm
            def fold(W32, b32, g, beta):
                Wf = (W32.float() * g.float()[None, :]).to(torch.bfloat16)
                return (Wf.to(dev).contiguous(), f32(Wf.float().sum(1)), f32(W32.float() @ beta.float() + b32.float()))
            for i in range(a.layers):
                p = f"encoder.layers.{i}."
                w[f"l{i}_fc1_wf"], w[f"l{i}_fc1_sf"], w[f"l{i}_fc1_cf"] = fold(sd[p + "fc1.weight"], sd[p + "fc1.bias"],
                                                                               sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"])
                if i > 0:
                    q = f"encoder.layers.{i - 1}."
                    Wqkv = torch.cat([sd[p + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0)
                    bqkv = torch.cat([sd[p + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0)
                    w[f"l{i}_qkv_wf"], w[f"l{i}_qkv_sf"], w[f"l{i}_qkv_cf"] = fold(Wqkv, bqkv, sd[q + "final_layer_norm.weight"],
                                                                                   sd[q + "final_layer_norm.bias"])
            # affines of the LayerNorm that turns raw state n into hidden state n (row 0: hidden[0] is materialised)
            w["lazy_gamma"] = f32(torch.stack([torch.ones(a.embed_dim)] + [sd[f"encoder.layers.{i}.final_layer_norm.weight"].float() for i in range(a.layers)]))
            w["lazy_beta"] = f32(torch.stack([torch.zeros(a.embed_dim)] + [sd[f"encoder.layers.{i}.final_layer_norm.bias"].float() for i in range(a.layers)]))
        self._w = w          # frozen device tensors (not nn.Parameters: no grads, no optimizer state)
        self._weights_dirty = True       # _encode_overlapped: the encoder stream has to see these before its next forward

    def trainable_params(self) -> list:
        """speech_encoder_plus.py:478-494.  Frozen encoder: only the weighted-sum weights train; with ``reinit_layers`` the
        reference hands the optimiser the re-initialised layers alone (not the weighted-sum weights), with ``unfreeze_layers``
        every parameter that still requires grad."""
        ws = list(self.weightedsum_layer.parameters()) if self.feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE else []
        if self.train_layers is None:
            return ws
        layer_params = list(self.train_layers.parameters())
        if self.frontend is not None:
            layer_params = layer_params + list(self.frontend.parameters())
        return layer_params if len(self.reinit_layers) > 0 else layer_params + ws

    # ------------------------------------------------------------------------------------------ forward
    def _seg_mode(self) -> bool:
        """Per-utterance row pitches (ops.RowSegments) for the frozen encoder; unfrozen layers / the LayerNorm-folded experiment keep
        the uniform layout their kernels index."""
        return self.train_layers is None and not _FUSED_LN

    LENGTH_BUCKET = 32000        # samples: plans of the segment layout are sized for the batch length rounded up to 2 s

    def _plan(self, B: int, L: int) -> _Plan:
        seg_mode = self._seg_mode()
        if conv_out_lengths(L, self.arch)[-1] < 1:
            raise ValueError(f"waveform too short for the conv stack: L={L}")
        # segment layout: the workspaces are views into capacity-sized buffers, so one plan serves every padded length of its 2 s
        # bucket (real batches change their longest utterance all the time); the uniform layout of unfrozen layers is per length
        cap = _roundup(L, self.LENGTH_BUCKET) if seg_mode else L
        key = (B, cap, seg_mode, self._parity)
        if key not in self._plans:
            if len(self._plans) >= (8 if self.enc_overlap else 4):          # keep HBM bounded when lengths vary
                # least recently used goes.  Its buffers may still be read by kernels in flight on the caller's stream while the
                # encoder stream would be handed the freed blocks: the plan itself is kept two more forwards (_retired), by when the
                # encoder's entry-event wait covers every reader
                self._retired.append((self._forwards, self._plans.pop(next(iter(self._plans)))))
            self._plans[key] = _Plan(self.arch, B, cap, self._dev, seg_mode=seg_mode)
        pl = self._plans.pop(key)
        self._plans[key] = pl                    # most recently used last
        self._forwards += 1
        while self._retired and self._retired[0][0] + 3 <= self._forwards:
            self._retired.pop(0)
        # a cascaded+/hybrid+ attention block reads the encoder's output rows in place (mha_block.resident_rows): output pitch a
        # multiple of 64, a few zero rows behind the buffer
        pl.out_align, pl.branch_rows = (64, 8) if self.branch_inplace else (ops.RowSegments.GRAN, 0)
        if seg_mode:
            pl.set_length(L)
        return pl

    def _claim(self, pl: _Plan) -> None:
        """Before the first write into a plan's resident buffers.  On the overlapped schedule the encoder stream runs ahead of the
        caller's stream, so 'the previous user of this plan is done' has to be said explicitly: the backward that read the
        plan last recorded ``readers_done`` (f1 f2 b1 b2 f3: f3 re-uses f1's plan and waits for b1's event, wherever b1 was enqueued);
        a forward whose backward has NOT been enqueued yet can only be waited for as 'everything on the caller's stream so far' - its
        late backward then fails the generation check loudly instead of reading overwritten states."""
        ov = self._ov
        if ov is not None:
            enc, entry = ov
            if pl.grad_pending:
                enc.wait_event(entry)
            elif pl.readers_done is not None:
                enc.wait_event(pl.readers_done)
        pl.readers_done = None
        pl.grad_pending = bool(self.training and torch.is_grad_enabled())

    def _encode_overlapped(self, padded: Optional[torch.Tensor], host_src: Optional[torch.Tensor], lens, ready, save: bool = False,
                           L: Optional[int] = None, off=None) -> _Plan:
        """The frozen encoder of this step on its own stream.  Nothing in it depends on the previous step (no trainable parameter, no
        activation), so it may run UNDER the previous step's branch / head / loss / backward kernels, which are launch-sized and leave
        most of the chip idle (cascaded+: 313 launches, 4.6 ms after a 12.3 ms encoder).  The host enqueues step N's tail, then this
        encoder for step N + 1; the device runs the two streams side by side:
          * two alternating plans (resident hidden states / workspaces): step N's backward still reads plan N % 2 (weighted-sum
            gradient) while the encoder of step N + 1 fills the other one;
          * the encoder of step N waits for the event recorded on the caller's stream at the ENTRY of step N - 1 (everything enqueued
            before it - through step N - 2's backward, the last reader of this plan - has finished); encoders run one after the
            other on the one encoder stream;
          * the caller's stream waits for the encoder's completion event before the weighted sum.
        The INPUT must be ready when the encoder stream gets to it, which may be a step's length before the caller's stream does:
          * a host batch (``host_src``) is copied to the device here, on the encoder stream (pinned memory: asynchronously) - the way
            a loop gets both the overlap and its H2D copy off the critical path;
          * a device batch marked ``wav._sc_ready = True`` (resident data: bench.py's synthetic batch) or ``= torch.cuda.Event`` (a
            prefetcher's copy-done event) is read as soon as that allows;
          * any other device tensor may still be the target of work queued on the caller's stream (e.g. the loop's own ``.to(device)``):
            the encoder stream waits for the caller's stream as of this call - correct, and little overlap.
        Same kernels on the same values: results are bit-identical to the single-stream schedule."""
        main = torch.cuda.current_stream()
        if self._enc_stream is None:
            self._enc_stream = ops.shared_stream("encoder", self._dev)
        enc = self._enc_stream
        entry = torch.cuda.Event()
        entry.record(main)
        if self._prev_entry is not None and not self._weights_dirty:
            enc.wait_event(self._prev_entry)
        else:
            # first overlapped forward, or weights (re)loaded since the last one: the casts / weight-norm fold / QKV concatenation of
            # _load_weights were enqueued on the caller's stream and nothing orders them against the encoder stream yet
            enc.wait_event(entry)
            self._weights_dirty = False
        if host_src is None:
            if isinstance(ready, torch.cuda.Event):
                enc.wait_event(ready)
            elif ready is not True:
                enc.wait_event(entry)                   # whatever produced the batch on the caller's stream comes first
        self._prev_entry = entry
        self._parity ^= 1
        self._consumer_stream = main
        switched = []

        def switch():               # unfrozen top layers (_first_trainable): the rest of the encoder runs on the caller's stream
            done = torch.cuda.Event()
            done.record(enc)
            torch.cuda.set_stream(main)
            main.wait_event(done)
            if self.before_trainable is not None:
                self.before_trainable()
            switched.append(True)

        self._switch = switch if self.train_layers is not None else None
        self._ov = (enc, entry)
        try:
            if host_src is not None:
                # a host batch: copied on the high-priority "h2d" stream (a hardware queue of its own: queued on the encoder's, the
                # copy would only start when the encoder in front of it has finished), the encoder stream waits for its event.  A pinned
                # source is asynchronous; a pageable one blocks the HOST for the copy - INTEGRATION.md: pin it
                # (data.collate_general(pin_memory=True) / transfer_batch_to_device do)
                cs = ops.shared_stream("h2d", self._dev, priority=-1)
                with torch.cuda.stream(cs):
                    padded = host_src.to(self._dev, torch.float32, non_blocking=True)
                    copied = torch.cuda.Event()
                    copied.record(cs)
                enc.wait_event(copied)
            with torch.cuda.stream(enc):
                padded.record_stream(enc)               # the caller's tensor / the copy above, read on this stream
                pl = self._encode(padded, lens, save, L=L, off=off)
            if not switched:
                done = torch.cuda.Event()
                done.record(enc)
                main.wait_event(done)
        finally:
            self._consumer_stream = None
            self._switch = None
            self._ov = None
        if pl.seg is not None:                          # layout tables uploaded on the encoder's stream, read by the step's kernels too
            pl.seg._dev.record_stream(main)
        return pl

    def segment_pitches(self, T: int, valid: Sequence[int], feat_len: Sequence[int], ragged: bool) -> Tuple[List[int], List[int]]:
        """-> (rows needed per utterance, pitch per utterance).  An utterance needs the frames the HuBERT key mask admits
        (``valid``, fairseq forward_padding_mask) and the frames the head / branches read (``feat_len`` + ``tail_rows``), never
        more than the padded length T; its pitch is that + 1 (the conv stack's last row of a segment is scratch) rounded to 8."""
        if not ragged:
            return [T] * len(valid), [_roundup(T + 1, ops.RowSegments.GRAN)] * len(valid)
        need = [max(1, min(T, max(int(v), int(f) + self.tail_rows))) for v, f in zip(valid, feat_len)]
        return need, [_roundup(n + 1, ops.RowSegments.GRAN) for n in need]

    @torch.no_grad()
    def _encode(self, padded: torch.Tensor, wav_len: List[int], save: bool = False, ragged: Optional[bool] = None,
                L: Optional[int] = None, off=None) -> _Plan:
        """customHubertForward + patched extract_features (speech_encoder_plus.py:29-107) on the device.
        ``padded``: the caller's [B, >= L] fp32 device batch, read in place (row stride = its own); ``L``: the padded length of the
        batch the reference would have built (default: the tensor's width); ``off``: per-utterance start samples of the in-forward
        training crop (host list -> uploaded with the step's other integers, or a device int64 tensor), ``wav_len`` then holds the
        CROPPED lengths."""
        a, w = self.arch, self._w
        B = padded.shape[0]
        L = int(padded.shape[1] if L is None else L)
        pl = self._plan(B, L)
        self._claim(pl)
        pl.generation += 1
        pl.src_L = L
        C, D, F, H = a.conv_dim, a.embed_dim, a.ffn_dim, a.heads
        R, M, T = pl.R, pl.M, pl.T
        # every length-derived integer of the step goes up in ONE asynchronous copy from pinned memory BEFORE the kernels are
        # enqueued (a pageable host -> device copy later in the stream would block the host until the GPU got there and
        # leave the head / loss / optimiser launch-bound):  [wav_len | valid frames | feat_len]
        # fairseq forward_padding_mask: frame t valid iff t*(L//T) < len                (:81-82)
        # feat_len = min(round(len / 320), T), python round = half to even              (:604-611)
        chunk = L // T
        if isinstance(wav_len, torch.Tensor):
            # lengths that exist on the DEVICE only (the reference's batch after Lightning moved it, DP scatter): no host read - the
            # length arithmetic runs in a few launches on B integers and the rows keep the uniform pitch (a ragged layout needs the
            # lengths on the host before the first launch; hand ``wav_len`` over as a CPU tensor / list for that, as collate does)
            l64 = wav_len.to(device=self._dev, dtype=torch.int64).clamp(min=0, max=L)
            valid_d = torch.clamp((l64 + (chunk - 1)) // chunk, max=T)
            feat_d = torch.clamp(torch.round(l64.double() / self.downsample_rate).long(), max=T)       # torch.round: half to even
            if pl.seg_mode:
                need, pitch = self.segment_pitches(T, [T] * B, [T] * B, False)
                pl.bind(pl.segments(pitch, [T] * B, self._dev, keys_known=False))
                pl.need = need
                R, M = pl.R, pl.M
                pl.alg_rows_l = [B * t for t in pl.T_l]
                pl.alg_attn_flops = 4.0 * D * B * float(T) ** 2
            pl.feat_len = feat_d
            pl.feat_len._sc_p1_i32 = (feat_d + 1).to(torch.int32)
            ints = (l64, valid_d.to(torch.int32))
            assert off is None or isinstance(off, torch.Tensor)
            pl.wav_off = off
        else:
            valid = [min(T, -(-int(l) // chunk)) for l in wav_len]
            feat_len = [min(round(int(l) / self.downsample_rate), T) for l in wav_len]
            if pl.seg_mode:
                # the row layout of THIS batch: pitches from the lengths, tables uploaded with the batch's other integers (a batch with
                # the layout of the previous one - every equal-length batch - re-uses its tables: _Plan.segments)
                need, pitch = self.segment_pitches(T, valid, feat_len, self.ragged if ragged is None else ragged)
                pl.bind(pl.segments(pitch, valid, self._dev))
                pl.need = need
                R, M = pl.R, pl.M
                # algorithmic work of this batch: every utterance at its OWN length (padding is not work)
                own = [conv_out_lengths(max(int(l), 400), a) for l in wav_len]
                pl.alg_rows_l = [sum(o[i] for o in own) for i in range(len(a.conv_kernels))]
                pl.alg_attn_flops = 4.0 * D * sum(float(o[-1]) ** 2 for o in own)
            # ONE pinned buffer, ONE copy: [wav_len i64 | feat_len i64 | round(feat_len / 20) i64 | valid i32 | feat_len + 1 i32] (the third =
            # the keyword-count targets of the cascaded branches, kwClip.py:876; the last = the key count of the parallel head's
            # [CLS ; frames] as the int32 vector its kernels take); the device tensors below are views of it
            # a fourth int64 row: the crop offsets (zeros without a crop)
            tgt20 = [int(v) for v in np.round(np.asarray(feat_len, dtype=np.float32) / np.float32(20.0))]      # = kw_branches.target_len_host
            assert off is None or (not isinstance(off, torch.Tensor) and len(off) == B)
            # straight from torch's caching pinned-host allocator (no cudaHostAlloc, no pageable staging copy per step; the allocator
            # keeps the block until the asynchronous copy below has finished, so dropping ``hb`` right away is safe)
            hb = torch.empty(40 * B, dtype=torch.uint8, pin_memory=self._dev.type == "cuda")
            hv = hb.numpy()
            hv[: 32 * B].view(np.int64)[:] = np.asarray([list(map(int, wav_len)), feat_len, tgt20, list(off) if off is not None else [0] * B],
                                                        dtype=np.int64).reshape(-1)
            hv[32 * B:].view(np.int32)[:] = np.asarray([valid, [f + 1 for f in feat_len]], dtype=np.int32).reshape(-1)
            db = hb.to(self._dev, non_blocking=True)
            if self._consumer_stream is not None:       # allocated on the encoder's stream, read by the step's stream
                db.record_stream(self._consumer_stream)
            i64, i32 = db[: 32 * B].view(torch.int64), db[32 * B:].view(torch.int32)
            pl.wav_off = i64[3 * B: 4 * B] if off is not None else None
            pl.feat_len = i64[B: 2 * B]
            pl.feat_len._sc_host = list(feat_len)       # host twin (length-derived integers are known before any kernel runs)
            pl.feat_len._sc_p1_i32 = i32[B:]
            pl.feat_len._sc_target20 = i64[2 * B: 3 * B]      # (feat_len / 20).round().long(), with its own host twin
            pl.feat_len._sc_target20._sc_host = tgt20
            ints = (i64[:B], i32[:B])
        if _USE_GRAPH:                                   # a captured graph holds the plan's own buffers
            pl.len_dev.copy_(ints[0])
            pl.valid.copy_(ints[1])
        else:
            pl.len_dev, pl.valid = ints[0], ints[1]
        # The frozen encoder is a fixed sequence of ~130 launches over the plan's resident buffers.  Opt-in (SC_ENCODER_GRAPH=1):
        # after one eager pass (which also sets the kernels' LDS attributes) it is captured into a hipGraph and replayed - one
        # launch instead of ~130.  Measured +-0 on one GPU (the host already runs a full step ahead of the device and the ~3 us
        # gaps between kernels are the hardware's dispatch, not the host's), so it stays off by default; not used with
        # unfrozen layers nor while bench.py's per-kernel event timer is attached.
        use_graph = (_USE_GRAPH and self._dev.type == "cuda" and self.train_layers is None and ops._timer is None
                     and not self._dropout_active() and not pl.seg_mode and pl.wav_off is None and padded.shape[1] == L)
        if not use_graph:
            self._encode_kernels(pl, padded, save)
            return pl
        pl.wav_in.copy_(padded)
        if pl.graph is None:
            if pl.graph_warm == 0:
                pl.graph_warm = 1
                self._encode_kernels(pl, pl.wav_in, False)
                return pl
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._encode_kernels(pl, pl.wav_in, False)
            pl.graph = g
        pl.graph.replay()
        return pl

    def _dropout_active(self) -> bool:
        a = self.arch
        return bool(self.training and self.hubert_dropout
                    and max(a.dropout, a.attention_dropout, a.activation_dropout, a.dropout_input) > 0.0)

    def _dropout_seeds(self):
        """Per-forward dropout seeds, one per site: a function of torch's seed (torch.manual_seed / seed_everything make a run
        reproducible), the rank and the number of train-mode forwards so far.  The masks themselves are stateless hashes of
        (element index, seed) evaluated inside the kernels (csrc/sc_common.h sc_keep8) - nothing is stored."""
        if not self._dropout_active():
            return None
        self._drop_calls += 1
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        base = _mix32(_mix32(torch.initial_seed() & 0xffffffff) ^ _mix32(0x9E3779B9 * (rank + 1)) ^ self._drop_calls)
        return lambda site: _mix32(base + 0x85EBCA6B * (site + 1))

    @torch.no_grad()
    def _encode_kernels(self, pl: _Plan, padded: torch.Tensor, save: bool) -> None:
        a, w = self.arch, self._w
        seeds = self._dropout_seeds()           # None in eval mode: every drop_p below is 0
        p_in, p_res, p_att = ((a.dropout_input, a.dropout, a.attention_dropout) if seeds else (0.0, 0.0, 0.0))
        if seeds and a.activation_dropout > 0.0:
            raise NotImplementedError("activation_dropout > 0 (0.0 in both released HuBERT configs)")
        sd = seeds if seeds else (lambda site: 0)
        B, L = pl.B, pl.L
        C, D, F, H = a.conv_dim, a.embed_dim, a.ffn_dim, a.heads
        R, M, T = pl.R, pl.M, pl.T
        len_dev = pl.len_dev
        if self._section_ev is not None:
            self._section_ev["start"] = torch.cuda.Event(enable_timing=True)
            self._section_ev["start"].record()
        # a1: (optional) utterance layer-norm + zero pad                                (:506-518)
        if pl.seg is not None:
            ops.wav_prep_seg(padded, len_dev, pl.wav_pad, pl.seg, pl.spr, a.normalize_wav, L=pl.src_L, wav_off=pl.wav_off)
        else:
            ops.wav_prep(padded, len_dev, pl.wav_pad, a.normalize_wav, L=pl.src_L, wav_off=pl.wav_off)
        scale = (D // H) ** -0.5
        train_front = self.frontend is not None      # (also in eval: the frozen copies in self._w are the INITIAL weights)
        if train_front:           # fully trainable encoder: the front end keeps its activations (hubert_frontend_train.py)
            self.frontend.refresh()
            self.frontend.forward_frontend(pl, L, p_in, p_res, sd(0), sd(1))
        else:
            self._frontend_frozen(pl, w, seeds, sd, p_in, p_res, padded)
        self._layers(pl, w, seeds, sd, p_res, p_att, save, scale, first_hidden_done=train_front)

    @torch.no_grad()
    def _frontend_frozen(self, pl, w, seeds, sd, p_in, p_res, padded=None) -> None:
        a = self.arch
        B, L = pl.B, pl.L
        C, D, F, H = a.conv_dim, a.embed_dim, a.ffn_dim, a.heads
        R, M, T = pl.R, pl.M, pl.T
        seg = pl.seg
        nl = len(a.conv_kernels)
        # a2: conv feature extractor                                                    (:75)
        ln_mode = a.extractor_mode == "layer_norm"
        if seg is not None:
            # ragged rows: conv 0 writes the rows of the segment layout only; the GroupNorm statistics still run over the padded
            # batch length T_0 (fairseq feeds the zero-padded batch) and come from the caller's batch masked by its lengths
            if ln_mode:
                ops.conv0_layernorm_gelu_seg(pl.wav_pad, seg, pl.spr, w["conv0_w"], w["conv0_bias"], w["conv0_ln_g"], w["conv0_ln_b"], pl.conv[0])
            else:
                ops.conv0_groupnorm_gelu_seg(padded, pl.len_dev, pl.wav_pad, seg, pl.spr, w["conv0_w"], w["gn_g"], w["gn_b"], pl.T_l[0], pl.conv[0],
                                             wav_off=pl.wav_off)
        elif ln_mode:       # large: conv (+bias) -> LayerNorm(512) -> GELU after every layer
            ops.conv0_layernorm_gelu(pl.wav_pad, w["conv0_w"], w["conv0_bias"], w["conv0_ln_g"], w["conv0_ln_b"],
                                     pl.R_l[0], pl.conv[0])
        else:             # base: GroupNorm(512, 512) over time after conv 0 only, no conv bias
            ops.conv0_groupnorm_gelu(pl.wav_pad, w["conv0_w"], w["gn_g"], w["gn_b"], pl.T_l[0], pl.R_l[0], pl.conv[0])
        alg = pl.alg_rows_l if seg is not None else [B * t for t in pl.T_l]     # rows that are algorithmic work (bench.py's timer)
        for i in range(1, nl):
            k, s = a.conv_kernels[i], a.conv_strides[i]
            rows = M * (2 ** (nl - 1 - i)) if seg is not None else B * pl.R_l[i]
            ops.gemm_raw(pl.conv[i - 1], s * C, w[f"conv{i}_w"], k * C, pl.conv[i], C, rows, C, k * C,
                         bias=w[f"conv{i}_bias"], act=0 if ln_mode else 1, alg_rows=alg[i],
                         tap_c=C if (k == 3 and s == 2) else 0)      # shared-tap K order: see sc_gemm_args.tap_c
            if ln_mode:
                ops.layernorm_bf16(pl.conv[i][:rows], w[f"conv{i}_ln_g"], w[f"conv{i}_ln_b"], out=pl.conv[i][:rows], act=1)
        if self._section_ev is not None:
            self._section_ev["conv_done"] = torch.cuda.Event(enable_timing=True)
            self._section_ev["conv_done"].record()
        # a3: LayerNorm(512) -> post_extract_proj                                       (:78, :84-85)
        ops.layernorm_bf16(pl.conv[-1][:M], w["ln_feat_g"], w["ln_feat_b"], out=pl.feat_ln)
        ops.linear_bf16(pl.feat_ln, w["proj_w"], w["proj_b"], out=pl.x_proj, alg_rows=alg[-1], drop_p=p_in, drop_seed=sd(0))   # dropout_input (:87)
        # a4: zero padded frames, grouped pos_conv + GELU, residual, LayerNorm          (:32-40)
        G, Kp = a.pos_conv_groups, a.pos_conv_kernel
        Dg, Rp = D // G, R + 2 * pl.halo
        if seg is not None:
            assert Dg in (48, 64) and Kp == 128, "the segment layout runs pos_conv on the slab kernel (48 / 64 channels per group, 128 taps)"
            ops.posconv_prep_seg(pl.x_proj, pl.valid, pl.xz, pl.xg, seg, D, G, pl.halo)
            ops.posconv_seg(pl.xg, w["pos_w"], w["pos_b"], pl.xz, pl.pre, seg, D, G, Kp, alg_rows=alg[-1])
            return
        ops.posconv_prep(pl.x_proj, pl.valid, pl.xz, pl.xg, B, R, D, G, pl.halo)
        if Dg in (48, 64) and Kp == 128:
            # the input slab of a (group, utterance) stays in LDS for all 128 taps (csrc/posconv.hip); same arithmetic as the GEMM below
            ops.posconv(pl.xg, w["pos_w"], w["pos_b"], pl.xz, pl.pre, B, R, D, G, Kp, alg_rows=T)
        else:
            ops.gemm_raw(pl.xg, Dg, w["pos_w"], Kp * Dg, pl.pre, D, R, Dg, Kp * Dg, bias=w["pos_b"], residual=pl.xz, ldr=D,
                         act=1, nb1=G, nb2=B, sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, R * D),
                         sBias=(Dg, 0), sR=(Dg, R * D), alg_rows=T)

    @torch.no_grad()
    def _first_trainable(self, tl) -> bool:
        """In front of the first unfrozen layer: hand over from the encoder stream to the caller's (overlapped schedule: the layers
        below are frozen and ran a step ahead; from here on the kernels read parameters, so the caller's stream joins the optimiser's
        first), then rebuild the layers' bf16 working copies."""
        if self._switch is not None:
            self._switch()
        tl.refresh()
        return True

    def _layers(self, pl, w, seeds, sd, p_res, p_att, save, scale, first_hidden_done) -> None:
        a = self.arch
        B, L = pl.B, pl.L
        C, D, F, H = a.conv_dim, a.embed_dim, a.ffn_dim, a.heads
        R, M, T = pl.R, pl.M, pl.T

        seg = pl.seg
        alg_M = pl.alg_rows_l[-1] if seg is not None else B * T                       # rows / flops that are algorithmic work
        alg_att = pl.alg_attn_flops if seg is not None else 4.0 * B * T * T * D

        def qkv_attn(x, i):
            ops.gemm_raw(x, D, w[f"l{i}_qkv_w"], D, pl.qk, 2 * D, M, 3 * D, D, bias=w[f"l{i}_qkv_b"], Ct=pl.vt,
                         n_split=2 * D, R=R, dh=D // H, alg_rows=alg_M, seg=seg)
            ops.attn_fwd(pl.qk, pl.vt, pl.valid, pl.ctx, B, R, H, D, scale, alg_flops=alg_att,
                         drop_p=p_att, drop_seed=sd(3 * i + 2), seg=seg)

        if not a.layer_norm_first:
            # a5: post-LN layers (base): x = LN1(x + attn(x)); x = LN2(x + ffn(x))       (:39-40, :49-53)
            if not first_hidden_done:
                ops.layernorm_bf16(pl.pre, w["ln_enc_g"], w["ln_enc_b"], out=pl.hidden[0])
                if p_res > 0:
                    ops.dropout_bf16(pl.hidden[0], p_res, sd(1), out=pl.hidden[0])             # F.dropout after the LN (:42)
            tl = self.train_layers
            refreshed = tl is None
            pl.lazy = None
            if _FUSED_LN and tl is None and self._dev.type == "cuda":
                self._layers_fused(pl, w, sd, p_res, p_att, scale)
                return
            for i in range(a.layers):
                x = pl.hidden[i]
                if tl is not None and (i in tl.ids or i in tl.pass_ids):   # unfrozen (or frozen above an unfrozen one): activations kept
                    if not refreshed:
                        refreshed = self._first_trainable(tl)
                    tl.layer_forward(i, x, pl.hidden[i + 1], pl, save,
                                     drops=(p_res, p_att, sd(3 * i + 2), sd(3 * i + 3), sd(3 * i + 4)) if seeds else None)
                    continue
                if ops._timer is None:          # one C-ABI call per frozen layer (sc_hubert_layer_fwd); the per-op path below is
                    ops.hubert_layer_fwd(x, pl.hidden[i + 1], pl.valid, w, i, pl, B, R, T, D, F, H, False, p_att, p_res,
                                         (sd(3 * i + 2), sd(3 * i + 3), sd(3 * i + 4)), seg=seg)   # kept for bench.py's per-kernel timer
                    continue
                qkv_attn(x, i)
                # train mode: dropout1 / dropout3 of fairseq's TransformerSentenceEncoderLayer in the GEMM epilogues (before the
                # residual add), attention dropout inside the attention kernel
                ops.linear_bf16(pl.ctx, w[f"l{i}_o_w"], w[f"l{i}_o_b"], out=pl.pre, residual=x, alg_rows=alg_M,
                                drop_p=p_res, drop_seed=sd(3 * i + 3))
                ops.layernorm_bf16(pl.pre, w[f"l{i}_ln1_g"], w[f"l{i}_ln1_b"], out=pl.x1)
                ops.linear_bf16(pl.x1, w[f"l{i}_fc1_w"], w[f"l{i}_fc1_b"], out=pl.ffn, act=1, alg_rows=alg_M)
                ops.linear_bf16(pl.ffn, w[f"l{i}_fc2_w"], w[f"l{i}_fc2_b"], out=pl.pre, residual=pl.x1, alg_rows=alg_M,
                                drop_p=p_res, drop_seed=sd(3 * i + 4))
                ops.layernorm_bf16(pl.pre, w[f"l{i}_ln2_g"], w[f"l{i}_ln2_b"], out=pl.hidden[i + 1])
        else:
            # pre-LN layers (large): x = x + attn(LN1(x)); x = x + ffn(LN2(x)); layer_results are NOT passed through
            # the encoder's final LayerNorm (fairseq applies it to `x` only, which the reference never reads)
            if not first_hidden_done:
                pl.hidden[0].copy_(pl.pre)
                if p_res > 0:
                    ops.dropout_bf16(pl.hidden[0], p_res, sd(1), out=pl.hidden[0])
            tl = self.train_layers
            refreshed = tl is None
            for i in range(a.layers):
                x = pl.hidden[i]
                if tl is not None and (i in tl.ids or i in tl.pass_ids):
                    if not refreshed:
                        refreshed = self._first_trainable(tl)
                    tl.layer_forward(i, x, pl.hidden[i + 1], pl, save,
                                     drops=(p_res, p_att, sd(3 * i + 2), sd(3 * i + 3), sd(3 * i + 4)) if seeds else None)
                    continue
                if ops._timer is None:
                    ops.hubert_layer_fwd(x, pl.hidden[i + 1], pl.valid, w, i, pl, B, R, T, D, F, H, True, p_att, p_res,
                                         (sd(3 * i + 2), sd(3 * i + 3), sd(3 * i + 4)), seg=seg)
                    continue
                ops.layernorm_bf16(x, w[f"l{i}_ln1_g"], w[f"l{i}_ln1_b"], out=pl.x1)
                qkv_attn(pl.x1, i)
                ops.linear_bf16(pl.ctx, w[f"l{i}_o_w"], w[f"l{i}_o_b"], out=pl.pre, residual=x, alg_rows=alg_M,
                                drop_p=p_res, drop_seed=sd(3 * i + 3))
                ops.layernorm_bf16(pl.pre, w[f"l{i}_ln2_g"], w[f"l{i}_ln2_b"], out=pl.x1)
                ops.linear_bf16(pl.x1, w[f"l{i}_fc1_w"], w[f"l{i}_fc1_b"], out=pl.ffn, act=1, alg_rows=alg_M)
                ops.linear_bf16(pl.ffn, w[f"l{i}_fc2_w"], w[f"l{i}_fc2_b"], out=pl.hidden[i + 1], residual=pl.pre,
                                alg_rows=alg_M, drop_p=p_res, drop_seed=sd(3 * i + 4))

    @torch.no_grad()
    def _layers_fused(self, pl, w, sd, p_res, p_att, scale) -> None:
        """a5 without LayerNorm launches (frozen post-LN layers): hidden[i + 1] receives the rows in front of layer i's final
        LayerNorm and pl.stats[i + 1] their statistics; consumers (the next layer, the weighted sum) normalise on the fly."""
        a = self.arch
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        D, F, H = a.embed_dim, a.ffn_dim, a.heads
        ns = ops.gemm_stats_strips(M, D)
        for i in range(a.layers):
            x, out = pl.hidden[i], pl.hidden[i + 1]
            xs = pl.stats[i] if i > 0 else None
            seeds = (sd(3 * i + 2), sd(3 * i + 3), sd(3 * i + 4))
            if ops._timer is None:
                ops.hubert_layer_fwd(x, out, pl.valid, w, i, pl, B, R, T, D, F, H, False, p_att, p_res, seeds, fused=(xs, ns, pl.stats[i + 1]))
                continue
            # the same sequence op by op (bench.py's per-kernel timer)
            ln = dict(ln_eps=1e-5)
            if xs is None:
                ops.gemm_raw(x, D, w[f"l{i}_qkv_w"], D, pl.qk, 2 * D, M, 3 * D, D, bias=w[f"l{i}_qkv_b"], Ct=pl.vt, n_split=2 * D, R=R,
                             dh=D // H, alg_rows=B * T)
            else:
                ops.gemm_raw(x, D, w[f"l{i}_qkv_wf"], D, pl.qk, 2 * D, M, 3 * D, D, bias=w[f"l{i}_qkv_cf"], Ct=pl.vt, n_split=2 * D, R=R,
                             dh=D // H, alg_rows=B * T, ln_stats=xs, ln_ns=ns, ln_colsum=w[f"l{i}_qkv_sf"], **ln)
            ops.attn_fwd(pl.qk, pl.vt, pl.valid, pl.ctx, B, R, H, D, scale, alg_flops=4.0 * B * T * T * D, drop_p=p_att, drop_seed=seeds[0])
            res = {} if xs is None else dict(res_stats=xs, res_ns=ns, res_gamma=w[f"l{i - 1}_ln2_g"], res_beta=w[f"l{i - 1}_ln2_b"])
            ns1 = ops.gemm_raw(pl.ctx, D, w[f"l{i}_o_w"], D, pl.pre, D, M, D, D, bias=w[f"l{i}_o_b"], residual=x, ldr=D, alg_rows=B * T,
                               drop_p=p_res, drop_seed=seeds[1], stats_out=pl.stats1, **res, **ln)
            ops.gemm_raw(pl.pre, D, w[f"l{i}_fc1_wf"], D, pl.ffn, F, M, F, D, bias=w[f"l{i}_fc1_cf"], act=1, alg_rows=B * T,
                         ln_stats=pl.stats1, ln_ns=ns1, ln_colsum=w[f"l{i}_fc1_sf"], **ln)
            ops.gemm_raw(pl.ffn, F, w[f"l{i}_fc2_w"], F, out, D, M, D, F, bias=w[f"l{i}_fc2_b"], residual=pl.pre, ldr=D, alg_rows=B * T,
                         drop_p=p_res, drop_seed=seeds[2], stats_out=pl.stats[i + 1], res_stats=pl.stats1, res_ns=ns1,
                         res_gamma=w[f"l{i}_ln1_g"], res_beta=w[f"l{i}_ln1_b"], **ln)
        pl.lazy = ops.LazyStates(pl.stats, w["lazy_gamma"], w["lazy_beta"], first_lazy=1, ns=ns, eps=1e-5)

    def _materialised_states(self, pl) -> tuple:
        """Hidden states as the reference returns them (fresh [B, T, D] tensors).  With the LayerNorm-free layers states 1.. are raw
        rows: their LayerNorm runs here, on request only (the weighted sum normalises on the fly).  Segment layout: one row gather
        per state; frames t >= rows-needed of an utterance (``_Plan.need``; all T when the forward ran un-ragged) are zero."""
        B, R, T, D = pl.B, pl.R, pl.T, self.arch.embed_dim
        if pl.seg is not None:
            idx, keep = self._gather_index(pl)
            return tuple(pl.hidden[n].index_select(0, idx).view(B, T, D) * keep for n in range(self.arch.layers + 1))
        if pl.lazy is None:
            return tuple(pl.hidden[n].view(B, R, D)[:, :T].clone() for n in range(self.arch.layers + 1))
        out = [pl.hidden[0].view(B, R, D)[:, :T].clone()]
        for n in range(1, self.arch.layers + 1):
            y = ops.layernorm_bf16(pl.hidden[n], self._w[f"l{n - 1}_ln2_g"], self._w[f"l{n - 1}_ln2_b"])
            out.append(y.view(B, R, D)[:, :T].clone())
        return tuple(out)

    def _gather_index(self, pl):
        """row index [B * T] of frame (b, t) in the segment layout (row 0 for frames the layout does not hold) + their 0 / 1 mask"""
        T, seg = pl.T, pl.seg
        t = torch.arange(T).unsqueeze(0)
        need = torch.tensor(pl.need).unsqueeze(1)
        r0 = torch.tensor(seg.row0_host[:-1]).unsqueeze(1)
        keep = t < need
        idx = torch.where(keep, r0 + t, torch.zeros_like(t)).reshape(-1)
        return (idx.to(self._dev, non_blocking=True),
                keep.reshape(pl.B, T, 1).to(device=self._dev, dtype=torch.bfloat16, non_blocking=True))

    def forward(self, wav: Union[torch.Tensor, list], wav_len: Union[torch.Tensor, list] = [],
                feat_select_idx: Union[str, list] = None, return_hidden_states: bool = False) -> Tuple:
        # :539-554.  A padded (B, width) batch - what collate_general builds and Lightning moves - goes to the kernels AS IS, with or
        # without the training crop: the two kernels that read it (sc_wav_prep*_crop, sc_conv0_stats_len_crop) take the padded length,
        # the lengths and the crop offsets and read utterance b from wav[b, off_b : off_b + len_b] in place; the reference's un-pad /
        # crop / re-pad (:539-552, :506-518) never happens as data movement.  A list of waveforms is padded once.
        crop = self.training and self.max_audio_len >= 0
        host_src, off, ready = None, None, None
        if isinstance(wav, torch.Tensor) and wav.dim() == 1:
            wav = wav.unsqueeze(0)
        if isinstance(wav, torch.Tensor) and wav.dim() == 2:
            ready = getattr(wav, "_sc_ready", None)
            width = int(wav.shape[1])
            lens = None
            if isinstance(wav_len, torch.Tensor) and wav_len.numel() > 0:
                lens = getattr(wav_len, "_sc_host", None)            # host twin (data.attach_host_lengths / transfer_batch_to_device)
                if lens is None:
                    lens = wav_len if wav_len.is_cuda else [int(l) for l in wav_len.tolist()]
            elif not isinstance(wav_len, torch.Tensor) and len(wav_len) > 0:
                lens = [int(l) for l in wav_len]
            if lens is None:
                lens = [width] * wav.shape[0]
            if isinstance(lens, torch.Tensor):
                # lengths on the DEVICE only (Lightning's transferred batch without a host twin): never read back.  The crop is then
                # drawn on the device from host uniforms (one per utterance: same distribution as the reference's randint, not the
                # same stream of draws), the rows keep the uniform pitch of the padded length
                L = width
                if crop and width > self.max_audio_len:
                    l64 = lens.to(device=self._dev, dtype=torch.int64).clamp(min=0, max=width)
                    uh = torch.empty(wav.shape[0], dtype=torch.float64, pin_memory=self._dev.type == "cuda")
                    uh.numpy()[:] = np.random.random_sample(wav.shape[0])
                    u = uh.to(self._dev, non_blocking=True)
                    room = (l64 - self.max_audio_len).clamp(min=0)
                    off = torch.minimum((u * room.double()).long(), (room - 1).clamp(min=0)).contiguous()
                    lens = torch.minimum(l64, torch.full_like(l64, self.max_audio_len))
                    L = self.max_audio_len
            else:
                lens = [min(int(l), width) for l in lens]
                if crop and max(lens) > self.max_audio_len:
                    off, lens = crop_windows(lens, self.max_audio_len)
                L = max(max(lens), 1)
            if not wav.is_cuda and self._dev.type == "cuda":
                # copied to the device on the stream the encoder runs on; whole rows (a sliced host view would be staged through a
                # pageable temporary and block), the kernels read the first L samples / the crop windows
                host_src, padded = (wav if wav.dtype == torch.float32 else wav.float()), None
            else:
                padded = wav.to(self._dev, torch.float32)
                if padded.stride(1) != 1 or (padded.shape[0] > 1 and padded.stride(0) < padded.shape[1]):
                    padded = padded.contiguous()           # also overlapping / zero row strides (wav.expand(B, L), as_strided views)
                if padded.data_ptr() != wav.data_ptr():
                    ready = None                       # a fresh tensor produced on the caller's stream just now
        else:
            wav = list(wav)
            if crop:                                                                   # :548-552
                wav = [random_crop_max_length(wav[b], self.max_audio_len, len(wav[b])) for b in range(len(wav))]
            lens = [len(w) for w in wav]
            L = max(lens)
            padded = torch.zeros(len(wav), L, device=self._dev, dtype=torch.float32)
            for b, x in enumerate(wav):
                padded[b, : lens[b]] = x.to(self._dev, torch.float32)
        save = self.train_layers is not None and self.training and torch.is_grad_enabled()
        # the overlapped schedule (see _encode_overlapped): a frozen encoder entirely, unfrozen TOP layers up to the first of them
        tl = self.train_layers
        ahead = (self.enc_overlap and not _USE_GRAPH and self.training and torch.is_grad_enabled() and self._dev.type == "cuda"
                 and not isinstance(lens, torch.Tensor)
                 and (tl is None or (self.frontend is None and save)))
        if self.train_layers is not None and self.before_trainable is not None and not ahead:
            # unfrozen layers read their parameters (refresh(): bf16 copies; LayerNorm affine and biases alias the masters) inside
            # the encoder: join the optimiser's side stream BEFORE the first kernel, not at the weighted sum
            self.before_trainable()
        # views of the plan's resident workspace; every PUBLIC return path below hands out clones (the reference returns fresh
        # tensors: holding encoder outputs across calls must be safe), the weighted-sum fast path reads the workspace in place
        want_states = return_hidden_states or (feat_select_idx or self.feat_select_idx) != FEAT_SELECT_IDX_WEIGHTED_SUM_MODE
        # returned hidden states carry every padded row, as the reference's do: that forward runs un-ragged (all B x T frames)
        if ahead and not want_states:
            pl = self._encode_overlapped(padded, host_src, lens, ready, save, L=L, off=off)
        else:
            if ahead and self.train_layers is not None and self.before_trainable is not None:
                self.before_trainable()
            if padded is None:
                padded = host_src.to(self._dev, torch.float32, non_blocking=True)
            elif isinstance(ready, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(ready)         # a prefetcher's copy-done event
            pl = self._encode(padded, lens, save, ragged=False if want_states else None, L=L, off=off)
        B, R, T, D = pl.B, pl.R, pl.T, self.arch.embed_dim
        # (without want_states the tuple is only a placeholder: the weighted sum reads the plan's workspace, raw or not, in place)
        hidden_states = self._materialised_states(pl) if want_states else ((None,) * (self.arch.layers + 1) if pl.seg is not None else tuple(
            pl.hidden[n].view(B, R, D)[:, :T] for n in range(self.arch.layers + 1)))
        feat = {"last_hidden_state": hidden_states[-1], "hidden_states": hidden_states}
        feat_len = pl.feat_len                                                          # :604-611 (uploaded in _encode)
        if feat_select_idx is None:
            feat_select_idx = self.feat_select_idx
        return_list = []
        if feat_select_idx == "all":
            return_list.extend([feat, feat_len])
        elif feat_select_idx == FEAT_SELECT_IDX_WEIGHTED_SUM_MODE:
            if self.before_trainable is not None:
                self.before_trainable()                 # train.ContrastiveTrainer: join the optimiser's side stream here
            ws_feat = self.weightedsum_layer.forward_padded(pl.hidden, B, pl.Rout, T, D, plan=pl)
            if save:                                    # the head's backward hands dX to the unfrozen layers (hubert_train.py)
                tl = self.train_layers
                ws_feat._sc_handle.layers_bwd = lambda dX, w_soft, _pl=pl: tl.backward(
                    _pl, dX, w_soft, normalize=self.weightedsum_layer.normalize_features)
            return_list.extend([ws_feat, feat_len])
        elif isinstance(feat_select_idx, list):
            return_list.extend([[feat["hidden_states"][i] for i in feat_select_idx], feat_len])
        elif feat_select_idx in feat:
            return_list.extend([feat[feat_select_idx], feat_len])
        else:
            raise KeyError(feat_select_idx)
        if return_hidden_states:
            return_list.append(feat["hidden_states"])
        return tuple(return_list)
