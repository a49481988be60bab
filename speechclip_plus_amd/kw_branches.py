"""Branch heads: mirror of avssl/model/kw_branches.py for the hot path.

``KW_ParallelBranch`` (kw_branches.py:200-282): 1 CLS token + 1-layer TransformerEncoder + Linear D -> E.
The committed reference constructor assigns ``None`` over ``self_att``; the intended computation
(:266-280) is what is built.  Same forward signature as the caller uses (``audio_feat``, ``audio_feat_len``
keyword at kwClip.py:881-884; ``audio_len`` accepted as the positional name the reference declares).
"""
import logging
from collections import defaultdict
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
from torch import nn

from . import transformer_models as TransformerModels
from . import vector_quantizers
from .cif import CIF
from .linear_fn import linear_f32_autograd
from .projections import MLPLayers
from .vector_quantizers import Kw_BatchNorm_dynamic

logger = logging.getLogger(__name__)


def get_keypadding_mask(max_length: int, data_lens: torch.Tensor, add: int = 0) -> torch.Tensor:
    """avssl/util/data_utils.py:6-22 (True = padding) for the lengths ``data_lens + add``, built on the device of the lengths."""
    if data_lens.is_cuda and data_lens.dtype == torch.int64 and data_lens.dim() == 1:
        from . import ops
        return ops.len_mask(data_lens.contiguous(), max_length, add)     # one launch; carries the lengths (``_sc_lens``)
    return torch.arange(max_length, device=data_lens.device).unsqueeze(0) >= (data_lens + add).unsqueeze(1)


def target_len_host(feat_len_host) -> list:
    """``(feat_len / 20).round().long()`` (kwClip.py:876) on host integers: float32 division, round half to even."""
    import numpy as np
    return [int(v) for v in np.round(np.asarray(feat_len_host, dtype=np.float32) / np.float32(20.0))]


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class GeneralBranch(nn.Module):
    def __init__(self, config, audio_dim: int, text_dim: int) -> None:
        super().__init__()
        logger.info(f"Using {type(self).__name__}")
        self.config = config
        self.audio_dim = audio_dim
        self.text_dim = text_dim

    def _create_self_attn_layer(self, branch_config):
        transformer_args = _get(branch_config, "transformer_args")
        transformer_type = _get(transformer_args, "type", None) or _get(branch_config, "transformer_type")
        logger.info(f"Using {transformer_type} as {type(self).__name__}")
        args = dict(transformer_args) if isinstance(transformer_args, dict) else dict(vars(transformer_args))
        args.pop("type", None)
        self.self_att = getattr(TransformerModels, transformer_type)(**args)

    def _create_cls(self, length: int, cls_dim: int) -> nn.Parameter:
        return torch.nn.Parameter(torch.randn([1, length, cls_dim]))

    def _create_kw_proj_layer(self):
        """kw_branches.py:44-73: Linear d_model -> text_dim, or an MLP when ``keyword.kw_projection`` is given."""
        cb = _get(_get(self.config, "model_settings"), "cascaded_branch")
        d_model = _get(_get(cb, "transformer_args"), "d_model")
        self.kw_projection_config = _get(_get(cb, "keyword"), "kw_projection", None)
        if self.kw_projection_config is None:
            self.linear_proj = nn.Linear(d_model, self.text_dim)
        else:
            dims = list(_get(self.kw_projection_config, "dimensions"))
            assert dims[0] == d_model, f"first dim({dims[0]}) should match the audio encoder dim({d_model})"
            assert dims[-1] == self.text_dim, f"last dim({dims[-1]}) should match the text encoder dim({self.text_dim})"
            self.linear_proj = MLPLayers(units=dims, dropout=_get(self.kw_projection_config, "dropout"))

    def _create_vector_quantizer(self):
        vq = _get(_get(_get(self.config, "model_settings"), "cascaded_branch"), "vq")
        self.vq_type = _get(vq, "type")
        if not hasattr(vector_quantizers, self.vq_type):
            raise NotImplementedError("Vq ({}) not implemented".format(self.vq_type))
        self.vector_quantizer = getattr(vector_quantizers, self.vq_type)(**dict(_get(vq, "args")))

    def project_feats_to_CLIPspace(self, features: torch.Tensor) -> torch.Tensor:
        """kw_branches.py:143-156.  Exact fp32 on the matrix pipe (linear_fn.LinearF32Fn): the projected keywords are compared
        against the whole vocabulary by an argmax, so their operands are not rounded to bf16."""
        if isinstance(self.linear_proj, nn.Linear):
            # (train steps: exact forward, gradient products on the bf16 GEMMs - only the forward decides a token)
            features = linear_f32_autograd(features.float(), self.linear_proj.weight, self.linear_proj.bias, bf16_backward=self.training)
        else:
            features = self.linear_proj(features.float())
        if hasattr(self, "bn_layer"):
            features = self.bn_layer(features)
        return features

    def get_keyword_cosine_score(self, keywords: torch.Tensor) -> torch.Tensor:
        """kw_branches.py:158-179: cosine of every keyword against every (reduced-vocabulary) token embedding, in exact fp32 on
        the matrix pipe (sc_vq_prep_f32 + sc_sgemm_mfma_f32).  Forward only: the training path is ``vq_audio_features``."""
        from . import ops
        from .vector_quantizers import VocabTables
        tb = VocabTables.of(self.clip.model.token_embedding.weight, self.vector_quantizer._tables)
        bsz, n, Et = keywords.shape
        kwn_T, _ = ops.vq_prep(keywords.detach().reshape(bsz * n, Et).float().contiguous())
        cos = ops.sgemm_mfma(kwn_T, tb.norm_T, a_kmajor=True, b_kmajor=True)
        return cos[: bsz * n, : tb.V].reshape(bsz, n, tb.V)

    def vq_audio_features(self, audio_feat: torch.Tensor):
        """kw_branches.py:181-197 as ONE autograd node (vector_quantizers.SimpleVectorQuantizer.quantize_keywords): cosine scores,
        special-token mask, argmax / straight-through softmax and the ``@ token_embedding`` product, forward and backward."""
        audio_feat = self.project_feats_to_CLIPspace(audio_feat)
        assert self.clip.model.token_embedding.weight.requires_grad == False
        return self.vector_quantizer.quantize_keywords(audio_feat, self.clip.model.token_embedding.weight)


class KW_ParallelBranch(GeneralBranch):
    def __init__(self, config, audio_dim: int, text_dim: int) -> None:
        super().__init__(config, audio_dim, text_dim)
        pb = _get(_get(config, "model_settings"), "parallel_branch")
        self._create_self_attn_layer(pb)
        self.cls = self._create_cls(length=1, cls_dim=_get(_get(pb, "transformer_args"), "d_model"))
        self.need_projection = _get(pb, "need_projection", True)
        if self.need_projection:
            self.linear_proj = nn.Linear(self.audio_dim, self.text_dim)

    def extract_hidden_states(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> Tuple:
        """kw_branches.py:223-249: hidden states of the branch layer over [CLS ; frames], CLS position dropped
        (full-sequence path: stock torch ops)."""
        bsz, T = audio_feat.shape[:2]
        src = torch.cat([self.cls.expand(bsz, -1, -1).to(audio_feat.dtype), audio_feat], dim=1)
        pad = get_keypadding_mask(T + 1, audio_len.to(audio_feat.device), add=1)
        return tuple(x[:, 1:, ...] for x in self.self_att.extract_hidden_states(src=src, key_padding_mask=pad))

    def forward(self, audio_feat: torch.Tensor, audio_len: Optional[torch.Tensor] = None, otherInputs: dict = None,
                audio_feat_len: Optional[torch.Tensor] = None) -> dict:
        if audio_len is None:
            audio_len = audio_feat_len
        output = defaultdict(lambda: None)
        # kw_branches.py:266-280: CLS + frames, key_padding_mask from audio_len + 1, row 0, projection
        lens = getattr(audio_len, "_sc_p1_i32", None)           # audio_len + 1 as int32, uploaded with the batch's other integers
        out = self.self_att.cls_forward(self.cls, audio_feat, audio_len + 1 if lens is None else lens,
                                        proj=getattr(self, "linear_proj", None))
        output["parallel_audio_feat"] = out
        return output


class KW_CascadedBranchPlus(GeneralBranch):
    """kw_branches.py:580-777: self-attention over the frames -> CIF downsampling -> keyword projection + BatchNorm ->
    cosine VQ against the CLIP token table -> frozen CLIP text encoder, every stage on the library's kernels (mha_block, cif,
    linear_fn, vector_quantizers, clip_text_hip)."""

    def __init__(self, config, audio_dim: int, text_dim: int, clip) -> None:
        super().__init__(config, audio_dim, text_dim)
        self.clip = clip
        cb = _get(_get(config, "model_settings"), "cascaded_branch")
        self._create_self_attn_layer(cb)
        self._create_kw_proj_layer()
        self._create_vector_quantizer()
        bn = _get(_get(cb, "keyword"), "batchnorms", None)
        if bn is not None:
            emb = self.clip.model.token_embedding.weight
            self.bn_layer = Kw_BatchNorm_dynamic(kw_dim=self.text_dim, init_bias=torch.mean(emb, dim=0),
                                                 init_scale=torch.std(emb, dim=0), std_scale=_get(bn, "std_scale"),
                                                 learnable=_get(bn, "learnable", True))
        ds = _get(cb, "downsampling")
        self.downsampling_type = _get(ds, "type")
        if self.downsampling_type != "cif":
            raise NotImplementedError("Unknown type:{}".format(self.downsampling_type))
        cif_cfg = dict(_get(ds, "cif"))
        self.using_gt_len = cif_cfg.get("using_gt_len", False)
        self.downsampling = CIF(**cif_cfg)

    def downsampling_audio_feat(self, audio_feat, audio_feat_len, audio_feat_pad_mask, global_step: int = 0,
                                target_len: Optional[torch.Tensor] = None, rows=None) -> dict:
        """kw_branches.py:644-699: the CIF target length is used only in training (round(len / 20) if not given).  ``rows``: the
        attention block's output rows (mha_block.BranchRows) instead of ``audio_feat`` - CIF reads them in place."""
        inputs = {"audio_feat": audio_feat, "audio_feat_len": audio_feat_len, "audio_feat_pad_mask": audio_feat_pad_mask,
                  "global_step": global_step}
        if rows is not None:
            inputs["audio_feat_rows"] = rows
        host = None
        if not self.training:
            input_target_len = None
        elif target_len is None:
            input_target_len = (audio_feat_len / 20).round().long()
            lens_host = getattr(audio_feat_len, "_sc_host", None)      # the encoder's host copy of the frame counts
            host = target_len_host(lens_host) if lens_host is not None else None
        else:
            input_target_len = target_len
            host = getattr(target_len, "_sc_host", None)
        # with the targets known on the host the CIF output is sized without a device read (cif.CIF.forward)
        res = self.downsampling(inputs, input_target_len, target_lengths_host=host)
        if target_len is not None:
            res["target_len"] = target_len
            res["dsample_len_diff"] = (res["dsample_feats_length"] - target_len).abs().float().mean()
        return res

    def _rows_path(self, audio_feat: torch.Tensor) -> bool:
        """train steps hand the attention block's bf16 output rows straight to CIF (no fp32 copy, no padded copies, gradients in
        the same layout); eval and the module-level API keep the reference's tensors"""
        return self.training and audio_feat.is_cuda and hasattr(self.self_att, "forward_rows")

    def _tail(self, output, feats, feat_len, pad_mask, otherInputs, rows=None):
        otherInputs = otherInputs or {}
        ds = self.downsampling_audio_feat(audio_feat=feats, audio_feat_len=feat_len, audio_feat_pad_mask=pad_mask,
                                          target_len=otherInputs.get("target_len", None),
                                          global_step=otherInputs.get("global_step", 0), rows=rows)
        output["dsample_results"] = ds
        vq_results, keywords = self.vq_audio_features(ds["dsample_feats"])
        output["vq_results"] = vq_results
        output["keywords"] = keywords
        output["cascaded_audio_feat"] = self.clip.encode_keywords(keywords, ds["dsample_feats_length"])
        return output

    def forward(self, audio_feat: torch.Tensor, audio_feat_len: torch.Tensor, otherInputs: dict = {}) -> dict:
        output = defaultdict(lambda: None)
        pad = get_keypadding_mask(audio_feat.shape[1], audio_feat_len.to(audio_feat.device))
        if self._rows_path(audio_feat):
            rows, _ = self.self_att.forward_rows(src=audio_feat, key_padding_mask=pad)
            return self._tail(output, None, audio_feat_len.to(audio_feat.device), pad, otherInputs, rows=rows)
        feats = self.self_att(src=audio_feat, key_padding_mask=pad)
        return self._tail(output, feats, audio_feat_len.to(audio_feat.device), pad, otherInputs)

    def extract_hidden_states(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> Tuple:
        pad = get_keypadding_mask(audio_feat.shape[1], audio_len.to(audio_feat.device))
        return tuple(self.self_att.extract_hidden_states(src=audio_feat, key_padding_mask=pad))


class _ClsSlotFn(torch.autograd.Function):
    """Writes the CLS token into row 0 of every utterance of the encoder's output buffer [B, R, D] (in place: that slot is free,
    weighted_sum._WeightedSumSrcFn) and returns the buffer; the CLS gradient is the batch sum of row 0's."""

    @staticmethod
    def forward(ctx, buf, cls):
        buf[:, 0] = cls.detach().reshape(-1)
        ctx.mark_dirty(buf)
        ctx.cls = (cls.shape, cls.dtype)
        return buf

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.cls
        return g, g[:, 0].float().sum(0).reshape(shape).to(dtype)


class KW_HybridBranchPlus(KW_CascadedBranchPlus):
    """kw_branches.py:780-891: one shared self-attention block over [CLS ; frames]; the CLS row is the parallel
    embedding, the frame rows feed the cascaded tail."""

    def __init__(self, config, audio_dim: int, text_dim: int, out_dim: int, clip) -> None:
        super().__init__(config, audio_dim, text_dim, clip)
        self.out_dim = out_dim
        cb = _get(_get(config, "model_settings"), "cascaded_branch")
        self.cls = self._create_cls(length=1, cls_dim=_get(_get(cb, "transformer_args"), "d_model"))
        self._create_self_attn_layer(cb)
        self.parallel_proj = nn.Linear(self.audio_dim, self.out_dim)

    def forward(self, audio_feat: torch.Tensor, audio_feat_len: torch.Tensor, otherInputs: dict = {}) -> dict:
        output = defaultdict(lambda: None)
        bsz, T = audio_feat.shape[:2]
        lens = audio_feat_len.to(audio_feat.device)
        pad = get_keypadding_mask(T + 1, lens, add=1)
        from .mha_block import resident_rows
        res = resident_rows(audio_feat)
        if res is not None and res[1] == 1:
            # the encoder's output rows keep a free slot in front of every utterance (weighted_sum.forward_padded): the CLS token
            # goes there and the attention block reads [CLS ; frames] in place - no concatenated copy of the features
            buf = _ClsSlotFn.apply(res[0], self.cls)
            src = buf[:, : T + 1]
            src._sc_handle = SimpleNamespace(src=buf, inplace_ok=True)
        else:
            src = torch.cat([self.cls.expand(bsz, -1, -1).to(audio_feat.dtype), audio_feat], dim=1)
        if self._rows_path(audio_feat):
            rows, cls_rows = self.self_att.forward_rows(src=src, key_padding_mask=pad, n_cls=1)
            output["parallel_audio_feat"] = linear_f32_autograd(cls_rows, self.parallel_proj.weight, self.parallel_proj.bias)
            frames_pad = pad[:, 1:]
            if getattr(pad, "_sc_lens", None) is not None:
                frames_pad._sc_lens = (lens, 0)           # the frames' mask is the lengths' own mask
            return self._tail(output, None, lens, frames_pad, otherInputs, rows=rows)
        post = self.self_att(src=src, key_padding_mask=pad)
        output["parallel_audio_feat"] = linear_f32_autograd(post[:, :1].reshape(-1, self.audio_dim).float(), self.parallel_proj.weight,
                                                            self.parallel_proj.bias)
        return self._tail(output, post[:, 1:].reshape(-1, T, self.audio_dim), lens, pad[:, 1:], otherInputs)

    def extract_hidden_states(self, audio_feat: torch.Tensor, audio_len: torch.Tensor) -> Tuple:
        bsz, T = audio_feat.shape[:2]
        pad = get_keypadding_mask(T + 1, audio_len.to(audio_feat.device) + 1)
        src = torch.cat([self.cls.expand(bsz, -1, -1).to(audio_feat.dtype), audio_feat], dim=1)
        return tuple(x[:, 1:, ...] for x in self.self_att.extract_hidden_states(src=src, key_padding_mask=pad))
