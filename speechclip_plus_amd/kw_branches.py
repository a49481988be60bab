"""Branch heads: mirror of avssl/model/kw_branches.py for the hot path.

``KW_ParallelBranch`` (kw_branches.py:200-282): 1 CLS token + 1-layer TransformerEncoder + Linear D -> E.
The committed reference constructor assigns ``None`` over ``self_att``; the intended computation
(:266-280) is what is built.  Same forward signature as the caller uses (``audio_feat``, ``audio_feat_len``
keyword at kwClip.py:881-884; ``audio_len`` accepted as the positional name the reference declares).
"""
import logging
from collections import defaultdict
from typing import Optional, Tuple

import torch
from torch import nn

from . import transformer_models as TransformerModels

logger = logging.getLogger(__name__)


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
        raise NotImplementedError("parallel-branch hidden states need the full-sequence layer (scope row f3)")

    def forward(self, audio_feat: torch.Tensor, audio_len: Optional[torch.Tensor] = None, otherInputs: dict = None,
                audio_feat_len: Optional[torch.Tensor] = None) -> dict:
        if audio_len is None:
            audio_len = audio_feat_len
        output = defaultdict(lambda: None)
        # kw_branches.py:266-280: CLS + frames, key_padding_mask from audio_len + 1, row 0, projection
        out = self.self_att.cls_forward(self.cls, audio_feat, audio_len + 1)
        if hasattr(self, "linear_proj"):
            out = self.linear_proj(out)
        output["parallel_audio_feat"] = out
        return output
