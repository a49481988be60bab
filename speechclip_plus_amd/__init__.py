"""speechclip_plus_amd: MI355X-native contrastive hot path of SpeechCLIP+ (HuBERT encoder -> CLS attention
pooling head -> speech<->image InfoNCE), behind the reference's module API.  See DESIGN.md."""
import os as _os

# The step runs on several HIP streams side by side (caller's, "encoder", "optimiser", "head_aux", "h2d", "allreduce" + RCCL's own).
# The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run one after the
# other: measured on MI355X / ROCm 7.2 with a (one-rank) RCCL world the encoder-under-the-previous-tail overlap disappears at the
# default (13.34 vs 13.13 ms one-stream) and is back at 8 queues (12.60 ms; without RCCL 12.55 vs 12.61: no cost).  The variable is read
# when the HIP runtime initialises (first device call, not `import torch`), so it is set here unless the user chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import load_config
from .model import (Config, KWClip_GeneralTransformer, base_parallel_config, cascaded_plus_base_config,
                    hybrid_plus_large_config, large_parallel_config, set_dropout)
from .speech_encoder import FairseqSpeechEncoder_Hubert, HubertArch, random_hubert_state_dict
from .kw_branches import KW_CascadedBranchPlus, KW_HybridBranchPlus, KW_ParallelBranch
from .transformer_models import MultiheadAttentionAndNorm, TransformerEncoder
from .weighted_sum import WeightedSumLayer
from .losses import MaskedContrastiveLoss
from .retrieval import mutualRetrieval

__all__ = ["Config", "load_config", "KWClip_GeneralTransformer", "base_parallel_config", "large_parallel_config", "cascaded_plus_base_config", "hybrid_plus_large_config", "KW_CascadedBranchPlus",
           "KW_HybridBranchPlus", "FairseqSpeechEncoder_Hubert", "HubertArch",
           "random_hubert_state_dict", "KW_ParallelBranch", "TransformerEncoder", "MultiheadAttentionAndNorm",
           "WeightedSumLayer", "MaskedContrastiveLoss", "mutualRetrieval", "set_dropout"]
