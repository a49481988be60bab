"""speechclip_plus_amd: MI355X-native contrastive hot path of SpeechCLIP+ (HuBERT encoder -> CLS attention
pooling head -> speech<->image InfoNCE), behind the reference's module API.  See DESIGN.md."""
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
