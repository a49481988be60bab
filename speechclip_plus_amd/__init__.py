"""speechclip_plus_amd: MI355X-native contrastive hot path of SpeechCLIP+ (HuBERT encoder -> CLS attention
pooling head -> speech<->image InfoNCE), behind the reference's module API.  See DESIGN.md."""
import os as _os

# The step runs on several HIP streams side by side (caller's, "encoder", "optimiser", "head_aux", "h2d", "allreduce" + RCCL's own).
# The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run one after the
# other: measured on MI355X / ROCm 7.2 with a (one-rank) RCCL world the encoder-under-the-previous-tail overlap disappears at the
# default (13.34 vs 13.13 ms one-stream) and is back at 8 queues (12.60 ms; without RCCL 12.55 vs 12.61: no cost).  The variable is read
# when the HIP runtime initialises (first device call, not `import torch`), so it is set here unless the user chose a value.
import torch as _torch

_HIP_UP_AT_IMPORT = _torch.cuda.is_initialized()          # a process that touched the GPU before importing the package
_HWQ_AT_IMPORT = _os.environ.get("GPU_MAX_HW_QUEUES")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_HWQ_WANTED, _HWQ_RUNTIME_DEFAULT = 8, 4


def hw_queue_status() -> dict:
    """What the HIP runtime actually multiplexes the package's streams onto.  ``effective`` = the value the runtime read when it
    initialised: the environment's at that moment, else its default (4).  ``ok`` False = the overlapped schedule shares hardware
    queues (measured with a one-rank RCCL world: 13.3 ms per step instead of 12.6, profiles/r05_collectives_hw_queues.json)."""
    def _int(v, default):
        try:
            return int(v)
        except (TypeError, ValueError):
            return default
    if _HIP_UP_AT_IMPORT:
        eff, why = _int(_HWQ_AT_IMPORT, _HWQ_RUNTIME_DEFAULT), "HIP was initialised before speechclip_plus_amd was imported"
    else:
        eff, why = _int(_os.environ.get("GPU_MAX_HW_QUEUES"), _HWQ_RUNTIME_DEFAULT), "set at import" if _HWQ_AT_IMPORT is None else "the user's value"
    return {"effective": eff, "wanted": _HWQ_WANTED, "ok": eff >= _HWQ_WANTED, "hip_initialised_before_import": _HIP_UP_AT_IMPORT, "source": why}


_hwq_warned = False


def warn_if_hw_queues_short() -> None:
    """Called where the package starts running streams side by side (ops.shared_stream): ONE warning per process when the hardware
    queue count cannot be what the overlapped schedule needs."""
    global _hwq_warned
    st = hw_queue_status()
    if st["ok"] or _hwq_warned:
        return
    _hwq_warned = True
    import warnings
    warnings.warn(f"speechclip_plus_amd: GPU_MAX_HW_QUEUES is effectively {st['effective']} ({st['source']}); the encoder / tail / collective "
                  f"streams then share hardware queues and the overlapped step loses its overlap under RCCL (measured 13.3 ms instead of 12.6 "
                  f"ms per step at B = 64 x 10 s).  Export GPU_MAX_HW_QUEUES={_HWQ_WANTED} before the process touches the GPU, or import "
                  f"speechclip_plus_amd before the first device call.", RuntimeWarning, stacklevel=3)


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
           "WeightedSumLayer", "MaskedContrastiveLoss", "mutualRetrieval", "set_dropout", "hw_queue_status"]
