"""Batch format of the reference's loaders (scope row f4): ``collate_general`` builds the dict that
``KWClip_GeneralTransformer.forward`` consumes - mirror of avssl/data/collate_function.py:7-36.

Semantics kept: a ``wav_len`` key is derived from the un-padded waveforms, ``wav`` is zero-padded to the longest
utterance of the batch (batch first), other tensors are stacked, non-tensor fields (ids, lengths) become int64 tensors.
"""
from typing import Dict, List, Sequence

import torch


def collate_general(batch: Sequence[dict]) -> Dict[str, torch.Tensor]:
    if len(batch) == 0:
        raise ValueError("empty batch")
    keys: List[str] = list(batch[0].keys())
    derive_len = "wav" in keys and isinstance(batch[0]["wav"], torch.Tensor)
    out: Dict[str, torch.Tensor] = {}
    for k in keys:
        vals = [row[k] for row in batch]
        if isinstance(vals[0], torch.Tensor):
            if k == "wav":
                L = max(int(v.shape[0]) for v in vals)
                padded = vals[0].new_zeros((len(vals), L) + tuple(vals[0].shape[1:]))
                for i, v in enumerate(vals):
                    padded[i, : v.shape[0]] = v
                out[k] = padded
            else:
                out[k] = torch.stack(vals, dim=0)
        else:
            out[k] = torch.tensor(vals, dtype=torch.long)
    if derive_len:
        out["wav_len"] = torch.tensor([int(row["wav"].shape[0]) for row in batch], dtype=torch.long)
    return out
