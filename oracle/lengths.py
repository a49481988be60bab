"""Length / mask rules of the hot path (oracle; test infrastructure only).

Three different frame-length rules coexist in the reference and must be reproduced exactly:

1. conv output length  T = f(L)            fairseq ConvFeatureExtractionModel (7 convs, no padding)
2. fairseq frame mask  "all samples padded" fairseq HubertModel.forward_padding_mask, called at
                                            avssl/module/speech_encoder_plus.py:81-82
3. head mask length    round(len / 320)     avssl/module/speech_encoder_plus.py:604-611 (python round,
                                            half-to-even), mask built by avssl/util/data_utils.py:6-22
"""
from typing import List, Sequence

import torch

CONV_KERNELS = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)


def conv_out_lengths(L: int, kernels: Sequence[int] = CONV_KERNELS,
                     strides: Sequence[int] = CONV_STRIDES) -> List[int]:
    """Per-layer output lengths of the un-padded strided conv stack: floor((T-k)/s)+1."""
    out = []
    t = int(L)
    for k, s in zip(kernels, strides):
        t = (t - k) // s + 1
        out.append(t)
    return out


def fairseq_valid_frames(wav_len: Sequence[int], L: int, T: int) -> List[int]:
    """Number of un-padded frames per utterance under fairseq's forward_padding_mask.

    fairseq: extra = L % T; mask = mask[:, :L-extra].view(B, T, L//T).all(-1)
    A frame t is padded iff every sample in [t*c, (t+1)*c) is padding, c = L // T,
    i.e. iff t*c >= len.  Valid frames = #{t < T : t*c < len} = min(T, ceil(len / c)).
    """
    c = L // T
    return [min(T, -(-int(l) // c)) for l in wav_len]


def feat_len_rule(wav_len: Sequence[int], T: int, downsample_rate: int = 320) -> List[int]:
    """speech_encoder_plus.py:604-611: min(round(l / 320), T) with python's banker's rounding."""
    return [min(round(int(l) / downsample_rate), T) for l in wav_len]


def get_keypadding_mask(max_length: int, data_lens: torch.Tensor) -> torch.Tensor:
    """avssl/util/data_utils.py:6-22: bool (B, max_length), True = padding."""
    ar = torch.arange(max_length).unsqueeze(0)
    return ar >= data_lens.reshape(-1, 1).to(ar.device)
