"""CPU-side checks that need no GPU: the C-ABI library loads and exports every declared symbol, host-side
length / layout logic agrees with the oracle, metric + schedule helpers, loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from speechclip_plus_amd import _lib
    header = open(os.path.join(ROOT, "include", "speechclip_hip.h")).read()
    declared = set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", header))
    declared -= {"sc_gemm_args"}
    assert {"sc_gemm_bf16", "sc_attn_fwd_bf16", "sc_infonce_lse", "sc_cls_pool_fwd"} <= declared
    lib = _lib.lib()                       # raises if the .so is missing (no fallback)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/speechclip_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"sc_last_error"} == declared
    assert lib.sc_abi_version() == 1
    # the ctypes mirror of sc_gemm_args must have the C struct's size (8-byte fields, natural alignment)
    assert ctypes.sizeof(_lib.GemmArgs) == 6 * 8 + 4 * 4 + 3 * 8 + 2 * 4 + 8 + 6 * 4 + 10 * 8 + 8


def test_ops_fail_loudly_on_cpu_tensors():
    from speechclip_plus_amd import ops
    x = torch.zeros(128, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear_bf16(x, x)


def test_plan_geometry_matches_oracle_lengths():
    from speechclip_plus_amd.speech_encoder import HubertArch, conv_out_lengths
    arch = HubertArch()
    for L in [400, 799, 16000, 102400, 160000, 163840, 40959 * 4]:
        t = conv_out_lengths(L, arch)
        assert t == oracle.conv_out_lengths(L)
        T = t[-1]
        R = (T + 2 + 127) // 128 * 128
        R_l = [R * 2 ** (6 - i) for i in range(7)]
        assert all(r >= tt for r, tt in zip(R_l, t)), (L, R_l, t)      # every valid conv row is stored
    assert conv_out_lengths(160000, arch)[-1] == 499 and conv_out_lengths(102400, arch)[-1] == 319


def test_retrieval_matches_golden(golden):
    from speechclip_plus_amd import mutualRetrieval
    fx = golden("retrieval.npz")
    s = torch.from_numpy(fx["score"])
    AB, BA, mean = mutualRetrieval(s, s.T, torch.from_numpy(fx["a_ids"]), torch.from_numpy(fx["b_ids"]), [1, 5, 10])
    for i, k in enumerate([1, 5, 10]):
        assert abs(AB[f"recall@{k}"] - fx["AB"][i]) < 1e-4 and abs(BA[f"recall@{k}"] - fx["BA"][i]) < 1e-4
        assert abs(mean[f"recall@{k}"] - fx["mean"][i]) < 1e-4


def test_lr_schedules():
    """against the closed forms of avssl/optim/scheduler.py (LambdaLR multipliers)."""
    from speechclip_plus_amd.optim import linear_warmup_decay, noam
    f = lambda s: linear_warmup_decay(s, 5000, 50000, 1e-4, 1e-8)
    assert abs(f(0) - 1 / 5000) < 1e-15 and abs(f(2499) - 0.5) < 1e-12 and abs(f(4999) - 1.0) < 1e-12
    assert abs(f(49999) - 1e-4) < 1e-9 and f(27500) < f(5001) < 1.0
    assert abs(noam(0, 4000) - 1 / 4000) < 1e-15 and abs(noam(15999, 4000) - 0.5) < 1e-12


def test_config_and_model_surface():
    from speechclip_plus_amd import base_parallel_config, KWClip_GeneralTransformer
    cfg = base_parallel_config()
    assert cfg.model_settings.parallel_branch.transformer_args.nhead == 8
    assert cfg.audio_encoder.feat_select_idx == "weighted_sum"
    for m in ["forward", "compute_loss", "encode_speech", "feature_extractor_s3prl", "processWavs",
              "getTrainableParams", "training_step", "training_step_end", "forward_audio", "forward_image"]:
        assert callable(getattr(KWClip_GeneralTransformer, m))
