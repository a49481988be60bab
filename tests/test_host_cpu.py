"""CPU-side checks that need no GPU: the C-ABI library loads and exports every declared symbol, host-side
length / layout logic agrees with the oracle, metric + schedule helpers, loud failure without a device."""
import ctypes
import sys
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
    declared -= {"sc_gemm_args", "sc_hubert_layer_args", "sc_rt_gemm_args", "sc_rt_ln_args", "sc_rt_ln_bwd_args"}
    assert {"sc_gemm_bf16", "sc_attn_fwd_bf16", "sc_infonce_fwd", "sc_cls_pool_fwd"} <= declared
    lib = _lib.lib()                       # raises if the .so is missing (no fallback)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/speechclip_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"sc_last_error", "sc_hash32", "sc_infonce_workspace_floats", "sc_workspace_bytes", "sc_sizeof"} == declared
    assert lib.sc_abi_version() == 4
    # the ctypes mirrors of the argument structs have the C structs' sizes (checked again at every load: _lib.lib())
    for what, cls in enumerate((_lib.GemmArgs, _lib.HubertLayerArgs, _lib.RtGemmArgs, _lib.RtLnArgs, _lib.RtLnBwdArgs)):
        assert lib.sc_sizeof(what) == ctypes.sizeof(cls), cls.__name__
    # host twin of the kernels' dropout hash (lowbias32): known answers
    assert lib.sc_hash32(0) == 0 and lib.sc_hash32(1) == 0x688990C0
    ref = lambda x: ((((x ^ (x >> 16)) * 0x7feb352d & 0xffffffff) ^ ((((x ^ (x >> 16)) * 0x7feb352d & 0xffffffff)) >> 15)) * 0x846ca68b) & 0xffffffff
    for x in (1, 12345, 0xdeadbeef, 0xffffffff):
        t = ref(x)
        assert lib.sc_hash32(x) == (t ^ (t >> 16))


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


def test_retrieval_degenerate_scores_are_misses():
    """NaN or collapsed embeddings must not read as recall 100 (the counting formulation compared against a NaN
    / tied best score and found nobody ahead).  Non-finite rows are misses, exact ties count against the query."""
    from speechclip_plus_amd import mutualRetrieval
    nA, nB = 12, 4
    a_ids, b_ids = torch.arange(nA) // 3, torch.arange(nB)
    for S in (torch.full((nA, nB), float("nan")), torch.zeros(nA, nB), torch.full((nA, nB), float("inf"))):
        AB, BA, mean = mutualRetrieval(S, S.t().contiguous(), a_ids, b_ids, [1, 2])
        assert AB["recall@1"] == 0.0 and BA["recall@1"] == 0.0 and mean["recall@2"] == 0.0, (S[0, 0], AB, BA)
    # one NaN row does not poison the others; a healthy matrix still scores 100
    S = torch.eye(nB)[a_ids] + 0.01 * torch.arange(nA * nB).view(nA, nB) / (nA * nB)
    AB, BA, _ = mutualRetrieval(S, S.t().contiguous(), a_ids, b_ids, [1])
    assert AB["recall@1"] == 100.0 and BA["recall@1"] == 100.0
    S[5] = float("nan")
    AB, BA, _ = mutualRetrieval(S, S.t().contiguous(), a_ids, b_ids, [1])
    # audio -> image: the NaN row misses; image -> audio: the NaN candidate (a caption of image 1) sorts ahead of every other image's captions
    assert abs(AB["recall@1"] - 100.0 * 11 / 12) < 1e-4 and BA["recall@1"] == 25.0, (AB, BA)


def test_segment_geometry():
    """Round 4, ragged rows (speech_encoder._Plan.bind / ops.RowSegments): an utterance that needs its first n frames gets a pitch of
    roundup(n + 1, 8) rows at the last conv layer and 2^(6-l) times that at layer l.  Walk the conv stack's receptive fields
    backwards: every row / sample those n frames depend on lies INSIDE the utterance's own segment, for every n."""
    from speechclip_plus_amd.speech_encoder import HubertArch, FairseqSpeechEncoder_Hubert
    from speechclip_plus_amd.ops import RowSegments
    arch = HubertArch()
    ks, ss = arch.conv_kernels, arch.conv_strides
    for n in list(range(1, 70)) + [99, 127, 128, 129, 319, 320, 499, 511, 512]:
        pitch = (n + 1 + 7) // 8 * 8
        need = n                                   # rows needed at layer l (walking from the last layer down)
        for l in range(len(ks) - 1, 0, -1):
            assert need <= pitch * 2 ** (len(ks) - 1 - l)
            need = (need - 1) * ss[l] + ks[l]      # rows of layer l - 1 that rows < need of layer l read
        assert need <= pitch * 64                  # conv layer 0 rows
        samples = (need - 1) * ss[0] + ks[0]
        assert samples <= pitch * 320, (n, samples)
    # pitches of a batch: frames the key mask admits, frames the head reads + the CIF conv's tail, never more than T
    enc = FairseqSpeechEncoder_Hubert.__new__(FairseqSpeechEncoder_Hubert)
    enc.tail_rows = 2
    need, pitch = enc.segment_pitches(499, [499, 100, 7, 300], [499, 101, 6, 301], ragged=True)
    assert need == [499, 103, 8, 303] and pitch == [504, 104, 16, 304]
    need, pitch = enc.segment_pitches(319, [319, 100], [319, 99], ragged=False)
    assert need == [319, 319] and pitch == [320, 320]            # the reference's 6.4 s training crop: 320 rows for 319 frames
    # host tables: chunk entries (first row, pitch, utterance, 0) per 8 rows; work list = q-blocks, longest key count first
    tab, row0, nwork = RowSegments.host_tables([16, 8, 160], [13, 2, 140])
    assert row0 == [0, 16, 24, 184] and nwork == 1 + 1 + 2
    t = tab.tolist()
    chunk, work, r0 = t[: 4 * 23], t[92: 92 + 16], t[108:]
    assert chunk[:8] == [0, 16, 0, 0, 0, 16, 0, 0] and chunk[8:12] == [16, 8, 1, 0] and chunk[12:16] == [24, 160, 2, 0] and chunk[-4:] == [24, 160, 2, 0]
    assert r0 == row0 and work == [2, 24, 160, 140, 2 | (1 << 16), 24, 160, 140, 0, 0, 16, 13, 1, 16, 8, 2]
    assert RowSegments.host_tables([16, 8], [9, 9], keys_known=False)[0].tolist()[12:20] == [0, 0, 16, -1, 1, 16, 8, -1]
    assert RowSegments.table_ints(3, 512 * 3) >= len(t)


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


def test_collate_general_matches_reference_format():
    """data.collate_general against the batch dict of avssl/data/collate_function.py:7-36 (semantics restated in the test:
    wav zero-padded batch-first + wav_len from the raw lengths, tensors stacked, scalars -> int64)."""
    from speechclip_plus_amd.data import collate_general
    g = torch.Generator().manual_seed(0)
    rows = [{"wav": torch.randn(n, generator=g), "image": torch.randn(3, 4, 4, generator=g), "id": i * 7}
            for i, n in enumerate([50, 120, 33])]
    out = collate_general(rows)
    assert set(out) == {"wav", "image", "id", "wav_len"}
    assert out["wav"].shape == (3, 120) and out["wav_len"].tolist() == [50, 120, 33] and out["wav_len"].dtype == torch.long
    for i, r in enumerate(rows):
        assert torch.equal(out["wav"][i, : len(r["wav"])], r["wav"]) and float(out["wav"][i, len(r["wav"]):].abs().sum()) == 0
    assert torch.equal(out["image"], torch.stack([r["image"] for r in rows])) and out["id"].tolist() == [0, 7, 14]


def test_split_reference_state_dict_key_map():
    from speechclip_plus_amd.model import KWClip_GeneralTransformer
    sd = {"audio_encoder.encoder.encoder.layers.3.fc1.weight": 1, "audio_encoder.encoder.feature_extractor.conv_layers.0.0.weight": 2,
          "audio_encoder.weightedsum_layer.weights": 3, "parallel_branch.cls": 4, "parallel_branch.self_att.model.layers.0.linear1.bias": 5,
          "criterion.temperature": 6, "clip.model.visual.conv1.weight": 7, "clip.model.token_embedding.weight": 8,
          "clip.model.logit_scale": 9}
    hubert, rest = KWClip_GeneralTransformer.split_reference_state_dict(sd)
    assert hubert == {"encoder.layers.3.fc1.weight": 1, "feature_extractor.conv_layers.0.0.weight": 2}
    assert set(rest) == {"audio_encoder.weightedsum_layer.weights", "parallel_branch.cls",
                         "parallel_branch.self_att.model.layers.0.linear1.bias", "criterion.temperature",
                         "clip.model.token_embedding.weight"}


def test_audio_transform_like_the_reference_test():
    """test/test_audio_transform.py of the reference, on this build's random_crop_max_length (audio_transforms.py:5-23)."""
    from speechclip_plus_amd.speech_encoder import random_crop_max_length
    wav = torch.randn((10000,), dtype=torch.float)
    out_1 = random_crop_max_length(wav, 1000)
    out_2 = random_crop_max_length(wav, 1000, 100)
    assert out_1.shape == (1000,)
    assert out_2.shape == (100,)
    short = torch.randn(300)
    assert torch.equal(random_crop_max_length(short, 1000), short)                    # shorter than the cap: returned as is


def test_crop_windows_draw_the_references_windows(golden):
    """speech_encoder.crop_windows against tests/golden/crop.npz - the windows avssl/data/audio_transforms.py:5-23 cut when driven
    as speech_encoder_plus.py:548-552 drives it (make_golden.py crop): same offsets, same lengths, the same NUMBER of draws from
    numpy's global generator (an utterance at or under the cap draws nothing; one sample over the cap draws randint(1))."""
    from speechclip_plus_amd.speech_encoder import crop_windows
    fx = golden("crop.npz")
    for ci in range(int(fx["n"])):
        B, max_len, seed, after = (int(v) for v in fx[f"c{ci}_meta"])
        np.random.seed(seed)
        off, out = crop_windows(fx[f"c{ci}_lens"].tolist(), max_len)
        assert off == fx[f"c{ci}_off"].tolist() and out == fx[f"c{ci}_out"].tolist(), ci
        assert int(np.random.randint(1 << 30)) == after, ci
        assert all(o + n <= l for o, n, l in zip(off, out, fx[f"c{ci}_lens"].tolist()))


def test_host_lengths_travel_with_the_batch():
    """data.collate_general / attach_host_lengths / transfer_batch_to_device (CPU -> CPU here: the device branch is a -m gpu test):
    the host twin of wav_len is what keeps the ragged layout and the crop free of device read-backs."""
    from speechclip_plus_amd.data import attach_host_lengths, collate_general, transfer_batch_to_device
    rows = [{"wav": torch.randn(n), "id": i} for i, n in enumerate([50, 120, 33])]
    out = collate_general(rows)
    assert out["wav_len"]._sc_host == [50, 120, 33]
    moved = transfer_batch_to_device(out, "cpu")
    assert moved["wav_len"]._sc_host == [50, 120, 33] and moved["wav"] is out["wav"]
    t = attach_host_lengths(torch.tensor([3, 4]), [3, 4])
    assert t._sc_host == [3, 4]


def test_recall_eval_generators_draw_the_oracles_streams():
    """tools/recall_eval.py (used by bench.py without the oracle) must generate exactly the weights the fixture was made with."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import recall_eval
    a = oracle.init_parallel_branch_weights(seed=recall_eval.SEED_HEAD)
    a["cls"] = a["cls"] * 0.1
    b = recall_eval.head_weights()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    ha = oracle.init_hubert_weights(oracle.HubertArch.base(), seed=recall_eval.SEED_W)
    hb = recall_eval.hubert_weights()
    assert set(ha) == set(hb) and all(torch.equal(ha[k], hb[k]) for k in ha)
    w0 = recall_eval.utterance(3, 1)
    assert 20000 <= len(w0) <= 40000 and len(w0) % 320 == 0 and torch.equal(w0, recall_eval.utterance(3, 1))


def test_reference_yaml_recipes_parse_unchanged():
    """Every yaml under the reference's config/ (present in the build container only) loads through load_config with the keys
    the model reads; a recipe-shaped yaml text of our own covers the same code on any box."""
    import glob
    from speechclip_plus_amd import load_config
    text = """
model_settings: {cascaded_objective_weight: 0.0, parallel_objective_weight: 1.0,
                 parallel_branch: {transformer_type: TransformerEncoder, need_projection: true,
                                   transformer_args: {n_layers: 1, d_model: 768, nhead: 8, dim_feedforward: 3072, dropout: 0.1}}}
cl_loss: {type: MaskedContrastiveLoss, args: {temperature: 0.07, temperature_trainable: false}}
clip: {name: ViT-L/14, reduce_subword_embbedding: ./avssl/data/coco_stat/none.npy}
audio_encoder: {type: FairseqHubert, name: hubert_large_ll60k, trainable: false, feat_select_idx: weighted_sum,
                optim: {name: Adam, args: {lr: 1.e-4, weight_decay: 1.e-6}}}
"""
    with pytest.raises(FileNotFoundError):
        load_config(text)
    cfg = load_config(text, allow_synthetic_vocab=True)
    assert cfg.clip.embed_dim == 768 and cfg.clip.reduce_subword_embbedding.numel() == 19787
    assert cfg.retrieval.recall_at == [1, 5, 10] and cfg.audio_encoder.optim.args.lr == 1e-4
    files = glob.glob("/root/reference/config/*/*/*.yaml") + glob.glob("/root/reference/config/*/*/*/*.yaml")
    for f in files:
        c = load_config(f, reference_root="/root/reference")
        assert c.audio_encoder.type == "FairseqHubert" and c.cl_loss.type == "MaskedContrastiveLoss"
        assert c.clip.embed_dim in (512, 768)


def test_builtin_configs_equal_the_reference_yamls():
    """every built-in recipe against load_config(<the reference's yaml>) on EVERY key the built-in holds (= the
    keys the path reads).  A key the yaml does not have must carry the reference's default, listed here with its source.  Runs in
    the build container only (the yamls do not travel)."""
    from speechclip_plus_amd import (base_parallel_config, cascaded_plus_base_config, hybrid_plus_large_config, large_parallel_config,
                                     load_config)
    ref = "/root/reference/config"
    if not os.path.isdir(ref):
        pytest.skip("the reference's yaml recipes exist in the build container only")
    defaults = {"audio_encoder.normalize_hiddenstates": False,                 # speech_encoder_plus.py:350
                "model_settings.parallel_branch.need_projection": True,       # the plus yamls do not configure a separate parallel head:
                "model_settings.parallel_branch.transformer_type": "TransformerEncoder",    # inherited, unused (hybrid's CLS row shares the block)
                "trainer.accumulate_grad_batches": 1}                         # pytorch_lightning.Trainer default

    def flat(c, pre=""):
        out = {}
        for k, v in c.items():
            if hasattr(v, "items"):
                out.update(flat(v, pre + k + "."))
            else:
                out[pre + k] = v
        return out

    for builtin, rel in ((base_parallel_config(), "speechCLIP/model_base/spchclp_p.yaml"),
                         (large_parallel_config(), "speechCLIP/model_large/flickr/spchclp_p.yaml"),
                         (cascaded_plus_base_config(), "speechCLIP+/model_base/spchclip_c+.yaml"),
                         (hybrid_plus_large_config(), "speechCLIP+/model_large/coco/spchclip_h+.yaml")):
        fy = flat(load_config(os.path.join(ref, rel), reference_root="/root/reference", allow_synthetic_vocab=True))
        for k, v in flat(builtin).items():
            if isinstance(v, torch.Tensor):                                    # the reduced vocabulary: same size as the yaml's table
                table = np.load(fy[k]) if isinstance(fy[k], str) else fy[k]
                assert v.numel() == len(table), (rel, k, v.numel(), len(table))
            elif k in fy:
                assert v == fy[k], (rel, k, v, fy[k])
            else:
                assert k in defaults and v == defaults[k], (rel, k, v, "not in the yaml and not a documented default")


def test_bench_host_helpers():
    """bench.py's host-side accounting: physical cores are counted from /proc/cpuinfo, and the algorithmic flops of a ragged batch
    count every utterance at its own length (padding is not achieved work)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pc = bench.physical_cores()
    assert pc is None or (isinstance(pc, int) and 1 <= pc <= (os.cpu_count() or 1))
    full = bench.forward_summary(12.0, 64, 160000, 499)
    assert abs(full["alg_gflop_per_utt"] - 148.14) < 0.05                      # 148.1 GFLOP per 10 s utterance
    ragged = bench.forward_summary(12.0, 64, 160000, 499, lens=[160000] + [80000] * 63)
    assert ragged["alg_gflop_per_utt"] < 0.55 * full["alg_gflop_per_utt"]      # shorter utterances count less than linearly (attention ~ T^2)
    same = bench.forward_summary(12.0, 64, 160000, 499, lens=[160000] * 64)
    assert abs(same["alg_gflop_per_utt"] - full["alg_gflop_per_utt"]) < 1e-6
    # round 6: roofline.traffic from the counter passes of the same invocation - KiB per launch, FETCH x 2 (gfx950), the first launch
    # of every instantiation dropped, every instantiation of the dominant kernel pooled; anything unusable -> None (stored-profile fallback)
    live = {"void gemm256_kernel<0, 256, 1, 0, 0, 0>(sc_gemm_args)": {"fetch_kib": [9e9, 100.0, 300.0], "write_kib": [9e9, 50.0, 70.0]},
            "void gemm256_kernel<0, 192, 0, 1, 1, 0>(sc_gemm_args)": {"fetch_kib": [9e9, 200.0], "write_kib": [9e9, 60.0]},
            "void gemm256_kernel<4, 256, 0, 0, 0, 0>(sc_gemm_args)": {"fetch_kib": [1.0, 1.0], "write_kib": [1.0, 1.0]},
            "attn_fwd_kernel<1, 0>": {"fetch_kib": [5.0, 5.0], "write_kib": [5.0, 5.0]}}
    t = bench.traffic_from_live(live, "gemm256_kernel<0,")
    assert t["traffic"] == round((2.0 * 600.0 + 180.0) * 1024.0 / 3) and t["traffic_fetch_bytes"] == round(2.0 * 600.0 * 1024.0 / 3)
    assert t["traffic_write_bytes"] == round(180.0 * 1024.0 / 3) and t["traffic_source"].startswith("live:")
    assert bench.traffic_from_live("rocprofv3 not found", "gemm256_kernel<0,") is None
    assert bench.traffic_from_live({"gemm256_kernel<0, x>": {"fetch_kib": [1.0], "write_kib": [1.0, 2.0]}}, "gemm256_kernel<0,") is None


@pytest.mark.parametrize("suffix", ["", "_b"])
def test_natural_margin_recall_fixture_is_consistent(suffix):
    """tests/golden/recall_eval_natural{,_b}.npz (make_recall_natural_fixture.py; _b = gallery B of round 5, another noise seed): the
    stored ranks give the recalls of its summary, the gallery is unit-norm and orthogonal to nothing planted (role-free), and the
    report code reproduces the emulation's own flip counts when it is fed the fixture's references."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import recall_eval
    fx = dict(np.load(recall_eval.FIXTURE_NATURAL.replace(".npz", suffix + ".npz")))
    summ = json.load(open(os.path.join(ROOT, "tests", "golden", f"recall_eval_natural{suffix}_margins.json")))
    r32, rem = torch.from_numpy(fx["rank_ai_fp32"]).long(), torch.from_numpy(fx["rank_ai_bf16emu"]).long()
    assert recall_eval.recalls(r32) == summ["fp32"]["audio_to_image"] and recall_eval.recalls(rem) == summ["bf16emu"]["audio_to_image"]
    assert 45.0 < summ["fp32"]["audio_to_image"][0] < 55.0                       # natural margins: recall@1 tuned to ~50 %
    assert [int(((r32 < k) != (rem < k)).sum()) for k in (1, 5, 10)] == summ["bf16emu"]["rank_flips_vs_fp32_audio_to_image"]
    img = torch.from_numpy(fx["image"])
    assert torch.allclose(img.norm(dim=-1), torch.ones(1000), atol=1e-5)
    assert max(summ["largest_fp32_margin_of_an_emulation_flip_in_sigma"]) < 4.0 and summ["fraction_of_queries_within_3_sigma_at_1_5_10"][0] > 0.05


def test_flat_optimiser_layout_is_16_byte_aligned():
    """optim.FlatAdam (host logic only, no kernel): every parameter starts on a 4-float boundary of the flat buffers whatever the
    sizes in front of it, parameters and gradients alias the buffers, `span` covers consecutive parameters including the padding
    between them, `n` counts parameters and `size` buffer elements."""
    import torch
    from speechclip_plus_amd.optim import ALIGN, FlatAdam
    shapes = [(13,), (8, 6), (1,), (3, 5, 7), (4,), (2,)]
    params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    frozen = torch.nn.Parameter(torch.randn(5), requires_grad=False)
    vals = [p.detach().clone() for p in params]
    opt = FlatAdam(params[:3] + [frozen] + params[3:], lr=1e-3)
    assert ALIGN == 4 and len(opt.params) == len(params) and all(o % ALIGN == 0 for o in opt.offsets)
    assert opt.n == sum(v.numel() for v in vals) and opt.size == sum((v.numel() + 3) // 4 * 4 for v in vals)
    for p, v, off in zip(params, vals, opt.offsets):
        assert torch.equal(p.detach(), v) and p.data_ptr() == opt.flat_p.data_ptr() + 4 * off
        assert p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off and p.grad.shape == p.shape
    lo, hi = opt.span(params[1:4])
    assert lo == opt.offsets[1] and hi == opt.offsets[3] + params[3].numel()
    used = torch.zeros(opt.size, dtype=torch.bool)
    for p, off in zip(params, opt.offsets):
        assert not used[off: off + p.numel()].any()
        used[off: off + p.numel()] = True
    assert float(opt.flat_p[~used].abs().sum()) == 0.0
    params[2].grad = None
    opt.zero_grad()
    assert params[2].grad.data_ptr() == opt.flat_g.data_ptr() + 4 * opt.offsets[2]

    assert all(getattr(p.grad, "_sc_flat", False) for p in params)                    # ops.grad_target adds only into these


def test_flat_optimiser_checkpoint_carries_its_layout():
    """the optimiser state records where every parameter's moments sit (offsets / numels) and load_state_dict re-packs a
    checkpoint written in another layout - the back-to-back layout of rounds 1-3 ('numel' = the parameters' total, no offsets) and
    any future one - instead of refusing it; a state for other parameter sizes is still refused."""
    from speechclip_plus_amd.optim import FlatAdamOptimizer
    shapes = [(13,), (8, 6), (1,), (3, 5)]
    mk = lambda: [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    a = FlatAdamOptimizer(mk(), lr=1e-3)
    g = torch.Generator().manual_seed(0)
    a.flat.m.copy_(torch.randn(a.flat.size, generator=g))
    a.flat.v.copy_(torch.rand(a.flat.size, generator=g))
    a.flat.step_count = 7
    sd = a.state_dict()
    assert sd["flat_adam"]["offsets"] == [int(o) for o in a.flat.offsets] and sd["flat_adam"]["numels"] == [13, 48, 1, 15]
    b = FlatAdamOptimizer(mk(), lr=1e-3)
    b.load_state_dict(sd)
    assert torch.equal(b.flat.m, a.flat.m) and torch.equal(b.flat.v, a.flat.v) and b.flat.step_count == 7
    # the old layout: moments back to back, no offsets in the entry
    dense = lambda buf: torch.cat([buf[o: o + n] for o, n in zip(a.flat.offsets, sd["flat_adam"]["numels"])])
    old = dict(sd)
    old["flat_adam"] = {"m": dense(a.flat.m), "v": dense(a.flat.v), "step_count": 7, "numel": 77}
    c = FlatAdamOptimizer(mk(), lr=1e-3)
    c.load_state_dict(old)
    for o, n in zip(c.flat.offsets, [13, 48, 1, 15]):
        assert torch.equal(c.flat.m[o: o + n], a.flat.m[o: o + n]) and torch.equal(c.flat.v[o: o + n], a.flat.v[o: o + n])
    other = FlatAdamOptimizer([torch.nn.Parameter(torch.zeros(5))], lr=1e-3)
    with pytest.raises(ValueError):
        other.load_state_dict(sd)


def test_hw_queue_status_and_warning(monkeypatch):
    """The overlapped schedule needs GPU_MAX_HW_QUEUES = 8 when the HIP runtime initialises: the package sets it at
    import, reports what is effective and warns ONCE where the first side stream is created if the process had the GPU up before."""
    import warnings
    import speechclip_plus_amd as pkg
    st = pkg.hw_queue_status()
    assert st["wanted"] == 8 and st["hip_initialised_before_import"] is False and st["ok"] == (st["effective"] >= 8)
    monkeypatch.setattr(pkg, "_HIP_UP_AT_IMPORT", True)
    monkeypatch.setattr(pkg, "_HWQ_AT_IMPORT", None)
    monkeypatch.setattr(pkg, "_hwq_warned", False)
    st = pkg.hw_queue_status()
    assert st == {"effective": 4, "wanted": 8, "ok": False, "hip_initialised_before_import": True,
                  "source": "HIP was initialised before speechclip_plus_amd was imported"}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pkg.warn_if_hw_queues_short()
        pkg.warn_if_hw_queues_short()
    assert len(w) == 1 and "GPU_MAX_HW_QUEUES" in str(w[0].message) and "13.3 ms" in str(w[0].message)
    monkeypatch.setattr(pkg, "_HWQ_AT_IMPORT", "8")            # the user exported it before touching the GPU: fine
    assert pkg.hw_queue_status()["ok"]


def test_collate_general_does_not_pin_inside_a_dataloader_worker(monkeypatch):
    """collate_fn runs in forked workers when num_workers > 0; pinning there would initialise CUDA per worker."""
    import torch.utils.data
    from speechclip_plus_amd import data
    rows = [{"wav": torch.randn(100 + 10 * i), "id": i} for i in range(3)]
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: object())
    out = data.collate_general(rows, pin_memory=True)            # would try to pin (and fail without a GPU) outside the guard
    assert not out["wav"].is_pinned() and out["wav_len"]._sc_host == [100, 110, 120]
