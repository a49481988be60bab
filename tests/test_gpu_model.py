"""GPU parity of the assembled path against the CPU oracle on identical seeded weights and inputs:
HuBERT-base encoder hidden states, weighted sum, parallel head, loss, gradients, recall@k.

Stated tolerances (bf16 storage + fp32 accumulate vs the fp32 oracle):
  hidden states rel-L2 <= 2e-2 per layer; pooled unit-norm embeddings cosine >= 0.999;
  loss |diff| <= 5e-3 at T = 0.07; recall@k identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.fixture(scope="module")
def setup():
    import oracle
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict, HubertArch
    assert torch.cuda.is_available()
    arch = HubertArch()
    sd = random_hubert_state_dict(arch, seed=7122)
    torch.manual_seed(7122)
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd).eval()
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(torch.linspace(-1, 1, 13))
    o_arch = oracle.HubertArch.base()
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    return model, sd, o_arch, head_W, oracle


def test_encoder_hidden_states(setup):
    model, sd, o_arch, head_W, oracle = setup
    g = torch.Generator().manual_seed(11)
    lens = [16000, 9000, 3300]
    wavs = [torch.randn(l, generator=g) for l in lens]
    with torch.no_grad():
        feat, feat_len, hs = model.audio_encoder([w.cuda() for w in wavs], return_hidden_states=True)
        hs_o, feat_len_o = oracle.speech_encoder_forward(sd, o_arch, wavs)
    assert feat_len.cpu().tolist() == feat_len_o.tolist()
    T = hs_o[0].shape[1]
    assert hs[0].shape == (3, T, 768) and len(hs) == 13
    valid = oracle.fairseq_valid_frames(lens, max(lens), T)
    worst = 0.0
    for n in range(13):
        # every frame < T exists in the reference (padded frames carry deterministic values the head may read)
        e_all = rel_l2(hs[n], hs_o[n])
        e_valid = max(rel_l2(hs[n][b, :v], hs_o[n][b, :v]) for b, v in enumerate(valid))
        worst = max(worst, e_all, e_valid)
        assert e_all < 2e-2 and e_valid < 2e-2, (n, e_all, e_valid)
    w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    ws_o = oracle.weighted_sum(w, hs_o)
    assert rel_l2(feat, ws_o) < 1.5e-2
    print("worst hidden-state rel-L2", worst)


def test_layernorm_free_layers_match_the_layernorm_kernels(setup):
    """Round 3, opt-in (SC_FUSED_LN=1; measured no faster, see speech_encoder.py): the frozen post-LN encoder without LayerNorm
    launches (LayerNorms folded into the GEMMs, hidden states kept raw + row statistics, speech_encoder._layers_fused).  Against the
    default form (explicit LayerNorm kernels) on the same weights and inputs: every hidden state, the weighted sum and - through the head - the weighted-sum gradient; eval and
    train mode (the dropout masks are the same stateless hashes in both forms, so the train-mode outputs are comparable too)."""
    from speechclip_plus_amd import speech_encoder as se
    model, sd, o_arch, head_W, oracle = setup
    enc = model.audio_encoder
    g = torch.Generator().manual_seed(12)
    lens = [14000, 9000, 16000, 3300, 12001]
    wavs = [torch.randn(l, generator=g).cuda() for l in lens]
    default = se._FUSED_LN
    res = {}
    try:
        for fused in (True, False):
            se._FUSED_LN = fused
            for train in (False, True):
                enc.train(train)
                enc._drop_calls = 0                      # same per-site dropout seeds in both forms
                with torch.no_grad():
                    feat, feat_len, hs = enc(wavs, return_hidden_states=True)
                pl = enc._plan(len(lens), max(lens))
                assert (pl.lazy is not None) == fused
                res[(fused, train)] = (feat.float().clone(), [h.float().clone() for h in hs])
    finally:
        se._FUSED_LN = default
        enc.eval()
    # (train mode: since round 4 the two forms lay their rows out differently - the LayerNorm-free form keeps the uniform pitch of
    # rounds 1-3, the default form the segment layout - so their hash masks hit different elements: eval is compared element-wise,
    # train mode only for being finite and different from eval)
    for train in (False, True):
        f1, h1 = res[(True, train)]
        f0, h0 = res[(False, train)]
        for n in range(13):
            if not train:
                assert rel_l2(h1[n], h0[n]) < 1.2e-2, (train, n, rel_l2(h1[n], h0[n]))
            assert torch.isfinite(h1[n]).all() and torch.isfinite(h0[n]).all()
        if not train:
            assert rel_l2(f1, f0) < 1.2e-2, (train, rel_l2(f1, f0))
        else:
            assert rel_l2(f1, res[(True, False)][0]) > 0.02
    # both against the oracle (eval): the LayerNorm-free form is not further away than the LayerNorm kernels were
    hs_o, _ = oracle.speech_encoder_forward(sd, o_arch, [w.cpu() for w in wavs])
    e1 = max(rel_l2(res[(True, False)][1][n], hs_o[n]) for n in range(13))
    e0 = max(rel_l2(res[(False, False)][1][n], hs_o[n]) for n in range(13))
    print("worst hidden-state rel-L2 vs oracle: LayerNorm-free", e1, " LayerNorm kernels", e0)
    assert e1 < 2e-2 and e1 < 1.15 * e0 + 1e-3


def test_encoder_large_arch():
    """HuBERT-large wiring (layer_norm extractor with conv bias, utterance-normalised waveform, pre-LN layers,
    D = 1024 / 16 heads / F = 4096) at reduced depth (3 layers) against the oracle."""
    import dataclasses
    import oracle
    from speechclip_plus_amd import FairseqSpeechEncoder_Hubert, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert_large_ll60k"], layers=3)
    sd = random_hubert_state_dict(arch, seed=99)
    enc = FairseqSpeechEncoder_Hubert(name="hubert_large_ll60k", device="cuda:0", feat_select_idx="all", state_dict=sd,
                                      arch=arch).eval()
    o_arch = oracle.HubertArch.large()
    o_arch.layers = 3
    g = torch.Generator().manual_seed(3)
    lens = [12000, 7000]
    wavs = [torch.randn(l, generator=g) * 0.3 + 0.05 for l in lens]
    with torch.no_grad():
        feat, feat_len = enc([w.cuda() for w in wavs])
        hs_o, fl_o = oracle.speech_encoder_forward(sd, o_arch, wavs)
    assert feat_len.cpu().tolist() == fl_o.tolist()
    assert len(feat["hidden_states"]) == 4 and feat["hidden_states"][0].shape[-1] == 1024
    for n in range(4):
        e = rel_l2(feat["hidden_states"][n], hs_o[n])
        assert e < 2e-2, (n, e)


def _large_parallel_model():
    import dataclasses
    import oracle
    from speechclip_plus_amd import KWClip_GeneralTransformer, large_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert_large_ll60k"], layers=2)
    sd = random_hubert_state_dict(arch, seed=5)
    torch.manual_seed(5)
    cfg = large_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).eval()
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(torch.tensor([0.3, -0.2, 0.5]))
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    o_arch = oracle.HubertArch.large()
    o_arch.layers = 2
    return model, sd, o_arch, head_W


def _large_parallel_data(data_seed):
    g = torch.Generator().manual_seed(data_seed)
    lens = [9000, 6000, 9000, 4100]
    wavs = [torch.randn(l, generator=g) * 0.5 for l in lens]
    img = torch.randn(4, 768, generator=g)
    ids = torch.tensor([0, 1, 1, 2])
    wav = torch.zeros(4, max(lens))
    for b, x in enumerate(wavs):
        wav[b, : len(x)] = x
    return wavs, img, ids, {"wav": wav.cuda(), "wav_len": torch.tensor(lens), "image": img.cuda(), "id": ids.cuda()}


def _product_step(model, batch):
    """one forward + backward of the product model -> (loss, unit-norm audio embeddings, head gradients by name, weighted-sum logit gradient)"""
    model.zero_grad(set_to_none=True)
    losses_, _, others = model(batch)
    out = model.compute_loss(losses_)
    out["loss"].backward()
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.parallel_branch.named_parameters() if p.grad is not None}
    return (out["loss"].item(), others["parallel_audio_feat"].detach().float().cpu(), grads,
            model.audio_encoder.weightedsum_layer.weights.grad.detach().float().cpu().clone())


def _wsum_floor(step):
    """The size the weighted-sum logit gradient g_n = w_n <G, c_n>, c_n = LN(h_n) - sum_m w_m LN(h_m), WOULD have if the signs of its
    summands were random: sqrt(sum_n w_n^2 ||G (.) c_n||^2).  Independent rounding noise of relative size r in G or c moves g by about
    r times this, however small |g| itself comes out after the cancellation - the scale errors of g are measured on."""
    return step["floor"]


def _oracle_step(oracle, sd, o_arch, head_W, ws_w, wavs, img, ids, normalize=False, control=False):
    """The reference arithmetic on the CPU.  ``control`` = the bf16-STORAGE-EMULATED oracle (oracle/hubert_ref.py header): the same fp32
    arithmetic with the GEMM weights rounded to bf16 and every tensor the HIP path keeps in bf16 rounded where the kernels store it
    (store hook), plus the weighted sum's bf16 output.  Against it the HIP path differs by summation order, activation form and the
    resulting 1-ulp re-rounding only - not by the storage format itself."""
    W_enc = oracle.bf16_weights(sd) if control else sd
    hs_o, fl = oracle.speech_encoder_forward(W_enc, o_arch, wavs, store=oracle.bf16_store if control else None)
    W = {k: v.clone().requires_grad_(True) for k, v in head_W.items()}
    w = ws_w.clone().requires_grad_(True)
    feat = oracle.weighted_sum(w, [h.detach() for h in hs_o], normalize)
    if control:
        feat = oracle.bf16_store(feat)
    feat.retain_grad()
    e = oracle.parallel_branch_forward(W, feat, fl, nhead=8)
    a = e / e.norm(dim=-1, keepdim=True)
    i = img / img.norm(dim=-1, keepdim=True)
    loss = oracle.masked_contrastive_loss(a, i, ids)
    loss.backward()
    hn = [F.layer_norm(h.detach(), (h.shape[-1],)) if normalize else h.detach() for h in hs_o]
    ws = torch.softmax(ws_w, 0)
    centre = sum(wn * h for wn, h in zip(ws, hn))
    floor = float(sum((wn * (feat.grad * (h - centre)).norm()) ** 2 for wn, h in zip(ws, hn)) ** 0.5)
    return {"loss": loss.detach(), "a": a.detach(), "W": W, "w": w, "floor": floor, "hs": hs_o}


def _step_errors(prod, ref):
    loss, a, grads, gw = prod
    errs = {n: rel_l2(g, ref["W"][n].grad) for n, g in grads.items()
            if ref["W"][n].grad is not None and float(ref["W"][n].grad.norm()) > 1e-7}
    return {"loss": abs(loss - ref["loss"].item()), "cos": float(F.cosine_similarity(a, ref["a"], dim=-1).min()), "head": errs,
            "wsum_rel": rel_l2(gw, ref["w"].grad), "wsum_e": float((gw - ref["w"].grad).norm()) / _wsum_floor(ref),
            "kappa": _wsum_floor(ref) / float(ref["w"].grad.norm())}


# Criteria of test_large_parallel_train_step, FIXED BEFORE THE FIRST RUN of this form of the test (round 6).
LP_HEAD_GRAD = 6e-2        # every head parameter gradient, rel-L2, every seed, vs the bf16-storage-emulated control (rounds 1-5's bound vs fp32)
LP_LOSS = 5e-3             # |loss - control|
LP_COS = 0.999             # unit-norm embeddings, cosine vs control
LP_HIDDEN = 2e-2           # hidden states, rel-L2 per layer vs control, and shipped build vs exact-GELU build
# weighted-sum logit gradient, in units of R = rms over the seeds of e(control, fp32) - what bf16 storage ALONE does to this quantity,
# computed by the two oracles on the CPU inside the test:
LP_WSUM_RMS = 1.0          # rms over the seeds of e(HIP, control) <= R: the kernels add no more than the storage format does
LP_WSUM_MAX = 3.0          # every seed: e(HIP, control) <= 3 R
LP_PAIR_RMS = 2.0 ** 0.5    # shipped five-term GELU build vs exact-GELU build: rms e(shipped, exact) <= sqrt(2) R (derivation: doc-string,
                           # "Paired activation check"; rounds 6a's value 1.0 was the EXPECTED value of the statistic for two equally good
                           # builds with independent noise, i.e. a coin flip), every seed <= 3 R
LP_PAIR_LOSS = 1e-3        # |loss(shipped) - loss(exact build)| per seed; the MEAN over the seeds <= 3e-4 (the rejected three-term fit moved the
                           # loss by 6e-4 systematically)
LP_PAIR_T = 3.0            # paired t statistic of e(shipped, control) - e(exact, control) over the seeds


def test_large_parallel_train_step():
    """Parallel-large recipe (HuBERT-large at reduced depth, normalised hidden states, 1024-wide head, E = 768): loss and gradients of
    one step over 12 data seeds (8..19), the SAME seeds through the shipped build and through the exact-GELU checker build
    (libspeechclip_hip_gelu_exact.so: A&S 7.1.28 erf-GELU at every bf16 site, nothing else differs) in one process.

    Control.  The bf16-storage-emulated oracle (_oracle_step(control=True)); the fp32 oracle's numbers are printed beside it as a
    diagnostic (they mix the storage format's error with the kernels').

    Weighted-sum logit gradient.  g_n = w_n <G, c_n> with c_n = LN(h_n) - sum_m w_m LN(h_m) the CENTRED states of a residual stream:
    three nearly equal tensors, so <G, c_n> is a sum of 1e5 terms of either sign that largely cancel.  Independent rounding noise of
    relative size r in the states moves g by about r * floor, floor = sqrt(sum_n w_n^2 ||G (.) c_n||^2) (_wsum_floor), however small |g|
    comes out; so rel-L2(g) = e * kappa with e = ||g - g_ref|| / floor and kappa = floor / ||g|| the cancellation factor of that seed's data
    (printed; on the CPU the control itself reaches rel-L2 0.12 against fp32 at kappa 19 with e = 6e-3).  rel-L2 is heavy-tailed over seeds
    for ANY implementation because kappa is; e is not (e ~ r |z|, z standard normal).  Rounds 1-5 bounded rel-L2 itself (6e-2 over four
    seeds, then median / cap over twelve after a red run - VERDICT r05 weak 2).  This form has no number taken from a GPU run: the scale
    is R = rms_seeds e(control, fp32), what bf16 storage alone does, from the two oracles on the CPU, and the HIP path must sit no farther
    from the control than that (LP_WSUM_RMS, LP_WSUM_MAX) - the criterion tests/test_gpu_recall.py uses for the embeddings.

    Paired activation check.  Both builds run the same seeds; a SYSTEMATIC activation error shows as (a) the builds'
    gradients apart by more than the storage noise (LP_PAIR_RMS), (b) a loss offset common to the seeds (mean <= 3e-4), (c) the shipped
    build systematically farther from the control (paired t <= 3), (d) hidden states apart by more than the tolerance.  Element level,
    high power: tests/test_gpu_kernels.py::test_gelu_fit_vs_exact_build_elementwise.

    LP_PAIR_RMS, derived (round 6, committed with the two-pass conv-0 kernel as the default, on which the statistic reads 1.11e-2 and
    passes the old and the new value alike).  g_ship - g_exact = (g_ship - g_ctrl) - (g_exact - g_ctrl).  Each bracket is bounded in rms by
    LP_WSUM_RMS R = R.  The two builds share their inputs but NOT their rounding decisions: one bf16 value that rounds the other way in a
    conv or FFN activation re-draws the noise of everything behind it, so the brackets are at most partially correlated, and for
    uncorrelated brackets rms e(shipped, exact) = sqrt(e_ship^2 + e_exact^2) <= sqrt(2) R.  The first version of this test asked <= 1.0 R,
    which with the measured e_ship = 1.2e-2, e_exact = 1.0 .. 1.2e-2, R = 1.62e-2 is the statistic's own expected value
    sqrt(1.2^2 + 1.1^2) = 1.63e-2 - a coin flip.  Evidence that the statistic is a noise realisation and not a property of the GELU fit:
    three conv-0 kernels of the layer_norm extractor that are equally accurate against fp64 (tools/bench_conv0ln.py: max error 1.56e-2,
    rel-L2 1.66e-3 each; they differ from one another in 2e-5 of the bf16 values, BOTH builds of a pair using the same one) read 1.11e-2
    (two-pass wave reductions), 1.63e-2 (closed-form statistics, the two-pass kernel's association), 1.87e-2 (closed form, regrouped
    affine).  A systematic activation bias b adds as sqrt(noise^2 + b^2) and is still caught at b ~ R; (b), (c), (d) and the element-level
    test keep their bounds."""
    import oracle
    from speechclip_plus_amd import _lib
    model, sd, o_arch, head_W = _large_parallel_model()
    ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    enc = model.audio_encoder
    rows = []
    for seed in range(8, 20):
        wavs, img, ids, batch = _large_parallel_data(seed)
        ship = _product_step(model, batch)
        with torch.no_grad():
            hs_ship = [h.float().cpu() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
        with _lib.using_library(_lib.GELU_EXACT_LIB_PATH):
            exact = _product_step(model, batch)
            with torch.no_grad():
                hs_exact = [h.float().cpu() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
        ctrl = _oracle_step(oracle, sd, o_arch, head_W, ws_w, wavs, img, ids, normalize=True, control=True)
        fp32 = _oracle_step(oracle, sd, o_arch, head_W, ws_w, wavs, img, ids, normalize=True)
        r = {"seed": seed, "ship": _step_errors(ship, ctrl), "exact": _step_errors(exact, ctrl), "ship_fp32": _step_errors(ship, fp32),
             "ctrl_fp32_e": float((ctrl["w"].grad - fp32["w"].grad).norm()) / _wsum_floor(fp32),
             "pair_e": float((ship[3] - exact[3]).norm()) / _wsum_floor(ctrl), "pair_loss": ship[0] - exact[0],
             "hs_ship": max(rel_l2(a, b) for a, b in zip(hs_ship, ctrl["hs"])), "hs_exact": max(rel_l2(a, b) for a, b in zip(hs_exact, ctrl["hs"])),
             "hs_pair": max(rel_l2(a, b) for a, b in zip(hs_ship, hs_exact)), "hs_ctrl_fp32": max(rel_l2(a, b) for a, b in zip(ctrl["hs"], fp32["hs"]))}
        rows.append(r)
        print("seed %2d kappa %5.1f | wsum e vs control: shipped %.2e exact-GELU %.2e pair %.2e (control vs fp32 %.2e) | rel-L2 vs control %.3f vs fp32 "
              "%.3f | head max vs control %.3f vs fp32 %.3f | hidden: shipped %.4f exact %.4f pair %.4f (control vs fp32 %.4f) | loss d %.1e pair %.1e" %
              (seed, r["ship"]["kappa"], r["ship"]["wsum_e"], r["exact"]["wsum_e"], r["pair_e"], r["ctrl_fp32_e"], r["ship"]["wsum_rel"],
               r["ship_fp32"]["wsum_rel"], max(r["ship"]["head"].values()), max(r["ship_fp32"]["head"].values()), r["hs_ship"], r["hs_exact"],
               r["hs_pair"], r["hs_ctrl_fp32"], r["ship"]["loss"], r["pair_loss"]))
    rms = lambda xs: (sum(x * x for x in xs) / len(xs)) ** 0.5
    R = rms([r["ctrl_fp32_e"] for r in rows])
    print("R = rms e(control, fp32) = %.3e; rms e(shipped, control) = %.3e, rms e(exact, control) = %.3e, rms e(shipped, exact) = %.3e" %
          (R, rms([r["ship"]["wsum_e"] for r in rows]), rms([r["exact"]["wsum_e"] for r in rows]), rms([r["pair_e"] for r in rows])))
    for r in rows:
        for build in ("ship", "exact"):
            e = r[build]
            assert e["cos"] > LP_COS and e["loss"] < LP_LOSS, (r["seed"], build, e["cos"], e["loss"])
            assert not {k: v for k, v in e["head"].items() if v > LP_HEAD_GRAD}, (r["seed"], build, e["head"])
            assert e["wsum_e"] <= LP_WSUM_MAX * R, (r["seed"], build, e["wsum_e"], R)
        assert max(r["hs_ship"], r["hs_exact"], r["hs_pair"]) <= LP_HIDDEN, (r["seed"], r["hs_ship"], r["hs_exact"], r["hs_pair"])
        assert r["pair_e"] <= LP_WSUM_MAX * R and abs(r["pair_loss"]) <= LP_PAIR_LOSS, (r["seed"], r["pair_e"], r["pair_loss"])
    for build in ("ship", "exact"):
        assert rms([r[build]["wsum_e"] for r in rows]) <= LP_WSUM_RMS * R, (build, R)
    assert rms([r["pair_e"] for r in rows]) <= LP_PAIR_RMS * R
    diff = torch.tensor([r["ship"]["wsum_e"] - r["exact"]["wsum_e"] for r in rows], dtype=torch.float64)
    t = float(diff.mean() / (diff.std() / len(rows) ** 0.5 + 1e-30))
    mean_dloss = sum(r["pair_loss"] for r in rows) / len(rows)
    print("paired over %d seeds: mean e(shipped) - e(exact) = %.2e (t = %.2f), mean loss difference %.2e" % (len(rows), float(diff.mean()), t, mean_dloss))
    assert t <= LP_PAIR_T, (t, diff.tolist())
    assert abs(mean_dloss) <= 3e-4, mean_dloss


# Bounds of test_train_step_parity against the bf16-storage-emulated control, fixed before its first run in this form (round 6)
TS_GRAD = 6e-2             # every head parameter gradient rel-L2 (rounds 1-5's bound, then against fp32)
TS_WSUM = 3.0              # the 13 weighted-sum logits: e(HIP, control) <= 3 r, e = || g - g_ref || / floor (test_large_parallel_train_step) and
                           # r = the worst layer's rel-L2 between the control's and the fp32 oracle's hidden states (what bf16 storage does to
                           # the states; e ~ r |z| with z standard normal, so 3 r is the three-sigma bound of an implementation AS noisy as the
                           # storage format - one batch, hence no rms over seeds)


def test_train_step_parity(setup):
    """Base recipe, one train step: loss, embeddings and every trainable gradient against the bf16-storage-emulated control
    (_oracle_step(control=True)); the fp32 oracle's figures are printed as a diagnostic and keep SURVEY 8d's loss / cosine tolerances."""
    model, sd, o_arch, head_W, oracle = setup
    g = torch.Generator().manual_seed(5)
    lens = [12000, 8000, 12000, 5000, 12000, 10300]
    wavs = [torch.randn(l, generator=g) for l in lens]
    B = len(lens)
    img = torch.randn(B, 512, generator=g)
    ids = torch.tensor([0, 0, 1, 2, 2, 3])
    L = max(lens)
    wav = torch.zeros(B, L)
    for b, x in enumerate(wavs):
        wav[b, : len(x)] = x
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens), "image": img.cuda(), "id": ids.cuda()}
    prod = _product_step(model, batch)
    ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    o_ctrl = _oracle_step(oracle, sd, o_arch, head_W, ws_w, wavs, img, ids, control=True)
    o_fp32 = _oracle_step(oracle, sd, o_arch, head_W, ws_w, wavs, img, ids)
    ctrl, fp32 = _step_errors(prod, o_ctrl), _step_errors(prod, o_fp32)
    R = max(rel_l2(a, b) for a, b in zip(o_ctrl["hs"], o_fp32["hs"]))
    print("train step vs control: head grads max %.3f, wsum e %.2e (r = %.2e; rel-L2 %.3f, kappa %.1f), loss d %.1e | vs fp32: head max "
          "%.3f, wsum rel-L2 %.3f, loss d %.1e" % (max(ctrl["head"].values()), ctrl["wsum_e"], R, ctrl["wsum_rel"], ctrl["kappa"], ctrl["loss"],
                                                   max(fp32["head"].values()), fp32["wsum_rel"], fp32["loss"]))
    for e in (ctrl, fp32):
        assert e["cos"] > 0.999 and e["loss"] < 5e-3, (e["cos"], e["loss"])
    bad = {k: v for k, v in ctrl["head"].items() if v > TS_GRAD}
    assert not bad, bad
    assert ctrl["wsum_e"] <= TS_WSUM * R, (ctrl["wsum_e"], R, ctrl["kappa"])
    bad = {k: v for k, v in fp32["head"].items() if v > 6e-2}          # rounds 1-5's fp32 comparison, kept
    assert not bad and fp32["wsum_rel"] <= 6e-2, (bad, fp32["wsum_rel"])


def test_encode_speech_api_and_recall(setup):
    """encode_speech on a synthetic Flickr8k-shaped eval set (5 utterances per image): recall@k from the HIP
    embeddings must equal recall@k from the oracle embeddings."""
    model, sd, o_arch, head_W, oracle = setup
    from speechclip_plus_amd import mutualRetrieval
    g = torch.Generator().manual_seed(21)
    n_img, per = 8, 5
    lens = [int(x) for x in torch.randint(4000, 9000, (n_img * per,), generator=g)]
    wavs = [torch.randn(l, generator=g) for l in lens]
    ids = torch.arange(n_img).repeat_interleave(per)
    ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    with torch.no_grad():
        emb = torch.cat([model.encode_speech([w.cuda() for w in wavs[i:i + 8]])["parallel_audio_feat"]
                         for i in range(0, len(wavs), 8)]).float().cpu()
        emb_o = []
        for i in range(0, len(wavs), 8):
            hs_o, fl = oracle.speech_encoder_forward(sd, o_arch, wavs[i:i + 8])
            emb_o.append(oracle.parallel_branch_forward(head_W, oracle.weighted_sum(ws_w, hs_o), fl, nhead=8))
        emb_o = torch.cat(emb_o)
    assert emb.shape == (n_img * per, 512)
    assert float(F.cosine_similarity(emb, emb_o, dim=-1).min()) > 0.999
    # images: noisy mean of their captions' oracle embeddings -> recall neither 0 nor 100
    a_o = emb_o / emb_o.norm(dim=-1, keepdim=True)
    img = torch.stack([a_o[ids == k].mean(0) for k in range(n_img)]) + 0.12 * torch.randn(n_img, 512, generator=g)
    img = img / img.norm(dim=-1, keepdim=True)
    a = emb / emb.norm(dim=-1, keepdim=True)
    r_hip = mutualRetrieval(a @ img.T, (a @ img.T).T, ids, torch.arange(n_img), [1, 5, 10])
    r_ora = oracle.mutual_retrieval(a_o @ img.T, (a_o @ img.T).T, ids, torch.arange(n_img), [1, 5, 10])
    for d_hip, d_ora in zip(r_hip, r_ora):
        for k in d_ora:
            assert abs(d_hip[k] - d_ora[k]) < 1e-6, (k, d_hip[k], d_ora[k])
    h13 = model.feature_extractor_s3prl([w.cuda() for w in wavs[:2]])
    # kwClip.py:965-997: 13 HuBERT states + the parallel branch layer's output (CLS position dropped)
    assert len(h13[1]) == 14 and h13[0].shape[-1] == 768 and h13[1][-1].shape == h13[1][0].shape


def test_trainer_side_stream_schedule_matches_synchronous():
    """train.ContrastiveTrainer enqueues all-reduce + Adam on a side stream and joins it before the first trainable module of
    the next step: three steps must leave exactly the parameters of the same steps run on one stream, and the gradients of the
    last step must still be readable."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=2)
    sd = random_hubert_state_dict(arch, seed=3)
    g = torch.Generator().manual_seed(11)
    B, L = 6, 9000
    batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.tensor([9000, 7000, 9000, 5000, 8000, 9000]),
             "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2, 3, 4]).cuda()}
    finals = []
    for sync in (False, True):
        torch.manual_seed(3)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        cfg.cl_loss.args.temperature_trainable = True
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        trainer = ContrastiveTrainer(model)
        if sync:
            trainer.side = None
        losses = [float(trainer.step(batch)) for _ in range(3)]
        torch.cuda.synchronize()
        assert float(trainer.opt.flat_g.abs().sum()) > 0
        finals.append((losses, trainer.opt.flat_p.clone(), trainer.opt.flat_g.clone()))
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    assert finals[0][0][2] < finals[0][0][0]


@pytest.mark.parametrize("kind", ["parallel", "cascaded", "unfrozen_top"])
def test_encoder_on_its_own_stream_under_the_previous_steps_tail(kind):
    """speech_encoder._encode_overlapped: the frozen encoder of step N + 1 runs on a stream of its own while step N's branch / head /
    loss / backward kernels are still in flight, on two alternating sets of resident buffers.  Five steps over DIFFERENT batches (so a
    stale or prematurely overwritten buffer changes a loss), ragged lengths that change the row layout from step to step: losses,
    parameters and last gradients must equal the single-stream schedule's bit for bit."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=3 if kind == "unfrozen_top" else 2)
    sd = random_hubert_state_dict(arch, seed=3)
    g = torch.Generator().manual_seed(23)
    B = 6
    lens = [[24000, 17000, 24000, 9000, 20000, 24000], [24000, 24000, 12000, 24000, 8000, 15000], [16000, 24000, 24000, 24000, 11000, 7000],
            [24000, 17000, 24000, 9000, 20000, 24000], [24000, 6000, 24000, 24000, 24000, 13000]]
    batches = [{"wav": torch.randn(B, 24000, generator=g).cuda(), "wav_len": torch.tensor(l), "image": torch.randn(B, 512, generator=g).cuda(),
                "id": torch.arange(B).cuda()} for l in lens]
    torch.cuda.synchronize()
    for i, b in enumerate(batches):          # resident inputs, marked ready (the encoder stream then runs a step ahead); one batch stays
        if i != 2:                           # unmarked (waits for the caller's stream) and one arrives as a host tensor (copied by the
            b["wav"]._sc_ready = True        # encoder stream itself)
    batches[4]["wav"] = batches[4]["wav"].cpu().pin_memory()
    finals = []
    for overlap in (False, True):
        torch.manual_seed(3)
        if kind in ("parallel", "unfrozen_top"):
            cfg = base_parallel_config()
            cfg.audio_encoder.max_audio_len = -1
            if kind == "unfrozen_top":          # conv stack + layer 0 run ahead on the encoder stream, layers 1-2 wait for the optimiser
                cfg.audio_encoder.trainable = True
                cfg.audio_encoder.unfreeze_layers = [1, 2]
                cfg.audio_encoder.optim.args.lr = 1e-3
            model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch)
        else:
            from speechclip_plus_amd import cascaded_plus_base_config
            cfg = cascaded_plus_base_config()
            cfg.audio_encoder.max_audio_len = -1
            cfg.clip.layers = 2
            model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch)
        model = set_dropout(model.train(), False)
        model.audio_encoder.enc_overlap = overlap
        trainer = ContrastiveTrainer(model)
        losses = [float(trainer.step(b)) for b in batches]
        torch.cuda.synchronize()
        finals.append((losses, trainer.opt.flat_p.clone(), trainer.opt.flat_g.clone()))
        assert (model.audio_encoder._enc_stream is not None) == overlap
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    assert len(set(finals[0][0])) == len(batches)


def test_encoder_stream_with_changing_padded_lengths_and_plan_eviction():
    """The overlapped schedule when the padded length changes from step to step (five 2-second buckets x two alternating plans = 10
    resident plans against a cache of 8: least-recently-used plans are evicted and released two forwards later) - 14 steps must
    reproduce the single-stream schedule bit for bit."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=2)
    sd = random_hubert_state_dict(arch, seed=5)
    g = torch.Generator().manual_seed(29)
    B = 4
    longest = [24000, 50000, 80000, 110000, 150000]
    batches = []
    for i in range(14):
        L = longest[i % 5]
        lens = torch.randint(L // 3, L + 1, (B,), generator=g)
        lens[i % B] = L
        batches.append({"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": lens, "image": torch.randn(B, 512, generator=g).cuda(),
                        "id": torch.arange(B).cuda()})
        batches[-1]["wav"]._sc_ready = True
    torch.cuda.synchronize()
    finals = []
    for overlap in (False, True):
        torch.manual_seed(5)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        enc = model.audio_encoder
        enc.enc_overlap = overlap
        trainer = ContrastiveTrainer(model)
        losses = [float(trainer.step(b)) for b in batches]
        torch.cuda.synchronize()
        finals.append((losses, trainer.opt.flat_p.clone()))
        if overlap:
            assert len(enc._plans) == 8 and len(enc._retired) <= 3
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1])


def test_accumulate_grad_batches_two_micro_steps_make_one_optimiser_step():
    """trainer.accumulate_grad_batches: 2 (config/speechCLIP+/model_large/coco/spchclip_h+.yaml:138, Lightning semantics): the first
    micro-step only back-propagates loss / 2 (no optimiser step, parameters and global_step unchanged), the second adds its loss / 2
    and steps once - with the gradient a plain trainer accumulates from the same two batches by hand."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=2)
    sd = random_hubert_state_dict(arch, seed=3)
    g = torch.Generator().manual_seed(13)
    B, L = 6, 9000
    mk = lambda: {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.tensor([9000, 7000, 9000, 5000, 8000, 9000]),
                  "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2, 3, 4]).cuda()}
    b1, b2 = mk(), mk()

    def build(acc):
        torch.manual_seed(3)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        cfg.trainer.accumulate_grad_batches = acc
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        return model, ContrastiveTrainer(model)
    model, tr = build(2)
    p0 = tr.opt.flat_p.clone()
    tr.step(b1)
    torch.cuda.synchronize()
    assert model.global_step == 0 and torch.equal(tr.opt.flat_p, p0)           # inside the window: nothing stepped
    g_half = tr.opt.flat_g.clone()
    assert float(g_half.abs().sum()) > 0
    tr.step(b2)
    torch.cuda.synchronize()
    assert model.global_step == 1 and not torch.equal(tr.opt.flat_p, p0)
    g_acc = tr.opt.flat_g.clone()
    # by hand: the two gradients of a plain trainer at the SAME (initial) parameters, halved and added
    grads = []
    for b in (b1, b2):
        m1, t1 = build(1)
        t1.step(b)
        torch.cuda.synchronize()
        grads.append(t1.opt.flat_g.clone())
    ref = 0.5 * grads[0] + 0.5 * grads[1]
    assert rel_l2(g_half, 0.5 * grads[0]) < 1e-6
    assert rel_l2(g_acc, ref) < 1e-5, rel_l2(g_acc, ref)
    # and the window closes: the next micro-step starts from a zeroed buffer
    tr.step(b1)
    torch.cuda.synchronize()
    assert model.global_step == 1 and rel_l2(tr.opt.flat_g, g_half) < 0.2      # (parameters moved one Adam step: close, not equal)


def test_unfrozen_layers_follow_the_optimiser_and_side_stream_matches_synchronous():
    """Unfrozen HuBERT layers over several optimiser steps: (1) the bf16 working copies the forward / dgrad kernels read are the
    CURRENT fp32 masters after every step (sc_adam_f32 writes through a raw pointer: no tensor version changes, the copies key on
    optim.param_generation()); (2) the loss keeps falling over 4 steps on one batch (stale forward weights made it stall or
    diverge); (3) the side-stream schedule (join at the top of the encoder forward when layers are unfrozen) leaves bitwise the
    parameters of the one-stream schedule."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=3)
    sd = random_hubert_state_dict(arch, seed=4)
    g = torch.Generator().manual_seed(12)
    B, L = 6, 9000
    batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.tensor([9000, 7000, 9000, 5000, 8000, 9000]),
             "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2, 3, 4]).cuda()}
    finals = []
    for sync in (False, True):
        torch.manual_seed(4)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        cfg.audio_encoder.trainable = True
        cfg.audio_encoder.unfreeze_layers = [1, 2]
        cfg.audio_encoder.optim.args.lr = 1e-3              # large enough for the bf16 copies to move every step
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        trainer = ContrastiveTrainer(model)
        if sync:
            trainer.side = None
        tl = model.audio_encoder.train_layers
        losses, prev = [], None
        for step in range(4):
            losses.append(float(trainer.step(batch)))
            torch.cuda.synchronize()
            # the copies used by the step that just ran were built from the masters as they stood BEFORE its Adam update
            w_now = tl.get(2, "fc1.weight").detach().to(torch.bfloat16)
            if prev is not None:
                assert torch.equal(tl._copies[2]["fc1_w"], prev), f"step {step}: forward ran on stale bf16 weights"
                assert torch.equal(tl._copies[2]["fc1_wT"], prev.t().contiguous())
            assert not torch.equal(w_now, tl._copies[2]["fc1_w"]), "lr too small for this check"
            prev = w_now.clone()
        assert losses[3] < losses[1] < losses[0], losses
        finals.append((losses, trainer.opt.flat_p.clone()))
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1])


@pytest.mark.parametrize("hubert_dropout", [False, True])
def test_unfrozen_hubert_layers_gradients_vs_oracle(hubert_dropout):
    """audio_encoder.trainable with unfreeze_layers = top two of a 3-layer HuBERT: loss and the gradients of every parameter of
    the unfrozen layers (attention backward, LayerNorm / GELU backward, dgrad and weight-gradient GEMMs) against the oracle's
    autograd through the same layers."""
    import dataclasses
    import oracle
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert"], layers=3)
    sd = random_hubert_state_dict(arch, seed=9)
    torch.manual_seed(9)
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    cfg.audio_encoder.trainable = True
    cfg.audio_encoder.unfreeze_layers = [1, 2]
    model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(torch.tensor([0.2, -0.1, 0.4, 0.3]))
    tl = model.audio_encoder.train_layers
    assert {id(p) for p in tl.parameters()} <= {id(p) for p in model.getTrainableParams()}
    g = torch.Generator().manual_seed(6)
    lens = [9000, 6100, 9000, 4100, 8000]
    wavs = [torch.randn(l, generator=g) * 0.5 for l in lens]
    B = len(lens)
    img = torch.randn(B, 512, generator=g)
    ids = torch.tensor([0, 1, 1, 2, 3])
    wav = torch.zeros(B, max(lens))
    for b, x in enumerate(wavs):
        wav[b, : len(x)] = x
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens), "image": img.cuda(), "id": ids.cuda()}
    enc = model.audio_encoder
    drop = None
    if hubert_dropout:          # fairseq's train-mode dropout in the frozen AND the unfrozen layers; masks rebuilt on the host
        enc.hubert_dropout = True
        torch.manual_seed(31)
        enc._drop_calls = 0
        seed_of = enc._dropout_seeds()
        enc._drop_calls = 0
    losses_, _, _ = model(batch)
    out = model.compute_loss(losses_)
    out["loss"].backward()
    if hubert_dropout:
        pl_ = enc._plan(B, max(lens))
        drop = _encoder_mask_hook(seed_of, B, pl_.T, 768, 12, pl_.R)
    # oracle: same weights, layers 1 and 2 require grad
    o_arch = oracle.HubertArch.base()
    o_arch.layers = 3
    W = {k: v.clone().float() for k, v in sd.items()}
    names = [n for n in W if n.startswith("encoder.layers.1.") or n.startswith("encoder.layers.2.")]
    for n in names:
        W[n].requires_grad_(True)
    hs_o, fl = oracle.speech_encoder_forward(W, o_arch, wavs, drop=drop)
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    feat = oracle.weighted_sum(ws_w, list(hs_o), False)
    e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
    loss_o = oracle.masked_contrastive_loss(e / e.norm(dim=-1, keepdim=True), img / img.norm(dim=-1, keepdim=True), ids)
    loss_o.backward()
    assert abs(out["loss"].item() - loss_o.item()) < 5e-3
    errs = {}
    for key, fname in tl.fairseq_names.items():
        ref = W[fname].grad
        got = tl.p[key].grad
        assert got is not None and ref is not None, fname
        if float(ref.norm()) > 1e-8:
            errs[fname] = rel_l2(got, ref)
    bad = {k: v for k, v in errs.items() if v > 8e-2}
    assert not bad, bad
    print("unfrozen-layer grads: max rel-L2 %.3g" % max(errs.values()))


def test_reference_checkpoint_round_trip_and_validation_epoch(setup):
    """f4: a checkpoint in the reference's key naming (audio_encoder.encoder.<fairseq keys>, branch / criterion keys as is) is
    rebuilt into an identical model; validation_step / validation_epoch_end reproduce the oracle's recall on the gathered
    features (one image embedding per id)."""
    import oracle
    from speechclip_plus_amd import KWClip_GeneralTransformer, base_parallel_config
    model, sd, o_arch, head_W, _ = setup
    ref_ckpt = {("audio_encoder.encoder." + k): v for k, v in sd.items()}
    ref_ckpt.update({k: v for k, v in model.state_dict().items() if not k.startswith("audio_encoder.encoder.")})
    ref_ckpt["clip.model.visual.proj"] = torch.zeros(3)                       # image tower keys are dropped
    cfg = base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    torch.manual_seed(123)                                                    # different init: everything must come from the checkpoint
    m2 = KWClip_GeneralTransformer.from_reference_checkpoint(cfg, ref_ckpt, device="cuda:0").eval()
    assert "parallel_branch.cls" in m2._reference_load_report["loaded"]
    g = torch.Generator().manual_seed(77)
    n_img, per = 12, 3
    outs_a, outs_b = [], []
    for chunk in range(3):
        B = n_img * per // 3
        ids = (torch.arange(B) + chunk * B) // per
        batch = {"wav": (torch.randn(B, 8000, generator=g) * 0.3).cuda(), "wav_len": torch.full((B,), 8000),
                 "image": torch.randn(B, 512, generator=g).cuda(), "id": ids.cuda()}
        # the reference's hook chain (kwClip.py:195-285): validation_step -> validation_step_end -> list for the epoch end
        step_out = model.validation_step(batch, chunk)
        assert set(step_out) == {"loss_feats", "log_metrics", "others"}
        outs_a.append(model.validation_step_end(step_out))
        outs_b.append(m2.validation_step_end(m2.validation_step(batch, chunk)))
    assert {"val_loss", "val_p_cl_loss", "val_cl_temp"} <= set(model.logged)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a["parallel_audio_feat"], b["parallel_audio_feat"])
    rec = model.validation_epoch_end(outs_a)
    ids = torch.cat([o["id"] for o in outs_a]).cpu()
    audio = torch.cat([o["parallel_audio_feat"] for o in outs_a]).float().cpu()
    imgs = torch.cat([o["image_feat"] for o in outs_a]).float().cpu()
    pairs = {}
    for i, v in zip(ids.tolist(), imgs):
        pairs[i] = v
    img_ids, img_feats = torch.tensor(list(pairs)), torch.stack(list(pairs.values()))
    score = audio @ img_feats.t()
    ref = oracle.mutual_retrieval(score, score.t(), ids, img_ids, model.recall_at)
    for got, want in zip(rec, ref):
        assert {k: round(float(v), 4) for k, v in got.items()} == {k: round(float(v), 4) for k, v in want.items()}
    # example.py:9-10: load_from_checkpoint(path) on a Lightning-style file (state_dict + hyper_parameters.config)
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".ckpt") as f:
        torch.save({"state_dict": ref_ckpt, "hyper_parameters": {"config": dict(cfg)}}, f.name)
        m3 = KWClip_GeneralTransformer.load_from_checkpoint(f.name, device="cuda:0").eval()
    out3 = m3.validation_step_end(m3.validation_step(batch, 0))
    assert torch.equal(out3["parallel_audio_feat"], outs_a[-1]["parallel_audio_feat"])


@pytest.mark.parametrize("large", [False, True])
def test_fully_trainable_hubert_gradients_vs_oracle(large):
    """audio_encoder.trainable: true with no reinit / unfreeze list (speech_encoder_plus.py:556-562): EVERY encoder parameter
    trains - conv feature extractor (GroupNorm / LayerNorm variants), feature LayerNorm, post_extract_proj, the weight-normalised
    positional conv (weight_g, weight_v), encoder LayerNorm and all transformer layers; fairseq's feature_grad_mult scales the
    gradient entering the extractor.  One step: loss and every gradient against the oracle's autograd."""
    import dataclasses
    import oracle
    from oracle.hubert_ref import fold_weight_norm
    from speechclip_plus_amd import (set_dropout, KWClip_GeneralTransformer, base_parallel_config, large_parallel_config,
                                     random_hubert_state_dict)
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert_large_ll60k" if large else "hubert"], layers=2, feature_grad_mult=0.1)
    sd = random_hubert_state_dict(arch, seed=19)
    torch.manual_seed(19)
    cfg = large_parallel_config() if large else base_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    cfg.audio_encoder.trainable = True
    model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
    enc = model.audio_encoder
    assert enc.frontend is not None and enc.train_layers.ids == [0, 1]
    with torch.no_grad():
        enc.weightedsum_layer.weights.copy_(torch.tensor([0.2, -0.1, 0.4]))
    g = torch.Generator().manual_seed(8)
    lens = [9000, 6100, 9000, 4100]
    wavs = [torch.randn(l, generator=g) * 0.5 for l in lens]
    B = len(lens)
    E = 768 if large else 512
    img = torch.randn(B, E, generator=g)
    ids = torch.tensor([0, 1, 1, 2])
    wav = torch.zeros(B, max(lens))
    for b, x in enumerate(wavs):
        wav[b, : len(x)] = x
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens), "image": img.cuda(), "id": ids.cuda()}
    names = dict(enc.train_layers.fairseq_names)
    params = {enc.train_layers.fairseq_names[k]: p for k, p in enc.train_layers.p.items()}
    params.update({enc.frontend.fairseq_names[k]: p for k, p in enc.frontend.p.items()})
    assert {id(p) for p in params.values()} <= {id(p) for p in model.getTrainableParams()}
    out = model.compute_loss(model(batch)[0])
    out["loss"].backward()
    # ---- oracle: same weights, everything requires grad; pos_conv through its weight-norm parameters
    o_arch = oracle.HubertArch.large() if large else oracle.HubertArch.base()
    o_arch.layers, o_arch.feature_grad_mult = 2, 0.1
    W = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if k != "encoder.pos_conv.0.weight"}
    v0 = sd["encoder.pos_conv.0.weight"].clone().float()
    W["encoder.pos_conv.0.weight_g"] = v0.pow(2).sum(dim=(0, 1), keepdim=True).sqrt().requires_grad_(True)
    W["encoder.pos_conv.0.weight_v"] = v0.clone().requires_grad_(True)
    Wf = dict(W)
    Wf["encoder.pos_conv.0.weight"] = fold_weight_norm(W["encoder.pos_conv.0.weight_g"], W["encoder.pos_conv.0.weight_v"])
    hs_o, fl = oracle.speech_encoder_forward(Wf, o_arch, wavs)
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    ws_w = enc.weightedsum_layer.weights.detach().cpu()
    feat = oracle.weighted_sum(ws_w, list(hs_o), large)
    e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
    loss_o = oracle.masked_contrastive_loss(e / e.norm(dim=-1, keepdim=True), img / img.norm(dim=-1, keepdim=True), ids)
    loss_o.backward()
    assert abs(out["loss"].item() - loss_o.item()) < 5e-3
    errs = {}
    for name, p in params.items():
        ref = W[name].grad
        if ref is None:           # pre-LN: the encoder's final LayerNorm acts on an output the reference never reads
            assert name.startswith("encoder.layer_norm.") and (p.grad is None or float(p.grad.abs().sum()) == 0.0), name
            continue
        assert p.grad is not None, name
        if float(ref.norm()) > 1e-9:
            errs[name] = rel_l2(p.grad, ref)
    bad = {k: v for k, v in errs.items() if v > 8e-2}
    assert not bad, (bad, max(errs.values()))
    front = {k: v for k, v in errs.items() if not k.startswith("encoder.layers.")}
    print("fully trainable (%s): worst front-end gradient rel-L2 %.3g (%s), worst layer gradient %.3g" %
          ("large" if large else "base", max(front.values()), max(front, key=front.get),
           max(v for k, v in errs.items() if k.startswith("encoder.layers."))))


def test_second_forward_before_backward_fails_loudly_and_outputs_survive(setup):
    """The encoder's hidden states live in a resident workspace per (B, L) geometry: a backward that arrives after another
    forward re-used it must raise (not differentiate against the wrong states), and hidden states handed out to the caller are
    fresh tensors that a later forward does not change."""
    model, sd, o_arch, head_W, _ = setup
    g = torch.Generator().manual_seed(3)
    mk = lambda: {"wav": (torch.randn(3, 8000, generator=g) * 0.5).cuda(), "wav_len": torch.tensor([8000, 6000, 7000]),
                  "image": torch.randn(3, 512, generator=g).cuda(), "id": torch.arange(3).cuda()}
    b1, b2 = mk(), mk()
    model.zero_grad(set_to_none=True)
    loss1 = model.compute_loss(model(b1)[0])["loss"]
    model(b2)                                               # same geometry: overwrites the plan's hidden states
    with pytest.raises(RuntimeError, match="another forward"):
        loss1.backward()
    with torch.no_grad():
        _, _, hs1 = model.audio_encoder(b1["wav"], b1["wav_len"], return_hidden_states=True)
        keep = [h.clone() for h in hs1]
        model.audio_encoder(b2["wav"], b2["wav_len"])
    assert all(torch.equal(a, b) for a, b in zip(hs1, keep))


def test_lightning_shaped_optimizer_hooks_match_the_trainer():
    """configure_optimizers (kwClip.py:646-674): a torch.optim.Optimizer + a step-interval LambdaLR.  Driving them by hand
    (zero_grad, backward, clip_grad_norm_, step, scheduler.step - what Lightning does) must give the ContrastiveTrainer's
    parameters after the same steps."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    arch = dataclasses.replace(ARCHS["hubert"], layers=2)
    sd = random_hubert_state_dict(arch, seed=3)
    g = torch.Generator().manual_seed(13)
    B, L = 6, 9000
    batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.tensor([9000, 7000, 9000, 5000, 8000, 9000]),
             "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2, 3, 4]).cuda()}
    finals = []
    for how in ("trainer", "hooks"):
        torch.manual_seed(3)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        cfg.audio_encoder.scheduler.warmup = 2                      # the schedule must matter within 3 steps
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        if how == "trainer":
            trainer = ContrastiveTrainer(model)
            for _ in range(3):
                trainer.step(batch)
            torch.cuda.synchronize()
            finals.append(trainer.opt.flat_p.clone())
        else:
            (opt,), (sch,) = model.configure_optimizers()
            assert isinstance(opt, torch.optim.Optimizer) and sch["interval"] == "step"
            for _ in range(3):
                opt.zero_grad()
                out = model.training_step_end(model.training_step(batch))
                out["loss"].backward()
                torch.nn.utils.clip_grad_norm_(model.getTrainableParams(), float(cfg.trainer.gradient_clip_val))
                opt.step()
                sch["scheduler"].step()
            torch.cuda.synchronize()
            finals.append(opt.flat.flat_p.clone())
    assert float((finals[0] - finals[1]).abs().max()) < 2e-6 * float(finals[0].abs().max()), float((finals[0] - finals[1]).abs().max())


def test_optimizer_state_dict_resumes_bit_identically():
    """FlatAdamOptimizer keeps the Adam moments and the step counter in flat buffers outside ``Optimizer.state``; its
    state_dict / load_state_dict must carry them: two steps, checkpoint, one more step == restore into a fresh
    model + optimiser, that same third step - bit for bit (the step is deterministic with dropout off)."""
    import dataclasses
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, base_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert"], layers=1)
    sd = random_hubert_state_dict(arch, seed=3)
    g = torch.Generator().manual_seed(17)
    B, L = 4, 8000
    batch = {"wav": torch.randn(B, L, generator=g).cuda(), "wav_len": torch.tensor([8000, 6000, 8000, 5000]),
             "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2]).cuda()}

    def make():
        torch.manual_seed(3)
        cfg = base_parallel_config()
        cfg.audio_encoder.max_audio_len = -1
        model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
        (opt,), (sch,) = model.configure_optimizers()
        return model, opt, sch["scheduler"]

    def one_step(model, opt, sch):
        opt.zero_grad()
        model.training_step_end(model.training_step(batch))["loss"].backward()
        opt.step()
        sch.step()

    model, opt, sch = make()
    one_step(model, opt, sch)
    one_step(model, opt, sch)
    ckpt = {"model": {k: v.clone() for k, v in model.state_dict().items()}, "opt": opt.state_dict(), "sch": sch.state_dict()}
    assert ckpt["opt"]["flat_adam"]["step_count"] == 2 and float(ckpt["opt"]["flat_adam"]["v"].abs().sum()) > 0
    one_step(model, opt, sch)
    torch.cuda.synchronize()
    want = opt.flat.flat_p.clone()
    model2, opt2, sch2 = make()
    model2.load_state_dict(ckpt["model"])
    opt2.load_state_dict(ckpt["opt"])
    sch2.load_state_dict(ckpt["sch"])
    one_step(model2, opt2, sch2)
    torch.cuda.synchronize()
    assert torch.equal(opt2.flat.flat_p, want)
    with pytest.raises(KeyError):
        opt2.load_state_dict({k: v for k, v in ckpt["opt"].items() if k != "flat_adam"})


YAML_HYBRID_BASE = """
model_settings:
  cascaded_objective_weight: 1.0
  parallel_objective_weight: 1.0
  parallel_branch:
    transformer_args: {type: TransformerEncoder, n_layers: 1, d_model: 768, nhead: 8, dim_feedforward: 3072, dropout: 0.1,
                       activation: gelu, layer_norm_eps: 1.0e-5, batch_first: true, norm_first: false}
  cascaded_branch:
    type: HybridBranch_dynamic
    vq: {activation: gelu, type: SimpleVectorQuantizer, args: {temp: fixed=0.1, time_first: true, use_gumbel: false, hard: true}}
    downsampling:
      type: cif
      cif: {quantity_loss_weight: 0.25, using_gt_len: false, cif_output_dim: 768, encoder_embed_dim: 768, produce_weight_type: conv,
            cif_threshold: 1.0, conv_cif_layer_num: 1, conv_cif_width: 3, conv_cif_dropout: 0.1, apply_scaling: true,
            scaling_step: 5000, apply_tail_handling: true, tail_handling_firing_threshold: 0.5, add_cif_ctxt_layers: false}
    keyword: {detokenized_K_neighbors: 5, retrieve_method: cosine, batchnorms: {type: eachKw, std_scale: 1.0, learnable: true, parallel: true}}
    transformer_args: {type: MultiheadAttentionAndNorm, n_layers: 1, d_model: 768, nhead: 8, dim_feedforward: 3072, dropout: 0.1,
                       activation: gelu, layer_norm_eps: 1.0e-5, batch_first: true, norm_first: false}
cl_loss: {type: MaskedContrastiveLoss, args: {temperature: 0.07, temperature_trainable: true, margin: 0.0, dcl: false, a2b: true, b2a: true}}
retrieval: {audio_feat_src: parallel, recall_at: [1, 5, 10]}
clip: {name: ViT-B/32, image_encoder_trainable: false, text_encoder_trainable: false,
       reduce_subword_embbedding: ./avssl/data/flickr_stat/text_clip_vocab_usage_byfreq.npy}
audio_encoder:
  type: FairseqHubert
  name: hubert_base
  downsampling_rate: 320
  pretrained: true
  trainable: false
  feat_select_idx: weighted_sum
  layer_drop: 0.0
  max_audio_len: 102400
  optim: {name: Adam, args: {lr: 1.e-4, weight_decay: 1.e-6}}
  scheduler: {name: linear_warmup_decay, warmup: 5000, max_step: 50000, final_lr: 1.e-8}
trainer: {max_steps: 50000, gradient_clip_val: 4, accumulate_grad_batches: 1, precision: 16, strategy: dp}
"""


def test_model_from_reference_shaped_yaml_hybrid_plus_base():
    """A recipe written with the reference's yaml keys (the Hybrid+ base geometry: d_model 768 with 8 heads = head_dim 96, the
    shape the attention block pads to 128) builds the model and trains a step; the reduced-vocabulary table path that does not
    exist here raises unless the synthetic 8112-sub-word table is asked for."""
    import dataclasses
    from speechclip_plus_amd import KWClip_GeneralTransformer, load_config, random_hubert_state_dict, set_dropout
    from speechclip_plus_amd.speech_encoder import ARCHS
    from speechclip_plus_amd.train import ContrastiveTrainer
    with pytest.raises(FileNotFoundError):               # a real recipe must find its vocabulary table ...
        load_config(YAML_HYBRID_BASE)
    cfg = load_config(YAML_HYBRID_BASE, allow_synthetic_vocab=True)       # ... random-weight runs opt in to a synthetic one
    assert cfg.clip.embed_dim == 512 and cfg.clip.reduce_subword_embbedding.numel() == 8112
    cfg.clip.layers = 2
    arch = dataclasses.replace(ARCHS["hubert_base"], layers=2)
    torch.manual_seed(5)
    model = KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=random_hubert_state_dict(arch, seed=5), hubert_arch=arch)
    mha = model.cascaded_branch.self_att.multihead_attn_layer
    assert mha.embed_dim // mha.num_heads == 96
    set_dropout(model.train(), False)
    g = torch.Generator().manual_seed(2)
    B = 4
    batch = {"wav": (torch.randn(B, 30000, generator=g) * 0.5).cuda(), "wav_len": torch.tensor([30000, 21000, 30000, 12000]),
             "image": torch.randn(B, 512, generator=g).cuda(), "id": torch.tensor([0, 1, 1, 2]).cuda()}
    trainer = ContrastiveTrainer(model)
    l1, l2 = float(trainer.step(batch)), float(trainer.step(batch))
    assert l1 == l1 and l2 == l2 and l2 < l1 + 0.1


def test_unfrozen_layer_prelN_large_with_frozen_layer_above():
    """HuBERT-large layer order (pre-LN), normalised weighted sum, unfreeze_layers = [1] of 3 layers: layer 2 stays frozen but must
    carry the gradient down (input-gradient half of its backward).  Gradients of layer 1 against the oracle's autograd."""
    import dataclasses
    import oracle
    from speechclip_plus_amd import set_dropout, KWClip_GeneralTransformer, large_parallel_config, random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS
    arch = dataclasses.replace(ARCHS["hubert_large_ll60k"], layers=3)
    sd = random_hubert_state_dict(arch, seed=12)
    torch.manual_seed(12)
    cfg = large_parallel_config()
    cfg.audio_encoder.max_audio_len = -1
    cfg.audio_encoder.trainable = True
    cfg.audio_encoder.unfreeze_layers = [1]
    model = set_dropout(KWClip_GeneralTransformer(cfg, device="cuda:0", hubert_state_dict=sd, hubert_arch=arch).train(), False)
    with torch.no_grad():
        model.audio_encoder.weightedsum_layer.weights.copy_(torch.tensor([0.2, -0.1, 0.4, 0.3]))
    tl = model.audio_encoder.train_layers
    assert tl.ids == [1] and tl.pass_ids == [2]
    g = torch.Generator().manual_seed(16)
    lens = [9000, 6100, 9000, 4100]
    wavs = [torch.randn(l, generator=g) * 0.5 for l in lens]
    B = len(lens)
    img = torch.randn(B, 768, generator=g)
    ids = torch.tensor([0, 1, 1, 2])
    wav = torch.zeros(B, max(lens))
    for b, x in enumerate(wavs):
        wav[b, : len(x)] = x
    batch = {"wav": wav.cuda(), "wav_len": torch.tensor(lens), "image": img.cuda(), "id": ids.cuda()}
    losses_, _, _ = model(batch)
    out = model.compute_loss(losses_)
    out["loss"].backward()
    o_arch = oracle.HubertArch.large()
    o_arch.layers = 3
    W = {k: v.clone().float() for k, v in sd.items()}
    names = [n for n in W if n.startswith("encoder.layers.1.")]
    for n in names:
        W[n].requires_grad_(True)
    hs_o, fl = oracle.speech_encoder_forward(W, o_arch, wavs)
    head_W = {k: v.detach().cpu().float() for k, v in model.parallel_branch.state_dict().items()}
    ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
    feat = oracle.weighted_sum(ws_w, list(hs_o), True)
    e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
    loss_o = oracle.masked_contrastive_loss(e / e.norm(dim=-1, keepdim=True), img / img.norm(dim=-1, keepdim=True), ids)
    loss_o.backward()
    assert abs(out["loss"].item() - loss_o.item()) < 5e-3
    errs = {}
    for key, fname in tl.fairseq_names.items():
        ref, got = W[fname].grad, tl.p[key].grad
        assert got is not None and ref is not None, fname
        if float(ref.norm()) > 1e-8:
            errs[fname] = rel_l2(got, ref)
    bad = {k: v for k, v in errs.items() if v > 8e-2}
    assert not bad, bad
    print("pre-LN unfrozen layer below a frozen one: max rel-L2 %.3g" % max(errs.values()))


def test_edge_batches_single_utterance_and_very_short_utterance(setup):
    """Edge cases of the reference's inputs: a batch of one utterance, and a ragged batch whose shortest utterance (400 samples,
    one conv frame, feat_len = round(400 / 320) = 1) sits next to full-length ones.  Encoder states of every valid frame and
    the pooled embedding against the oracle."""
    model, sd, o_arch, head_W, oracle = setup
    g = torch.Generator().manual_seed(40)
    for lens in ([7000], [12000, 400, 3999, 12000]):
        wavs = [torch.randn(l, generator=g) * 0.5 for l in lens]
        with torch.no_grad():
            out = model.encode_speech(wavs)["parallel_audio_feat"].float().cpu()
            _, fl_m, hs_m = model.forward_audio(*model.processWavs(wavs), return_hidden_states=True)
        hs_o, fl = oracle.speech_encoder_forward(sd, o_arch, wavs)
        assert fl_m.cpu().tolist() == fl.tolist()
        for b, n in enumerate(fl.tolist()):
            assert rel_l2(hs_m[-1][b, :n], hs_o[-1][b, :n]) < 2e-2, (lens, b)
        ws_w = model.audio_encoder.weightedsum_layer.weights.detach().cpu()
        feat = oracle.weighted_sum(ws_w, list(hs_o), False)
        e = oracle.parallel_branch_forward(head_W, feat, fl, nhead=8)
        assert float(F.cosine_similarity(out, e, dim=-1).min()) > 0.999, lens


def _encoder_mask_hook(seed_of, B, T, D, H, R, p=0.1, seg=False):
    """oracle ``drop`` hook that rebuilds the kernels' stateless hash masks on the host (site numbering of
    speech_encoder._encode_kernels: 0 input, 1 encoder, 3 i + 2 attention, 3 i + 3 out_proj, 3 i + 4 fc2 of layer i).
    ``seg``: the frozen encoder's segment layout with every utterance at pitch R (an un-ragged forward): the attention kernel then
    numbers a probability ((h rows + row0[b] + q) max_pitch + k), sc_attn_fwd_seg_bf16."""
    from test_gpu_kernels import _keep_mask, _keep_mask8
    site_of = {"input": lambda i: 0, "encoder": lambda i: 1, "attn": lambda i: 3 * i + 2, "dropout1": lambda i: 3 * i + 3,
               "dropout3": lambda i: 3 * i + 4}
    b_, t_, d_ = np.meshgrid(np.arange(B), np.arange(T), np.arange(D), indexing="ij")
    row_idx = ((b_ * R + t_) * D + d_).astype(np.int64)
    bb, hh, qq, kk = np.meshgrid(np.arange(B), np.arange(H), np.arange(T), np.arange(T), indexing="ij")
    att_idx = (((bb * H + hh) * R + qq) * R + kk).astype(np.int64)
    if seg:
        att_idx = (((hh * (B * R) + bb * R + qq)) * R + kk).astype(np.int64)

    def drop(site, layer, t):
        if site == "attn":               # probabilities: one hash word per four keys, 8-bit fields, applied rate round(256 p) / 256
            keep, p_att = _keep_mask8(att_idx, seed_of(site_of[site](layer)), p)
            return t * torch.from_numpy(keep).float() / (1.0 - p_att)
        keep = _keep_mask(row_idx, seed_of(site_of[site](layer)), p)
        return t * torch.from_numpy(keep).float() / (1.0 - p)

    return drop


def test_encoder_train_mode_dropout_vs_oracle(setup):
    """The reference's TRAINING step runs the frozen HuBERT in train mode (oracle/hubert_ref.py hubert_forward): dropout_input,
    the dropout after the encoder LayerNorm and, per layer, attention / out_proj / fc2 dropout are live (p = 0.1, base).  The
    kernels' masks are stateless hashes of (element, seed) (csrc/sc_common.h), so the host rebuilds every mask from the seeds of
    the call and feeds them to the oracle: all 13 hidden states must agree as in eval mode."""
    model, sd, o_arch, head_W, oracle = setup
    enc = model.audio_encoder
    g = torch.Generator().manual_seed(12)
    lens = [16000, 9000, 3300]
    wavs = [torch.randn(l, generator=g) for l in lens]
    with torch.no_grad():
        hs_eval = [h.clone() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
    try:
        enc.train()
        torch.manual_seed(5)
        enc._drop_calls = 0
        with torch.no_grad():
            hs = [h.clone() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
            hs_again = [h.clone() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
        enc._drop_calls = 0
        seed_of = enc._dropout_seeds()                     # the site -> seed map of call 1
        enc._drop_calls = 0
        with torch.no_grad():
            hs_same = [h.clone() for h in enc([w.cuda() for w in wavs], return_hidden_states=True)[2]]
    finally:
        enc.eval()
    B, T, D = hs[0].shape
    H, R = 12, (T + 1 + 7) // 8 * 8             # returned hidden states: the un-ragged segment layout, pitch roundup(T + 1, 8)
    a = enc.arch
    assert (a.dropout, a.attention_dropout, a.dropout_input) == (0.1, 0.1, 0.1)
    drop = _encoder_mask_hook(seed_of, B, T, D, H, R, seg=True)
    hs_o, _ = oracle.speech_encoder_forward(sd, o_arch, wavs, drop=drop)
    valid = oracle.fairseq_valid_frames(lens, max(lens), T)
    zero_frac = float((hs[0][0, : valid[0]] == 0).float().mean())
    assert 0.08 < zero_frac < 0.12, zero_frac              # F.dropout after the encoder LayerNorm leaves exact zeros in layer_results[0]
    for n in range(13):
        e_valid = max(rel_l2(hs[n][b, :v], hs_o[n][b, :v]) for b, v in enumerate(valid))
        assert e_valid < 2.5e-2, (n, e_valid)
        assert rel_l2(hs[n], hs_eval[n]) > 0.05, n         # and differs from the eval-mode states
        assert torch.equal(hs[n], hs_same[n])              # same seed, same call index -> same masks
        assert not torch.equal(hs[n], hs_again[n])         # next call: new masks
    with torch.no_grad():                                  # eval mode afterwards: deterministic again
        hs_e2 = enc([w.cuda() for w in wavs], return_hidden_states=True)[2]
    assert all(torch.equal(x, y) for x, y in zip(hs_e2, hs_eval))


def test_speech_encoder_api_like_the_reference_test():
    """The call patterns and properties of the reference's own encoder test (test/test_speech_encoder.py:15-45, try_model):
    list-of-waveforms input with descending lengths, feat_select_idx "all" / "hidden_states" / index list, out_dim,
    downsample_rate == 320, frame count within 2 of max(feat_len)."""
    import dataclasses
    from speechclip_plus_amd import random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS, FairseqSpeechEncoder_Hubert
    arch = dataclasses.replace(ARCHS["hubert"], layers=3)
    model = FairseqSpeechEncoder_Hubert(name="hubert", pretrained=False, trainable=False, device="cuda:0", feat_select_idx="all",
                                        state_dict=random_hubert_state_dict(arch, seed=1), arch=arch).eval()
    with torch.no_grad():
        wav_len = [80000 - 97 * i for i in range(8)]
        wav = [torch.randn(l, dtype=torch.float).cuda() for l in wav_len]
        feat_all, feat_len = model(wav, wav_len)
        feat_hid, _ = model(wav, wav_len, "hidden_states")
        feat_hid = tuple(h.clone() for h in feat_hid)                  # the encoder's states live in a reused workspace
        max_hidden = len(feat_hid)
        feat_layers_2, _ = model(wav, wav_len, [2, max_hidden - 1])
        assert isinstance(feat_all, dict)
        assert isinstance(feat_hid, (tuple, list)) and isinstance(feat_hid[0], torch.Tensor)
        assert feat_hid[0].shape[0] == 8 and feat_hid[0].shape[-1] == model.out_dim
        assert (feat_layers_2[0].float() - feat_hid[2].float()).abs().mean() < 1e-5
        assert (feat_layers_2[1].float() - feat_hid[max_hidden - 1].float()).abs().mean() < 1e-5
        assert model.downsample_rate == 320
        for h in range(len(feat_hid)):
            assert abs(feat_hid[h].shape[1] - feat_len.max().item()) <= 2


def test_long_utterance_geometry_vs_oracle():
    """15 s next to 6.25 s (T = 749 -> R = 768, six 128-row attention blocks, conv rows R_l = 768 * 2^(6-l)): hidden states vs the
    oracle on a 3-layer encoder; exercises a padded-row geometry other than the benchmark's R = 512."""
    import dataclasses
    import oracle
    from speechclip_plus_amd import random_hubert_state_dict
    from speechclip_plus_amd.speech_encoder import ARCHS, FairseqSpeechEncoder_Hubert
    arch = dataclasses.replace(ARCHS["hubert"], layers=3)
    sd = random_hubert_state_dict(arch, seed=3)
    enc = FairseqSpeechEncoder_Hubert(name="hubert", device="cuda:0", feat_select_idx="all", state_dict=sd, arch=arch).eval()
    g = torch.Generator().manual_seed(2)
    lens = [240000, 100000]
    wavs = [torch.randn(l, generator=g) * 0.3 for l in lens]
    with torch.no_grad():
        feat, feat_len = enc([w.cuda() for w in wavs])
        hs = [h.clone() for h in feat["hidden_states"]]
    o_arch = oracle.HubertArch.base()
    o_arch.layers = 3
    hs_o, fl_o = oracle.speech_encoder_forward(sd, o_arch, wavs)
    T = hs_o[0].shape[1]
    assert T == 749 and hs[0].shape == (2, T, 768) and feat_len.cpu().tolist() == fl_o.tolist()
    valid = oracle.fairseq_valid_frames(lens, max(lens), T)
    for n in range(4):
        for b, v in enumerate(valid):
            assert rel_l2(hs[n][b, :v], hs_o[n][b, :v]) < 2e-2, (n, b)


@pytest.mark.parametrize("which", ["base", "large"])
def test_fp32_debug_mode_matches_the_oracle(which):
    """SURVEY 8d's "fp32 kernel mode (for debugging) <= 1e-4 rel": the encoder run with every tensor in fp32 on the library's exact-fp32
    kernels (speechclip_plus_amd/debug_fp32.py - same algorithm, masks and row layouts as the production path) must reproduce the fp32
    oracle's hidden states to ~1e-5, ragged batch, padded frames and key masks included.  What the bf16 production path differs by
    beyond that is storage precision (cf. tests/test_gpu_recall.py's bf16-storage-emulated oracle)."""
    import dataclasses
    import oracle
    from speechclip_plus_amd import random_hubert_state_dict
    from speechclip_plus_amd.debug_fp32 import hubert_hidden_states_fp32
    from speechclip_plus_amd.speech_encoder import ARCHS
    if which == "base":
        arch = dataclasses.replace(ARCHS["hubert"], layers=4)
        o_arch = oracle.HubertArch.base()
        o_arch.layers = 4
    else:
        arch = dataclasses.replace(ARCHS["hubert_large_ll60k"], layers=3)
        o_arch = oracle.HubertArch.large()
        o_arch.layers = 3
    sd = random_hubert_state_dict(arch, seed=11)
    g = torch.Generator().manual_seed(5)
    lens = [16000, 9000, 12345]
    wavs = [torch.randn(l, generator=g) * 0.3 + 0.05 for l in lens]
    with torch.no_grad():
        hs, fl = hubert_hidden_states_fp32(sd, arch, wavs, device="cuda:0")
        hs_o, fl_o = oracle.speech_encoder_forward(sd, o_arch, wavs)
    assert fl == fl_o.tolist()
    assert len(hs) == len(hs_o) == arch.layers + 1
    worst = 0.0
    for n in range(len(hs)):
        for b, f in enumerate(fl):                      # valid frames (the oracle's padded frames hold unmasked values too: compare all T)
            e = rel_l2(hs[n][b], hs_o[n][b])
            worst = max(worst, e)
    print(which, "fp32 debug mode: worst hidden-state rel-L2 vs the fp32 oracle", worst)
    assert worst < 1e-4, worst
