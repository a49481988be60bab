"""CPU: the oracle (torch fp32 restatement) against the golden vectors produced by the reference's own
leaf files (tests/golden/make_golden.py) and against the independent HF HuBERT implementation."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle.lengths import conv_out_lengths, fairseq_valid_frames, feat_len_rule, get_keypadding_mask
from conftest import ROOT, weights_from

T = torch.from_numpy


@pytest.mark.parametrize("name", ["loss_b8", "loss_b32_dup", "loss_b32_dup_trainT", "loss_b256_dup", "loss_b512_cap_lifted"])
def test_loss(golden, name):
    fx = golden(name + ".npz")
    A = T(fx["A"]).requires_grad_(True)
    B = T(fx["B"]).requires_grad_(True)
    if "temp_param" in fx:
        p = T(fx["temp_param"]).clone().requires_grad_(True)
        inv_t = p.exp()
    else:
        p, inv_t = None, 1 / 0.07
    loss = oracle.masked_contrastive_loss(A, B, T(fx["ids"]), inv_temperature=inv_t)
    loss.backward()
    assert abs(loss.item() - fx["loss"].item()) < 1e-5
    np.testing.assert_allclose(A.grad.numpy(), fx["dA"], rtol=1e-4, atol=1e-6)
    if "dB" in fx:
        np.testing.assert_allclose(B.grad.numpy(), fx["dB"], rtol=1e-4, atol=1e-6)
    if p is not None:
        np.testing.assert_allclose(p.grad.numpy(), fx["dtemp_param"], rtol=1e-4)
    if "loss_noindex" in fx:
        l2 = oracle.masked_contrastive_loss(A.detach(), B.detach(), None, inv_temperature=float(inv_t))
        assert abs(l2.item() - fx["loss_noindex"].item()) < 1e-5


@pytest.mark.parametrize("name", ["head_d64_h8", "head_d64_h1"])
def test_parallel_head(golden, name):
    fx = golden(name + ".npz")
    W = {k: v.clone().requires_grad_(True) for k, v in weights_from(fx).items()}
    feat = T(fx["feat"]).requires_grad_(True)
    out = oracle.parallel_branch_forward(W, feat, T(fx["audio_len"]), nhead=int(fx["nhead"]))
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], rtol=1e-4, atol=2e-5)
    (out * T(fx["gout"])).sum().backward()
    np.testing.assert_allclose(feat.grad.numpy(), fx["g_feat"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(W["cls"].grad.numpy(), fx["g_cls"], rtol=1e-3, atol=2e-5)
    for k in W:
        if k == "cls":
            continue
        g = fx["g_" + k] if "g_" + k in fx else None
        if g is not None:
            np.testing.assert_allclose(W[k].grad.numpy(), g, rtol=1e-3, atol=3e-5, err_msg=k)
    m = get_keypadding_mask(feat.shape[1] + 1, T(fx["audio_len"]) + 1)
    assert (m.numpy() == fx["kpm"]).all()


def test_mha_and_norm(golden):
    fx = golden("mha_norm_d64_h8.npz")
    W = weights_from(fx)
    kpm = get_keypadding_mask(fx["src"].shape[1], T(fx["lens"]))
    out = oracle.mha_and_norm_forward(W, "", T(fx["src"]), kpm, int(fx["nhead"]))
    np.testing.assert_allclose(out.numpy(), fx["out"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["mha_norm_d128_h2", "mha_norm_d192_h2"])
def test_mha_and_norm_with_gradients(golden, name):
    """head_dim 64 and 96 (the hybrid+ base geometry): output, input gradient and every parameter gradient."""
    fx = golden(name + ".npz")
    W = {k: v.clone().requires_grad_(True) for k, v in weights_from(fx).items()}
    src = T(fx["src"]).requires_grad_(True)
    kpm = get_keypadding_mask(src.shape[1], T(fx["lens"]))
    out = oracle.mha_and_norm_forward(W, "", src, kpm, int(fx["nhead"]))
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], rtol=1e-4, atol=2e-5)
    (out * T(fx["gout"])).sum().backward()
    np.testing.assert_allclose(src.grad.numpy(), fx["g_src"], rtol=1e-3, atol=3e-5)
    for k in W:
        np.testing.assert_allclose(W[k].grad.numpy(), fx["g_" + k], rtol=1e-3, atol=5e-5, err_msg=k)


@pytest.mark.parametrize("name", ["loss_v_margin", "loss_v_dcl", "loss_v_a2b", "loss_v_b2a", "loss_v_margin_dcl_trainT"])
def test_loss_variants(golden, name):
    """margin, decoupled (dcl) and one-sided forms of MaskedContrastiveLoss (losses.py:213,226-245)."""
    fx = golden(name + ".npz")
    A = T(fx["A"]).requires_grad_(True)
    B = T(fx["B"]).requires_grad_(True)
    kw = dict(margin=float(fx["margin"]), dcl=bool(fx["dcl"]), a2b=bool(fx["a2b"]), b2a=bool(fx["b2a"]))
    p = None
    inv_t = 1 / 0.07
    if int(fx["trainT"]):
        p = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
        inv_t = p.exp()
    loss = oracle.masked_contrastive_loss(A, B, T(fx["ids"]), inv_temperature=inv_t, **kw)
    loss.backward()
    assert abs(loss.item() - fx["loss"].item()) < 1e-5
    np.testing.assert_allclose(A.grad.numpy(), fx["dA"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(B.grad.numpy(), fx["dB"], rtol=1e-4, atol=1e-6)
    if p is not None:
        np.testing.assert_allclose(p.grad.numpy(), fx["dtemp_param"], rtol=1e-4)
    l2 = oracle.masked_contrastive_loss(A.detach(), B.detach(), None, inv_temperature=float(inv_t), **kw)
    assert abs(l2.item() - fx["loss_noindex"].item()) < 1e-5


def test_cif_oracle_vs_reference_fixture(golden):
    """oracle.cif_forward against the reference's cif.py vectors: training (target scaling, tail dropped), inference (tail
    firing) and the post-scaling-step path; gradient to the features."""
    fx = golden("cif_d32.npz")
    W = {"downsampling." + k: v for k, v in weights_from(fx).items()}
    feat = T(fx["feat"]).requires_grad_(True)
    lens = T(fx["lens"])
    pad = get_keypadding_mask(feat.shape[1], lens)
    tgt = T(fx["tr_target"])
    out, n, q = oracle.cif_forward(W, "downsampling.", feat, pad, tgt)
    assert n.tolist() == fx["tr_len"].tolist()
    np.testing.assert_allclose(out.detach().numpy(), fx["tr_feats"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(q.detach().numpy(), fx["tr_quantity"], rtol=1e-5)
    (out * T(fx["tr_gout"])).sum().backward()
    np.testing.assert_allclose(feat.grad.numpy(), fx["tr_gfeat"], rtol=1e-3, atol=1e-5)
    with torch.no_grad():
        out, n, q = oracle.cif_forward(W, "downsampling.", feat.detach(), pad, None)
        assert n.tolist() == fx["ev_len"].tolist()
        np.testing.assert_allclose(out.numpy(), fx["ev_feats"], rtol=1e-4, atol=1e-5)
        out, n, _ = oracle.cif_forward(W, "downsampling.", feat.detach(), pad, tgt, apply_scaling=False)
        assert n.tolist() == fx["ns_len"].tolist()
        np.testing.assert_allclose(out.numpy(), fx["ns_feats"], rtol=1e-4, atol=1e-5)


def test_vq_oracle_vs_reference_fixture(golden):
    fx = golden("vq_v50.npz")
    x = T(fx["x"])
    assert torch.equal(oracle.vq_forward(x, 0.1, training=False), T(fx["ev_prob"]))
    xt = x.clone().requires_grad_(True)
    prob = oracle.vq_forward(xt * 1.0, 0.1, training=True)
    np.testing.assert_allclose(prob.detach().numpy(), fx["tr_prob"], rtol=1e-5, atol=1e-6)
    ((prob @ T(fx["emb"])) * T(fx["gk"])).sum().backward()
    np.testing.assert_allclose(xt.grad.numpy(), fx["tr_gx"], rtol=1e-3, atol=1e-6)


def test_weighted_sum(golden):
    fx = golden("wsum.npz")
    w = T(fx["weights"]).requires_grad_(True)
    hs = list(T(fx["hs"]))
    out = oracle.weighted_sum(w, hs)
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], rtol=1e-5, atol=1e-6)
    (out * T(fx["gout"])).sum().backward()
    np.testing.assert_allclose(w.grad.numpy(), fx["dweights"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(oracle.weighted_sum(w.detach(), hs, True).numpy(), fx["out_norm"], rtol=1e-4, atol=1e-5)


def test_masks_and_lengths(golden):
    fx = golden("masks.npz")
    assert (get_keypadding_mask(17, T(fx["kpm_lens"])).numpy() == fx["kpm"]).all()
    for L, t in zip(fx["Ls"], fx["T"]):
        assert conv_out_lengths(int(L))[-1] == int(t)
    assert conv_out_lengths(160000)[-1] == 499 and conv_out_lengths(102400)[-1] == 319
    for i, Lmax in enumerate(fx["fm_Lmax"]):
        wl, valid = fx[f"fm_L{i}"]
        Tn = conv_out_lengths(int(Lmax))[-1]
        assert fairseq_valid_frames(wl.tolist(), int(Lmax), Tn) == valid.tolist()
    assert feat_len_rule(fx["round_in"].tolist(), 10**9) == fx["round_out"].tolist()
    # half-to-even cases the survey calls out: 160/320 = .5 -> 0, 480/320 = 1.5 -> 2, 800/320 = 2.5 -> 2
    assert feat_len_rule([160, 480, 800], 499) == [0, 2, 2]


def test_retrieval(golden):
    fx = golden("retrieval.npz")
    s = T(fx["score"])
    AB, BA, mean = oracle.mutual_retrieval(s, s.T, T(fx["a_ids"]), T(fx["b_ids"]), [1, 5, 10])
    for i, k in enumerate([1, 5, 10]):
        assert abs(AB[f"recall@{k}"] - fx["AB"][i]) < 1e-4
        assert abs(BA[f"recall@{k}"] - fx["BA"][i]) < 1e-4
        assert abs(mean[f"recall@{k}"] - fx["mean"][i]) < 1e-4


@pytest.mark.parametrize("name", ["hubert_small", "hubert_small_preln"])
def test_hubert_vs_hf(golden, name):
    fx = golden(name + ".npz")
    W = weights_from(fx)
    stable = bool(fx["stable"])
    arch = oracle.HubertArch(embed_dim=32, ffn_dim=64, layers=2, heads=4, conv_dim=16,
                             extractor_mode="layer_norm" if stable else "default", conv_bias=stable,
                             layer_norm_first=stable)
    wav = T(fx["wav"])
    hs = oracle.hubert_forward(W, arch, wav, None)
    n = len(hs) - 1 if stable else len(hs)      # HF applies encoder.layer_norm to its last pre-LN state
    for i in range(n):
        np.testing.assert_allclose(hs[i].numpy(), fx["hf_hidden"][i], rtol=1e-3, atol=2e-5)
    # padded batch: the two mask rules agree on these lengths -> valid frames must match HF
    lens = fx["lens"].tolist()
    L = wav.shape[1]
    wavs = [wav[b, :l] for b, l in enumerate(lens)]
    hs_p, feat_len = oracle.speech_encoder_forward(W, arch, wavs)
    Tn = hs_p[0].shape[1]
    valid = fairseq_valid_frames(lens, L, Tn)
    assert valid == fx["hf_valid"].tolist()
    for i in range(n):
        for b, v in enumerate(valid):
            np.testing.assert_allclose(hs_p[i][b, :v].numpy(), fx["hf_hidden_padded"][i][b, :v], rtol=1e-3, atol=5e-5)
    assert feat_len.tolist() == feat_len_rule(lens, Tn)


@pytest.mark.parametrize("name", ["clip_text_w64", "clip_text_w128"])
def test_clip_text_tower_vs_hf(golden, name):
    """oracle.clip_text_transformer / clip_encode_keywords (the restatement of openai/CLIP's text tower as clip_official.py:222-279
    drives it) against transformers' independent CLIPTextModelWithProjection (tests/golden/make_golden.py make_clip_text): plain
    token ids, spliced continuous keyword vectors, and the gradient that flows back through the frozen tower."""
    fx = golden(name + ".npz")
    W = weights_from(fx)
    heads, sot, eot = int(fx["heads"]), int(fx["sot"]), int(fx["eot"])
    n_kw = T(fx["n_kw"])
    emb = W["clip.model.token_embedding.weight"]
    out = oracle.clip_encode_keywords(W, "clip.model.", emb[T(fx["tok"])], n_kw, heads, sot, eot)
    np.testing.assert_allclose(out.numpy(), fx["out_ids"], rtol=1e-4, atol=2e-5)
    kw = T(fx["kw"]).clone().requires_grad_(True)
    out = oracle.clip_encode_keywords(W, "clip.model.", kw, n_kw, heads, sot, eot)
    np.testing.assert_allclose(out.detach().numpy(), fx["out_kw"], rtol=1e-4, atol=2e-5)
    (out * T(fx["gout"])).sum().backward()
    np.testing.assert_allclose(kw.grad.numpy(), fx["g_kw"], rtol=1e-3, atol=2e-6)
    # the tower alone (before ln_final / pooling), every one of the 77 rows
    x = emb[0].expand(len(n_kw), 77, -1).clone()
    x[:, 0] = emb[sot]
    for b, n in enumerate(n_kw.tolist()):
        x[b, 1: 1 + n] = T(fx["kw"])[b, :n]
        x[b, 1 + n] = emb[eot]
    h = oracle.clip_text_transformer(W, "clip.model.", x + W["clip.model.positional_embedding"], heads)
    np.testing.assert_allclose(h.numpy(), fx["tower_out_kw"], rtol=1e-4, atol=5e-5)


def test_head_train_mode_dropout_sites_vs_torch_layer():
    """Train mode of the oracle head (drop hook) against torch's own nn.TransformerEncoderLayer - the module the reference
    instantiates (avssl/module/kw_modules/TransformerModels.py:62-73) - in train mode, with its three nn.Dropout modules
    replaced by recorded-mask multipliers (the attention-probability site lives inside scaled_dot_product_attention and
    cannot be intercepted: it is set to p = 0 here and covered by the kernel-level mask tests)."""
    from torch import nn
    D, H, Fd, B, S = 32, 4, 48, 3, 9
    g = torch.Generator().manual_seed(3)
    layer = nn.TransformerEncoderLayer(D, H, Fd, dropout=0.1, activation="gelu", batch_first=True, norm_first=False)
    enc = nn.TransformerEncoder(layer, 1, nn.LayerNorm(D), enable_nested_tensor=False).train()
    masks = {}

    class Rec(nn.Module):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def forward(self, x):
            m = (torch.rand(x.shape, generator=g) >= 0.3).float() / 0.7
            masks[self.name] = m
            return x * m

    L = enc.layers[0]
    L.self_attn.dropout = 0.0
    L.dropout1, L.dropout, L.dropout2 = Rec("dropout1"), Rec("dropout"), Rec("dropout2")
    x = torch.randn(B, S, D, generator=g)
    lens = torch.tensor([9, 5, 7])
    kpm = get_keypadding_mask(S, lens)
    ref = enc(x, src_key_padding_mask=kpm)
    W = {"enc.model." + k: v.detach() for k, v in enc.state_dict().items()}
    out = oracle.head_ref.transformer_encoder_forward(
        W, "enc.", x, kpm, 1, H, drop=lambda site, i, t: t if site == "attn" else t * masks[site])
    valid = ~kpm
    np.testing.assert_allclose(out[valid].detach().numpy(), ref[valid].detach().numpy(), rtol=1e-4, atol=2e-5)


def test_recall_fixture_is_reproduced_by_the_oracle_and_resolves_the_target(golden):
    """tests/golden/recall_eval.npz (round 3): (a) the fp32 oracle and its bf16-storage emulation reproduce the stored embeddings
    of the utterances of one protocol batch (the batch that holds utterance 0: same length-sorted composition, BATCH = 40);
    (b) the stored ranks give the recalls of recall_eval_margins.json, and the two references agree on every rank-1 / 5 / 10
    decision; (c) the eval set resolves +-0.1: < 0.2 % of the held-out queries have an oracle margin within 3 sigma of the
    emulation's margin noise at any of the three boundaries."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import recall_eval as rev
    fx = golden("recall_eval.npz")
    summ = json.load(open(os.path.join(ROOT, "tests", "golden", "recall_eval_margins.json")))
    n_ids = int(fx["n_ids"])
    assert int(fx["batch"]) == rev.BATCH == summ["protocol"]["batch"]
    wavs, ids = rev.eval_set(n_ids)
    order = sorted(range(len(wavs)), key=lambda i: len(wavs[i]))
    pos = order.index(0)
    sel = order[pos // rev.BATCH * rev.BATCH: pos // rev.BATCH * rev.BATCH + rev.BATCH]
    Wh, Whead, arch = rev.hubert_weights(), rev.head_weights(), oracle.HubertArch.base()
    members = [i for i in sel if i < 64]
    assert 0 in members
    for name, W, store in (("fp32", Wh, None), ("bf16emu", oracle.bf16_weights(Wh), oracle.bf16_store)):
        with torch.no_grad():
            hs, fl = oracle.speech_encoder_forward(W, arch, [wavs[i] for i in sel], store=store)
            f = oracle.weighted_sum(rev.WS_WEIGHTS, hs)
            e = oracle.parallel_branch_forward(Whead, f if store is None else store(f), fl, nhead=8)
        want = T(fx["emb_head_" + name])
        for i in members:
            got = e[sel.index(i)]
            assert float((got - want[i]).norm() / want[i].norm()) < 2e-4, (name, i)      # thread count / summation order only
    held = (torch.arange(len(ids)) % rev.PER_ID) >= rev.GALLERY
    for name in ("fp32", "bf16emu"):
        r_ai, r_ia = T(fx["rank_ai_" + name]).long(), T(fx["rank_ia_" + name]).long()
        assert rev.recalls(r_ai) == summ[name]["audio_to_image"] and rev.recalls(r_ia) == summ[name]["image_to_audio"]
        assert rev.recalls(r_ai, held) == summ[name]["audio_to_image_heldout"]
    for k in (1, 5, 10):
        assert torch.equal(T(fx["rank_ai_fp32"]).long() < k, T(fx["rank_ai_bf16emu"]).long() < k)
        assert torch.equal(T(fx["rank_ia_fp32"]).long() < k, T(fx["rank_ia_bf16emu"]).long() < k)
    sigma = summ["bf16emu"]["margin_noise_sigma"]
    m = T(fx["margin_ai_fp32"])
    assert float((m[held].abs() < 3 * sigma).float().mean(0).max()) < 0.002
    assert 0 < rev.recalls(T(fx["rank_ai_fp32"]).long())[0] < 100          # neither 0 nor 100
