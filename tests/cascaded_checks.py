"""Helpers of tests/test_gpu_cascaded.py (not collected on their own): the product modules of scope row a11 (cascaded+/hybrid+
tail: CIF, vector quantiser, keyword BatchNorm, the two plus-branches) against the golden vectors produced by the reference's own
leaf files (cif.py, my_vector_quantizer.py, kw_bn.py; tests/golden/make_golden.py) and against the oracle.  The product modules
run on the HIP kernels only, so every check here needs the GPU; the same fixtures pin the ORACLE on the CPU in
tests/test_oracle_golden.py."""
import numpy as np
import pytest
import torch

from conftest import weights_from

T = torch.from_numpy


def _cif(fx, dev="cuda"):
    from speechclip_plus_amd.cif import CIF
    m = CIF(cif_threshold=1.0, cif_output_dim=32, encoder_embed_dim=32, produce_weight_type="conv", num_layer=1,
            conv_cif_width=3, apply_scaling=True, apply_tail_handling=True, tail_handling_firing_threshold=0.5,
            scaling_step=5000).eval()
    m.load_state_dict(weights_from(fx), strict=True)
    return m.to(dev)


def check_cif(golden, dev="cuda"):
    fx = golden("cif_d32.npz")
    m = _cif(fx, dev)
    feat = T(fx["feat"]).to(dev).requires_grad_(True)
    lens = T(fx["lens"]).to(dev)
    pad = torch.arange(feat.shape[1], device=dev).unsqueeze(0) >= lens.unsqueeze(1)
    tgt = T(fx["tr_target"]).to(dev)
    r = m({"audio_feat": feat, "audio_feat_pad_mask": pad, "global_step": 0}, tgt)
    assert r["dsample_feats_length"].cpu().tolist() == fx["tr_len"].tolist()
    np.testing.assert_allclose(r["dsample_feats"].detach().cpu().numpy(), fx["tr_feats"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["quantity_out"].detach().cpu().numpy(), fx["tr_quantity"], rtol=1e-5)
    np.testing.assert_allclose(r["alpha"].detach().cpu().numpy(), fx["tr_alpha"], rtol=1e-4, atol=1e-6)
    (r["dsample_feats"] * T(fx["tr_gout"]).to(dev)).sum().backward()
    np.testing.assert_allclose(feat.grad.cpu().numpy(), fx["tr_gfeat"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(m.conv[0].weight.grad.cpu().numpy(), fx["tr_gconvw"], rtol=1e-3, atol=1e-5)
    with torch.no_grad():
        r = m({"audio_feat": feat.detach(), "audio_feat_pad_mask": pad, "global_step": 0}, None)
    assert r["dsample_feats_length"].cpu().tolist() == fx["ev_len"].tolist()
    np.testing.assert_allclose(r["dsample_feats"].cpu().numpy(), fx["ev_feats"], rtol=1e-4, atol=1e-5)
    assert (r["dsample_feats_pad_mask"].cpu().numpy() == fx["ev_pad"]).all()
    assert (r["fired_marks"].cpu().numpy() == fx["ev_fired"]).all()
    with torch.no_grad():
        r = m({"audio_feat": feat.detach(), "audio_feat_pad_mask": pad, "global_step": 6000}, tgt)
    assert not m.apply_scaling
    assert r["dsample_feats_length"].cpu().tolist() == fx["ns_len"].tolist()
    np.testing.assert_allclose(r["dsample_feats"].cpu().numpy(), fx["ns_feats"], rtol=1e-4, atol=1e-5)


def check_vq(golden, dev="cuda"):
    from speechclip_plus_amd.vector_quantizers import SimpleVectorQuantizer
    fx = golden("vq_v50.npz")
    vq = SimpleVectorQuantizer(temp="fixed=0.1", time_first=True, use_gumbel=False, hard=True).to(dev)
    vq.eval()
    re = vq(x=T(fx["x"]).clone().to(dev))
    assert torch.equal(re["subword_prob"].cpu(), T(fx["ev_prob"]))
    assert torch.equal(re["targets"].cpu(), T(fx["ev_targets"]))
    for k, g in [("code_perplexity", "ev_code_ppl"), ("prob_perplexity", "ev_prob_ppl"), ("ent_per_t", "ev_ent"),
                 ("diversity_loss", "ev_div")]:
        np.testing.assert_allclose(re[k].cpu().numpy(), fx[g], rtol=1e-4, atol=1e-6)
    vq.train()
    x = T(fx["x"]).clone().to(dev).requires_grad_(True)
    rt = vq(x=x * 1.0)
    np.testing.assert_allclose(rt["subword_prob"].detach().cpu().numpy(), fx["tr_prob"], rtol=1e-5, atol=1e-6)
    ((rt["subword_prob"] @ T(fx["emb"]).to(dev)) * T(fx["gk"]).to(dev)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), fx["tr_gx"], rtol=1e-3, atol=1e-6)


def check_bn(golden, dev="cuda"):
    from speechclip_plus_amd.vector_quantizers import Kw_BatchNorm_dynamic
    fx = golden("kwbn_e16.npz")
    bn = Kw_BatchNorm_dynamic(kw_dim=16, init_bias=T(fx["init_bias"]), init_scale=T(fx["init_weight"]), std_scale=1.0).to(dev)
    kw = T(fx["kw"]).to(dev)
    bn.train()
    np.testing.assert_allclose(bn(kw).detach().cpu().numpy(), fx["y_train"], rtol=1e-4, atol=1e-5)
    bn.eval()
    np.testing.assert_allclose(bn(kw).detach().cpu().numpy(), fx["y_eval"], rtol=1e-4, atol=1e-5)
    sd = {k: v.cpu() for k, v in bn.state_dict().items()}
    for k, v in weights_from(fx).items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def check_clip_text_encode_keywords_matches_loop(dev="cuda"):
    """encode_keywords: the vectorised keyword splice equals the reference's per-row loop semantics
    (clip_official.py:261-263) and the int keyword_num variant; causal mask: EOT row ignores later positions."""
    from speechclip_plus_amd.clip_text import ClipModel, CONTEXT_LEN
    ids = torch.cat([torch.arange(0, 40), torch.tensor([49406, 49407])])
    clip = ClipModel("ViT-B/32", device=dev, reduce_subword_embbedding=ids, layers=2, seed=7).eval()
    W = clip.model.token_embedding.weight.shape[1]
    g = torch.Generator().manual_seed(0)
    kws = (torch.randn(3, 6, W, generator=g) * 0.02).to(dev)
    n = torch.tensor([6, 2, 4]).to(dev)
    out = clip.encode_keywords(kws, n)
    assert out.shape == (3, 512)
    for b in range(3):
        one = clip.encode_keywords(kws[b: b + 1, : int(n[b])], int(n[b]))
        np.testing.assert_allclose(out[b].detach().cpu().numpy(), one[0].detach().cpu().numpy(), rtol=2e-2, atol=2e-3)
    kws2 = kws.clone()
    kws2[1, 2:] += 5.0                       # beyond utterance 1's 2 keywords: must not matter
    np.testing.assert_allclose(clip.encode_keywords(kws2, n)[1].detach().cpu().numpy(), out[1].detach().cpu().numpy(), rtol=1e-5,
                               atol=1e-6)
    assert all(not p.requires_grad for p in clip.parameters())
    kws.requires_grad_(True)
    clip.encode_keywords(kws, n).sum().backward()         # gradient flows THROUGH the frozen tower
    assert float(kws.grad[0].abs().sum()) > 0 and float(kws.grad[1, 2:].abs().sum()) == 0


def _small_cfg(kind, D=64, E=48):
    from speechclip_plus_amd import Config
    cif = {"quantity_loss_weight": 0.25, "using_gt_len": False, "cif_output_dim": D, "encoder_embed_dim": D,
           "produce_weight_type": "conv", "cif_threshold": 1.0, "conv_cif_width": 3, "apply_scaling": True,
           "scaling_step": 5000, "apply_tail_handling": True, "tail_handling_firing_threshold": 0.5}
    kw = {"batchnorms": {"type": "eachKw", "std_scale": 1.0, "learnable": True}}
    if kind == "hybrid":
        kw["kw_projection"] = {"dropout": 0.0, "dimensions": [D, D, 512]}
    return Config({"model_settings": {"cascaded_branch": {
        "type": "HybridBranch_dynamic" if kind == "hybrid" else "CascadedBranch_dynamic",
        "vq": {"type": "SimpleVectorQuantizer", "args": {"temp": "fixed=0.1", "time_first": True, "use_gumbel": False, "hard": True}},
        "downsampling": {"type": "cif", "cif": cif}, "keyword": kw,
        "transformer_args": {"type": "MultiheadAttentionAndNorm", "n_layers": 1, "d_model": D, "nhead": 8 if kind == "hybrid" else 1,
                             "dim_feedforward": 128, "dropout": 0.0, "activation": "gelu", "layer_norm_eps": 1e-5,
                             "batch_first": True, "norm_first": False}}}})


def build_branch(kind, dev="cuda"):
    from speechclip_plus_amd import KW_CascadedBranchPlus, KW_HybridBranchPlus
    from speechclip_plus_amd.clip_text import ClipModel
    torch.manual_seed(11 if kind == "hybrid" else 12)
    ids = torch.cat([torch.arange(0, 200), torch.tensor([49406, 49407])])
    clip = ClipModel("ViT-B/32", device=dev, reduce_subword_embbedding=ids, layers=2, seed=3)
    cfg = _small_cfg(kind)
    if kind == "hybrid":
        br = KW_HybridBranchPlus(cfg, audio_dim=64, text_dim=512, out_dim=512, clip=clip)
    else:
        br = KW_CascadedBranchPlus(cfg, audio_dim=64, text_dim=512, clip=clip)
    for m in br.modules():                      # deterministic training path: every dropout off
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with torch.no_grad():
        br.downsampling.weight_proj[1].bias.add_(-0.3)
        br.linear_proj.apply(lambda m: m.weight.mul_(30.0) if isinstance(m, torch.nn.Linear) else None)
    return br.to(dev), clip


def check_branch_vs_oracle(kind, dev="cuda"):
    import oracle
    br, clip = build_branch(kind, dev)
    W = {k: v.detach().cpu().float() for k, v in br.state_dict().items()}
    sot, eot = clip.startOfTxt_reduced, clip.endOfTxt_reduced
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(4, 50, 64, generator=g)
    lens = torch.tensor([50, 33, 20, 41])
    nhead = 8 if kind == "hybrid" else 1
    fn = oracle.hybrid_plus_forward if kind == "hybrid" else oracle.cascaded_plus_forward
    for training in (False, True):
        br.train(training)
        tgt = (lens / 20).round().long()
        out = br(audio_feat=feat.to(dev), audio_feat_len=lens.to(dev),
                 otherInputs={"global_step": 0, "target_len": tgt.to(dev)} if training else {})
        ref = fn(W, feat, lens, nhead, training=training, target_len=tgt if training else None, nhead_clip=8, sot=sot, eot=eot)
        def close(got, want, rtol, atol):
            got, want = got.detach().cpu().float(), want.detach().float()
            if dev == "cpu":
                np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=rtol, atol=atol)
            else:       # bf16 projections / text tower on the GPU: rel-L2 <= 2e-2 (SURVEY 8d tolerances)
                assert float((got - want).norm() / want.norm()) < 2e-2

        if kind == "hybrid":
            close(out["parallel_audio_feat"], ref[0], 2e-3, 2e-4)
            ref = ref[1:]
        assert out["dsample_results"]["dsample_feats_length"].cpu().tolist() == ref[2].tolist()
        np.testing.assert_allclose(out["dsample_results"]["quantity_out"].detach().cpu().numpy(), ref[3].numpy(), rtol=1e-3)
        close(out["keywords"], ref[1], 2e-3, 2e-4)
        got, want = out["cascaded_audio_feat"].detach().cpu().float(), ref[0].detach().float()
        if dev == "cpu":
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=5e-3, atol=5e-4)
        else:
            # on the GPU the frozen text tower runs on the bf16 kernels (clip_text_hip): bf16 storage + fp32 accumulate through
            # every block -> rel-L2 <= 2e-2, cosine >= 0.999 (the tolerances of the HuBERT hidden states)
            assert float((got - want).norm() / want.norm()) < 2e-2
            assert float(torch.nn.functional.cosine_similarity(got, want, dim=-1).min()) > 0.999
