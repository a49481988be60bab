#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference, which does not travel to the GPU box).
It imports the reference's *leaf* files by path (their packages' __init__ pull in clip /
pytorch_lightning / fairseq, which are absent - SURVEY F5/F6), runs them on CPU on small seeded
inputs and stores inputs + weights + expected outputs.  No reference source is copied: the .npz
files hold data only.

    python tests/golden/make_golden.py

Fixtures:
  loss_*.npz       MaskedContrastiveLoss (avssl/module/losses.py) fwd + grads
  head_*.npz       KW_ParallelBranch computation composed exactly as kw_branches.py:266-280 from
                   TransformerModels.TransformerEncoder ; MultiheadAttentionAndNorm
  wsum.npz         WeightedSumLayer (avssl/module/weighted_sum.py)
  masks.npz        get_keypadding_mask (avssl/util/data_utils.py) + the three length rules
  retrieval.npz    mutualRetrieval (avssl/module/retrieval.py)
  clip_text_w64.npz transformers.CLIPTextModelWithProjection (independent implementation, local config): the
                   cross-check of the openai/CLIP text tower restated in oracle/cascaded_ref.py
  hubert_small.npz transformers.HubertModel (independent implementation, local config) outputs for a
                   tiny HuBERT: the only available cross-check at the fairseq boundary
"""
import importlib.util
import os
import sys
import zlib

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))


def load_leaf(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def unit(x):
    return x / x.norm(dim=-1, keepdim=True)


# --------------------------------------------------------------------------- loss
def make_loss():
    ref = load_leaf("avssl/module/losses.py", "ref_losses")
    cases = [
        ("loss_b8", 8, 16, False, False),
        ("loss_b32_dup", 32, 32, True, False),
        ("loss_b32_dup_trainT", 32, 32, True, True),
        ("loss_b256_dup", 256, 32, True, False),
    ]
    for name, B, E, dup, trainT in cases:
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 2**31)
        A = unit(torch.randn(B, E, generator=g)).requires_grad_(True)
        Bm = unit(torch.randn(B, E, generator=g) + 0.5 * A.detach()).requires_grad_(True)
        ids = torch.arange(B) // 5 if dup else torch.arange(B)
        crit = ref.MaskedContrastiveLoss(temperature=0.07, temperature_trainable=trainT)
        loss = crit(A, Bm, ids)
        loss.backward()
        extra = {}
        if trainT:
            extra["dtemp_param"] = crit.temperature.grad
            extra["temp_param"] = crit.temperature.detach()
        loss_noidx = crit(A.detach(), Bm.detach(), None)
        npz(name + ".npz", A=A, B=Bm, ids=ids, loss=loss, dA=A.grad, dB=Bm.grad, loss_noindex=loss_noidx, **extra)
    # lifted cap (F8): B = 512 has no runnable reference unless MAX_EYE is raised before constructing
    ref.MAX_EYE = 512
    g = torch.Generator().manual_seed(512)
    B, E = 512, 16
    A = unit(torch.randn(B, E, generator=g)).requires_grad_(True)
    Bm = unit(torch.randn(B, E, generator=g) + 0.5 * A.detach())
    ids = torch.arange(B) // 5
    crit = ref.MaskedContrastiveLoss(temperature=0.07)
    loss = crit(A, Bm, ids)
    loss.backward()
    npz("loss_b512_cap_lifted.npz", A=A, B=Bm, ids=ids, loss=loss, dA=A.grad)


# --------------------------------------------------------------------------- head
def make_head():
    TM = load_leaf("avssl/module/kw_modules/TransformerModels.py", "ref_tm")
    du = load_leaf("avssl/util/data_utils.py", "ref_du")
    for name, D, nhead, F_, T, lens in [
        ("head_d64_h8", 64, 8, 128, 16, [16, 0, 7, 11]),
        ("head_d64_h1", 64, 1, 128, 16, [16, 1, 9, 3]),
    ]:
        torch.manual_seed(zlib.crc32(name.encode()) % 2**31)
        enc = TM.TransformerEncoder(n_layers=1, d_model=D, nhead=nhead, dim_feedforward=F_, dropout=0.1,
                                    activation="gelu", layer_norm_eps=1e-5, batch_first=True, norm_first=False)
        # make every affine / bias term non-trivial
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if "norm" in n or "bias" in n:
                    p.add_(0.1 * torch.randn_like(p))
        enc.eval()
        cls = torch.nn.Parameter(torch.randn(1, 1, D))
        proj = torch.nn.Linear(D, 24)
        B = len(lens)
        feat = torch.randn(B, T, D, requires_grad=True)
        audio_len = torch.tensor(lens, dtype=torch.long)
        # kw_branches.py:266-280
        src = torch.cat([torch.cat([cls] * B, dim=0), feat], dim=1)
        kpm = du.get_keypadding_mask(max_length=T + 1, data_lens=audio_len + 1)
        # grad mode on => stock (non-fused) TransformerEncoderLayer path
        out_full = enc(src=src, key_padding_mask=kpm)
        out = proj(out_full[:, :1].reshape(-1, D))
        gout = torch.randn(out.shape)
        (out * gout).sum().backward()
        W = {"cls": cls}
        W.update({"self_att." + k: v for k, v in enc.state_dict().items()})
        W["linear_proj.weight"] = proj.weight
        W["linear_proj.bias"] = proj.bias
        G = {"g_cls": cls.grad, "g_feat": feat.grad, "g_linear_proj.weight": proj.weight.grad}
        for n, p in enc.named_parameters():
            G["g_self_att." + n] = p.grad
        hid = enc.extract_hidden_states(src=src, key_padding_mask=kpm)
        npz(name + ".npz", feat=feat, audio_len=audio_len, kpm=kpm, out=out, out_full=out_full.detach(), gout=gout,
            hidden_last=hid[-1].detach(), nhead=np.int64(nhead),
            **{"W_" + k: v for k, v in W.items()}, **G)
    # MultiheadAttentionAndNorm (cascaded+/hybrid+ block)
    torch.manual_seed(77)
    D, nhead, T = 64, 8, 12
    blk = TM.MultiheadAttentionAndNorm(d_model=D, nhead=nhead, dropout=0.1, layer_norm_eps=1e-5, batch_first=True)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if "Norm" in n or "bias" in n:
                p.add_(0.1 * torch.randn_like(p))
    blk.eval()
    src = torch.randn(3, T, D)
    lens = torch.tensor([12, 5, 1])
    kpm = du.get_keypadding_mask(T, lens)
    with torch.no_grad():
        out = blk(src, kpm)
    npz("mha_norm_d64_h8.npz", src=src, lens=lens, out=out, nhead=np.int64(nhead),
        **{"W_" + k: v for k, v in blk.state_dict().items()})


def make_mha():
    """MultiheadAttentionAndNorm at head_dim 64 (D = 128, 2 heads) and head_dim 96 (D = 192, 2 heads: the hybrid+ base
    geometry 768 / 8 scaled down): output, input gradient and every parameter gradient (eval mode: dropout off)."""
    TM = load_leaf("avssl/module/kw_modules/TransformerModels.py", "ref_tm")
    du = load_leaf("avssl/util/data_utils.py", "ref_du")
    for name, D, nhead, T, lens in [("mha_norm_d128_h2", 128, 2, 20, [20, 9, 1]), ("mha_norm_d192_h2", 192, 2, 20, [20, 13, 4])]:
        torch.manual_seed(zlib.crc32(name.encode()) % 2**31)
        blk = TM.MultiheadAttentionAndNorm(d_model=D, nhead=nhead, dropout=0.1, layer_norm_eps=1e-5, batch_first=True)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                if "Norm" in n or "bias" in n:
                    p.add_(0.1 * torch.randn_like(p))
        blk.eval()
        src = torch.randn(len(lens), T, D, requires_grad=True)
        kpm = du.get_keypadding_mask(T, torch.tensor(lens))
        out = blk(src, kpm)
        gout = torch.randn_like(out)
        (out * gout).sum().backward()
        npz(name + ".npz", src=src, lens=np.array(lens), out=out, gout=gout, g_src=src.grad, nhead=np.int64(nhead),
            **{"W_" + k: v for k, v in blk.state_dict().items()}, **{"g_" + n: p.grad for n, p in blk.named_parameters()})


def make_loss_variants():
    """MaskedContrastiveLoss options beyond the shipped defaults (losses.py:213,226-245): margin, dcl, one-sided."""
    ref = load_leaf("avssl/module/losses.py", "ref_losses_v")
    for name, kw in [("loss_v_margin", dict(margin=0.3)), ("loss_v_dcl", dict(dcl=True)), ("loss_v_a2b", dict(b2a=False)),
                     ("loss_v_b2a", dict(a2b=False)), ("loss_v_margin_dcl_trainT", dict(margin=0.2, dcl=True, temperature_trainable=True))]:
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 2**31)
        B, E = 40, 32
        A = unit(torch.randn(B, E, generator=g)).requires_grad_(True)
        Bm = unit(torch.randn(B, E, generator=g) + 0.5 * A.detach()).requires_grad_(True)
        ids = torch.arange(B) // 5
        crit = ref.MaskedContrastiveLoss(temperature=0.07, **kw)
        loss = crit(A, Bm, ids)
        loss.backward()
        extra = {}
        if kw.get("temperature_trainable"):
            extra["dtemp_param"] = crit.temperature.grad
        loss_noidx = crit(A.detach(), Bm.detach(), None)
        npz(name + ".npz", A=A, B=Bm, ids=ids, loss=loss, dA=A.grad, dB=Bm.grad, loss_noindex=loss_noidx,
            margin=np.float32(kw.get("margin", 0.0)), dcl=np.int64(kw.get("dcl", False)), a2b=np.int64(kw.get("a2b", True)),
            b2a=np.int64(kw.get("b2a", True)), trainT=np.int64(kw.get("temperature_trainable", False)), **extra)


# --------------------------------------------------------------------------- weighted sum
def make_wsum():
    ws = load_leaf("avssl/module/weighted_sum.py", "ref_ws")
    torch.manual_seed(5)
    layer = ws.WeightedSumLayer(n_weights=13)
    with torch.no_grad():
        layer.weights.copy_(torch.randn(13))
    hs = [torch.randn(2, 9, 32) for _ in range(13)]
    out = layer(hs)
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    layer_n = ws.WeightedSumLayer(n_weights=13, normalize_features=True)
    with torch.no_grad():
        layer_n.weights.copy_(layer.weights)
        out_n = layer_n(hs)
    npz("wsum.npz", weights=layer.weights, hs=torch.stack(hs), out=out, gout=gout, dweights=layer.weights.grad, out_norm=out_n)


# --------------------------------------------------------------------------- masks / lengths
def make_masks():
    du = load_leaf("avssl/util/data_utils.py", "ref_du")
    import torch.nn.functional as F
    from transformers import HubertConfig, HubertModel
    lens = torch.tensor([0, 1, 5, 16, 17])
    m = du.get_keypadding_mask(17, lens)
    # conv length formula vs real conv stack output sizes and vs HF's formula
    Ls = [400, 401, 799, 800, 1000, 16000, 32000, 102400, 159999, 160000, 160001, 163840]
    hf = HubertModel(HubertConfig(num_hidden_layers=1, hidden_size=32, intermediate_size=32, num_attention_heads=2,
                                  conv_dim=(2,) * 7))
    T_hf = [int(hf._get_feat_extract_output_lengths(torch.tensor(L))) for L in Ls]
    T_conv = []
    for L in Ls:
        x = torch.zeros(1, 1, L)
        for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
            x = F.conv1d(x, torch.zeros(1, 1, k), stride=s)
        T_conv.append(x.shape[-1])
    assert T_hf == T_conv, (T_hf, T_conv)
    # fairseq forward_padding_mask executed literally on bool masks (its 5 published lines)
    Lmax_cases, valid = [], []
    for Lmax, wl in [(160000, [160000, 159681, 159680, 320, 1, 80000, 321, 640, 641]),
                     (102400, [102400, 102000, 96400, 321, 322, 642, 643, 51200]),
                     (16000, [16000, 15680, 15681, 3200, 323, 324])]:
        T = T_conv[Ls.index(Lmax)]
        pm = torch.arange(Lmax).unsqueeze(0) >= torch.tensor(wl).unsqueeze(1)
        extra = pm.size(1) % T
        if extra > 0:
            pm = pm[:, :-extra]
        fm = pm.view(pm.size(0), T, -1).all(-1)
        Lmax_cases.append(Lmax)
        valid.append(np.array([wl, (~fm).sum(1).tolist()]))
    npz("masks.npz", kpm_lens=lens, kpm=m, Ls=np.array(Ls), T=np.array(T_conv),
        fm_L0=valid[0], fm_L1=valid[1], fm_L2=valid[2], fm_Lmax=np.array(Lmax_cases),
        round_in=np.array([160, 480, 800, 1120, 159, 161, 479, 481, 160000, 102400, 102000]),
        round_out=np.array([round(l / 320) for l in [160, 480, 800, 1120, 159, 161, 479, 481, 160000, 102400, 102000]]))


# --------------------------------------------------------------------------- retrieval
def make_retrieval():
    rt = load_leaf("avssl/module/retrieval.py", "ref_rt")
    g = torch.Generator().manual_seed(11)
    nA, nB = 40, 8
    a_ids = torch.arange(nA) // 5
    b_ids = torch.arange(nB)
    img = unit(torch.randn(nB, 16, generator=g))
    aud = unit(img[a_ids] + 0.9 * torch.randn(nA, 16, generator=g))
    score = aud @ img.T
    AB, BA, mean = rt.mutualRetrieval(score_per_A=score, score_per_B=score.T, AB_answers=a_ids, BA_answers=b_ids,
                                      recall_at=[1, 5, 10])
    ks = [1, 5, 10]
    npz("retrieval.npz", aud=aud, img=img, a_ids=a_ids, b_ids=b_ids, score=score,
        AB=np.array([AB[f"recall@{k}"] for k in ks]), BA=np.array([BA[f"recall@{k}"] for k in ks]),
        mean=np.array([mean[f"recall@{k}"] for k in ks]))


# --------------------------------------------------------------------------- HuBERT (HF cross-check)
def make_hubert():
    from transformers import HubertConfig, HubertModel
    import oracle
    from oracle.hubert_ref import fold_weight_norm
    for name, stable in [("hubert_small", False), ("hubert_small_preln", True)]:
        torch.manual_seed(3 if not stable else 4)
        cfg = HubertConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
                           conv_dim=(16,) * 7, conv_bias=stable, feat_extract_norm="layer" if stable else "group",
                           do_stable_layer_norm=stable, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                           hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0,
                           layerdrop=0.0, apply_spec_augment=False, feat_proj_layer_norm=True)
        hf = HubertModel(cfg).eval()
        with torch.no_grad():
            for n, p in hf.named_parameters():
                if "norm" in n or n.endswith("bias"):
                    p.add_(0.1 * torch.randn_like(p))
        sd = hf.state_dict()
        W = {}
        for i in range(7):
            W[f"feature_extractor.conv_layers.{i}.0.weight"] = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
            if stable:
                W[f"feature_extractor.conv_layers.{i}.0.bias"] = sd[f"feature_extractor.conv_layers.{i}.conv.bias"]
                W[f"feature_extractor.conv_layers.{i}.2.1.weight"] = sd[f"feature_extractor.conv_layers.{i}.layer_norm.weight"]
                W[f"feature_extractor.conv_layers.{i}.2.1.bias"] = sd[f"feature_extractor.conv_layers.{i}.layer_norm.bias"]
            elif i == 0:
                W["feature_extractor.conv_layers.0.2.weight"] = sd["feature_extractor.conv_layers.0.layer_norm.weight"]
                W["feature_extractor.conv_layers.0.2.bias"] = sd["feature_extractor.conv_layers.0.layer_norm.bias"]
        W["layer_norm.weight"] = sd["feature_projection.layer_norm.weight"]
        W["layer_norm.bias"] = sd["feature_projection.layer_norm.bias"]
        W["post_extract_proj.weight"] = sd["feature_projection.projection.weight"]
        W["post_extract_proj.bias"] = sd["feature_projection.projection.bias"]
        gk = [k for k in sd if "pos_conv_embed" in k]
        g_key = [k for k in gk if k.endswith("original0") or k.endswith("weight_g")][0]
        v_key = [k for k in gk if k.endswith("original1") or k.endswith("weight_v")][0]
        W["encoder.pos_conv.0.weight"] = fold_weight_norm(sd[g_key], sd[v_key])
        W["encoder.pos_conv.0.bias"] = sd["encoder.pos_conv_embed.conv.bias"]
        W["encoder.layer_norm.weight"] = sd["encoder.layer_norm.weight"]
        W["encoder.layer_norm.bias"] = sd["encoder.layer_norm.bias"]
        for i in range(2):
            p, q = f"encoder.layers.{i}.", f"encoder.layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                W[p + f"self_attn.{n}.weight"] = sd[q + f"attention.{n}.weight"]
                W[p + f"self_attn.{n}.bias"] = sd[q + f"attention.{n}.bias"]
            W[p + "self_attn_layer_norm.weight"] = sd[q + "layer_norm.weight"]
            W[p + "self_attn_layer_norm.bias"] = sd[q + "layer_norm.bias"]
            W[p + "fc1.weight"] = sd[q + "feed_forward.intermediate_dense.weight"]
            W[p + "fc1.bias"] = sd[q + "feed_forward.intermediate_dense.bias"]
            W[p + "fc2.weight"] = sd[q + "feed_forward.output_dense.weight"]
            W[p + "fc2.bias"] = sd[q + "feed_forward.output_dense.bias"]
            W[p + "final_layer_norm.weight"] = sd[q + "final_layer_norm.weight"]
            W[p + "final_layer_norm.bias"] = sd[q + "final_layer_norm.bias"]
        B, L = 3, 8000
        wav = torch.randn(B, L)
        with torch.no_grad():
            out = hf(wav, output_hidden_states=True)
        hs = torch.stack(out.hidden_states)                       # (3, B, T, D); [0] = encoder input (post-LN variant)
        # padded batch where HF's conv-formula mask and fairseq's chunk-all() mask agree on the valid count
        # for every utterance (lens chosen so) -> comparable on valid frames
        lens = [8000, 4980, 2320]   # both mask rules give the same valid-frame count for these
        am = (torch.arange(L).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)).long()
        wav_p = wav * am
        with torch.no_grad():
            out_p = hf(wav_p, attention_mask=am, output_hidden_states=True)
        hf_valid = hf._get_feature_vector_attention_mask(out_p.last_hidden_state.shape[1], am).sum(1)
        arch = oracle.HubertArch(embed_dim=32, ffn_dim=64, layers=2, heads=4, conv_dim=16,
                                 extractor_mode="layer_norm" if stable else "default", conv_bias=stable,
                                 layer_norm_first=stable)
        # sanity: oracle reproduces HF on the un-padded batch (this is the cross-check itself)
        mine = oracle.hubert_forward(W, arch, wav, None)
        if not stable:
            err = max(float((m - h).abs().max()) for m, h in zip(mine, hs))
        else:
            # HF pre-LN: hidden_states[i] are the same pre-final-LN layer outputs except the last, which gets encoder.layer_norm
            err = max(float((m - h).abs().max()) for m, h in zip(mine[:-1], hs[:-1]))
        print(name, "oracle vs HF un-padded max abs err", err)
        assert err < 2e-4, err
        npz(name + ".npz", wav=wav, hf_hidden=hs, lens=np.array(lens), hf_hidden_padded=torch.stack(out_p.hidden_states),
            hf_valid=hf_valid, stable=np.int64(stable), **{"W_" + k: v for k, v in W.items()})


# --------------------------------------------------------------------------- cascaded+/hybrid+ pieces (row a11)
def make_cascaded():
    cif = load_leaf("avssl/module/cif.py", "ref_cif")
    vqm = load_leaf("avssl/module/speechclip_c_modules/my_vector_quantizer.py", "ref_vq")
    bnm = load_leaf("avssl/module/speechclip_c_modules/kw_bn.py", "ref_bn")
    torch.manual_seed(31)
    D, B, T = 32, 4, 60
    m = cif.CIF(cif_threshold=1.0, cif_output_dim=D, encoder_embed_dim=D, produce_weight_type="conv", num_layer=1,
                conv_cif_width=3, apply_scaling=True, apply_tail_handling=True, tail_handling_firing_threshold=0.5,
                scaling_step=5000).eval()          # eval(): the p=0.5 dropouts of the weight generator are off
    with torch.no_grad():
        m.weight_proj[1].bias.add_(-0.5)           # alphas around 0.3-0.6 -> several fires per utterance
    feat = torch.randn(B, T, D, requires_grad=True)
    lens = torch.tensor([60, 41, 23, 7])
    pad = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    out = {}
    # training path: target lengths given (scaling on, tail dropped)
    tgt = (lens / 20).round().long().clamp(min=1)
    r = m({"audio_feat": feat, "audio_feat_pad_mask": pad, "global_step": 0}, tgt)
    g = torch.randn_like(r["dsample_feats"])
    (r["dsample_feats"] * g).sum().backward()
    out.update(tr_feats=r["dsample_feats"], tr_len=r["dsample_feats_length"], tr_quantity=r["quantity_out"],
               tr_alpha=r["alpha"], tr_gout=g, tr_gfeat=feat.grad.clone(), tr_target=tgt,
               tr_gconvw=m.conv[0].weight.grad.clone())
    # inference path: no target (no scaling, tail firing)
    with torch.no_grad():
        r = m({"audio_feat": feat.detach(), "audio_feat_pad_mask": pad, "global_step": 0}, None)
    out.update(ev_feats=r["dsample_feats"], ev_len=r["dsample_feats_length"], ev_quantity=r["quantity_out"],
               ev_pad=r["dsample_feats_pad_mask"], ev_fired=r["fired_marks"])
    # after the scaling step: apply_scaling switches off for good
    with torch.no_grad():
        r = m({"audio_feat": feat.detach(), "audio_feat_pad_mask": pad, "global_step": 6000}, tgt)
    out.update(ns_feats=r["dsample_feats"], ns_len=r["dsample_feats_length"])
    npz("cif_d32.npz", feat=feat, lens=lens, **out, **{"W_" + k: v for k, v in m.state_dict().items()})

    # vector quantizer: eval (hard one-hot) and train (straight-through softmax / 0.1)
    torch.manual_seed(32)
    V = 50
    x = torch.randn(3, 5, V) * 0.3
    vq = vqm.SimpleVectorQuantizer(temp="fixed=0.1", time_first=True, use_gumbel=False, hard=True)
    vq.eval()
    re = vq(x=x.clone())
    vq.train()
    xt = x.clone().requires_grad_(True)
    rt = vq(x=xt * 1.0)
    emb = torch.randn(V, 8)
    gk = torch.randn(3, 5, 8)
    ((rt["subword_prob"] @ emb) * gk).sum().backward()
    npz("vq_v50.npz", x=x, emb=emb, gk=gk, ev_prob=re["subword_prob"], ev_targets=re["targets"],
        ev_code_ppl=re["code_perplexity"], ev_prob_ppl=re["prob_perplexity"], ev_ent=re["ent_per_t"],
        ev_div=re["diversity_loss"], tr_prob=rt["subword_prob"], tr_gx=xt.grad)

    # keyword BatchNorm (dynamic): train-mode batch statistics and eval-mode running statistics
    torch.manual_seed(33)
    E = 16
    bn = bnm.Kw_BatchNorm_dynamic(kw_dim=E, init_bias=torch.randn(E) * 0.1, init_scale=torch.rand(E) + 0.5, std_scale=1.0)
    kw = torch.randn(4, 6, E)
    bn.train()
    y_tr = bn(kw)
    bn.eval()
    y_ev = bn(kw)
    npz("kwbn_e16.npz", kw=kw, y_train=y_tr, y_eval=y_ev, **{"W_" + k: v for k, v in bn.state_dict().items()},
        init_weight=bn.bn_layer.weight, init_bias=bn.bn_layer.bias)


# --------------------------------------------------------------------------- CLIP text tower (HF cross-check)
def make_clip_text():
    """openai/CLIP is absent offline (requirements.txt:4), so the text tower every cascaded+/hybrid+ embedding flows through
    (clip_official.py:222-279) is cross-checked the way HuBERT is: against transformers' INDEPENDENT implementation built from a
    local config.  The HF model runs (a) its own forward on plain token ids and (b) its encoder on hand-spliced embeddings with
    an explicit additive causal mask (what encode_keywords feeds the tower: sot, n keyword vectors, eot, token 0 ...); weights are
    stored under openai/CLIP's names so the oracle reads them as it reads the reference's state dict."""
    for name, seed, Wd, heads, E in [("clip_text_w64", 41, 64, 4, 24), ("clip_text_w128", 42, 128, 2, 32)]:
        _clip_text_case(name, seed, Wd, heads, E)


def _clip_text_case(name, seed, Wd, heads, E):
    """w64: head_dim 16; w128: head_dim 64 = what the HIP tower is built for (the -m gpu test runs on this one)."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    layers, V, S = 2, 100, 77
    sot, eot = V - 2, V - 1
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=Wd, intermediate_size=4 * Wd, projection_dim=E, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=S, hidden_act="quick_gelu", bos_token_id=sot,
                         eos_token_id=eot, pad_token_id=0, attention_dropout=0.0)
    cfg._attn_implementation = "eager"
    hf = CLIPTextModelWithProjection(cfg).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "norm" in n or n.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
            elif "embedding" in n or "projection" in n:
                p.copy_(torch.randn_like(p) * (0.3 if "token" in n else 0.1))
    sd = hf.state_dict()
    P = "clip.model."
    W = {P + "token_embedding.weight": sd["text_model.embeddings.token_embedding.weight"],
         P + "positional_embedding": sd["text_model.embeddings.position_embedding.weight"],
         P + "ln_final.weight": sd["text_model.final_layer_norm.weight"], P + "ln_final.bias": sd["text_model.final_layer_norm.bias"],
         P + "text_projection": sd["text_projection.weight"].t().contiguous()}
    for i in range(layers):
        a, b = f"text_model.encoder.layers.{i}.", f"{P}transformer.resblocks.{i}."
        W[b + "attn.in_proj_weight"] = torch.cat([sd[a + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        W[b + "attn.in_proj_bias"] = torch.cat([sd[a + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        for src, dst in [("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                         ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")]:
            W[b + dst + ".weight"], W[b + dst + ".bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
    # (a) plain token ids through HF's own forward (its embeddings, its causal mask, its EOS pooling, its projection)
    n_kw = torch.tensor([3, 1, 75, 10, 8])
    B = len(n_kw)
    ids = torch.zeros(B, S, dtype=torch.long)
    ids[:, 0] = sot
    tok = torch.randint(4, V - 2, (B, 75))
    for b in range(B):
        ids[b, 1: 1 + n_kw[b]] = tok[b, : n_kw[b]]
        ids[b, 1 + n_kw[b]] = eot
    with torch.no_grad():
        out_ids = hf(input_ids=ids)
    # (b) spliced continuous keyword vectors through HF's encoder layers
    kw = (torch.randn(B, 75, Wd) * 0.3).requires_grad_(True)
    emb = sd["text_model.embeddings.token_embedding.weight"]
    rows = []
    for b in range(B):
        n = int(n_kw[b])
        rows.append(torch.cat([emb[sot][None], kw[b, :n], emb[eot][None], emb[0].expand(S - 2 - n, Wd)]))
    x = torch.stack(rows) + sd["text_model.embeddings.position_embedding.weight"]
    causal = torch.full((S, S), float("-inf")).triu_(1)[None, None].expand(B, 1, S, S)
    h = hf.text_model.encoder(inputs_embeds=x, attention_mask=causal).last_hidden_state
    last = hf.text_model.final_layer_norm(h)
    out_kw = hf.text_projection(last[torch.arange(B), n_kw + 1])
    gout = torch.randn_like(out_kw)
    (out_kw * gout).sum().backward()
    npz(name + ".npz", ids=ids, tok=tok, n_kw=n_kw, out_ids=out_ids.text_embeds, last_hidden_ids=out_ids.last_hidden_state,
        kw=kw, tower_out_kw=h.detach(), out_kw=out_kw, gout=gout, g_kw=kw.grad, heads=np.int64(heads), sot=np.int64(sot),
        eot=np.int64(eot), **{"W_" + k: v for k, v in W.items()})


# --------------------------------------------------------------------------- in-forward training crop
def make_crop():
    """random_crop_max_length (avssl/data/audio_transforms.py:5-23) driven exactly as speech_encoder_plus.py:548-552 drives it: a
    loop over the un-padded utterances under one numpy seed per case.  Each "waveform" holds its own sample indices, so what comes
    back IS the (offset, length) window the reference cut."""
    at = load_leaf("avssl/data/audio_transforms.py", "ref_audio_transforms")
    rs = np.random.RandomState(11)
    cases = {}
    for ci, (B, max_len) in enumerate([(8, 102400), (64, 102400), (16, 1000), (5, -1), (12, 4000)]):
        lens = rs.randint(max(1, abs(max_len) // 3), 3 * abs(max_len), size=B)
        lens[0] = abs(max_len)                       # exactly at the cap: returned whole, NO draw
        if B > 2:
            lens[2] = abs(max_len) + 1               # one over: randint(1) == 0, a draw is consumed
        seed = 1000 + ci
        np.random.seed(seed)
        offs, outl = [], []
        for b in range(B):
            w = torch.arange(int(lens[b]))
            o = at.random_crop_max_length(w, max_len, len(w))
            offs.append(int(o[0]) if len(o) else 0)
            outl.append(len(o))
            assert len(o) == 0 or torch.equal(o, torch.arange(int(o[0]), int(o[0]) + len(o)))
        after = np.random.randint(1 << 30)           # the generator's state behind the loop: the number of draws is pinned too
        cases[f"c{ci}_lens"], cases[f"c{ci}_meta"] = lens, np.asarray([B, max_len, seed, after])
        cases[f"c{ci}_off"], cases[f"c{ci}_out"] = np.asarray(offs), np.asarray(outl)
    npz("crop.npz", n=np.asarray(5), **cases)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["loss", "head", "mha", "loss_variants", "wsum", "masks", "retrieval", "hubert", "cascaded", "crop", "clip_text"]
    for what in todo:                       # e.g. `make_golden.py mha loss_variants` regenerates only those fixtures
        globals()["make_" + what]()
