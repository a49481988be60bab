"""GPU: the kernels of the cascaded+/hybrid+ tail on their own (csrc/vq.hip, the CIF bookkeeping of csrc/cif.hip, the padded-head
attention block) against plain fp32 / fp64 torch references of the same op and against the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import weights_from

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (200, 333, 52), (1600, 8112, 512), (77, 5, 4), (1, 130, 768), (6, 512, 6), (13, 7, 5)])
@pytest.mark.parametrize("a_k,b_k", [(False, False), (True, True), (False, True), (True, False)])
def test_sgemm_mfma_f32_every_layout_vs_fp64(M, N, K, a_k, b_k):
    """exact-fp32 MFMA GEMM: every operand layout, ragged M / N / K, bias; error at the fp32 rounding level."""
    from speechclip_plus_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double().t() + bias.double()

    def lay(x, kmajor):
        return x.t().contiguous().cuda() if kmajor else x.cuda().contiguous()

    out = ops.sgemm_mfma(lay(A, a_k), lay(B, b_k), a_kmajor=a_k, b_kmajor=b_k, bias=bias.cuda())
    assert out.shape == (M, N)
    err = (out.double().cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-6 * max(1.0, K ** 0.5 / 4), float(err)


def test_linear_f32_autograd_matches_torch():
    from speechclip_plus_amd.linear_fn import linear_f32_autograd
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 37, 96, generator=g).cuda().requires_grad_()
    w = (torch.randn(40, 96, generator=g) * 0.1).cuda().requires_grad_()
    b = torch.randn(40, generator=g).cuda().requires_grad_()
    gy = torch.randn(3, 37, 40, generator=g).cuda()
    linear_f32_autograd(x, w, b).backward(gy)
    got = [x.grad.clone(), w.grad.clone(), b.grad.clone()]
    x.grad = w.grad = b.grad = None
    y = F.linear(x.double(), w.double(), b.double())
    y.backward(gy.double())
    assert rel(linear_f32_autograd(x, w, b), y) < 1e-6
    for a, r in zip(got, [x.grad, w.grad, b.grad]):
        assert rel(a, r) < 1e-5


def _vq_reference(kw, table, temp, training):
    """kw_branches.py:158-197 + my_vector_quantizer.py:64-165 in fp64 torch."""
    kw, table = kw.double(), table.double()
    cos = F.normalize(kw, dim=-1, eps=1e-8) @ F.normalize(table, dim=-1, eps=1e-8).t()
    x = cos.clone()
    x[:, [0, 2, 3]] = float("-inf")
    k = x.argmax(-1)
    hard = F.one_hot(k, x.shape[-1]).double()
    soft = torch.softmax(x / temp, -1)
    prob = hard + soft - soft.detach() if training else hard
    p1 = torch.softmax(x, -1)
    ent = -(p1 * torch.log(p1 + 1e-9)).sum(-1)
    hp = hard.mean(0)
    code_ppl = torch.exp(-(hp * torch.log(hp + 1e-7)).sum())
    ap = p1.mean(0)
    prob_ppl = torch.exp(-(ap * torch.log(ap + 1e-7)).sum())
    return prob @ table, k, ent, code_ppl, prob_ppl, x


@pytest.mark.parametrize("Nk,V,Et", [(512, 8112, 512), (512, 19787, 768), (37, 205, 48)])
def test_cosine_scores_as_one_bf16_gemm_over_three_way_splits_vs_fp64(Nk, V, Et):
    """Round 6: the keyword quantiser's cosine scores = ONE bf16 GEMM over the six K-blocks of the operands' three-way bf16 splits
    (sc_split3_bf16 + sc_gemm_bf16).  Every bf16 x bf16 product is exact in fp32, the six blocks carry the product to 2^-24, so the
    result must sit as close to the fp64 product as the exact-fp32 MFMA GEMM it replaces does (criterion fixed before the first run:
    max |error| <= 2 x the fp32 GEMM's own max error + 1e-7, and the split is a lossless decomposition to 2^-22 relative)."""
    from speechclip_plus_amd import ops
    g = torch.Generator().manual_seed(V + Et)
    table = torch.randn(V, Et, generator=g) * 0.02
    kw = torch.randn(Nk, Et, generator=g) * 3.0
    wn = table / table.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    kw_d, wn_d = kw.cuda(), wn.cuda().contiguous()
    kwn_T, rnorm = ops.vq_prep(kw_d)
    Vp = (V + 127) // 128 * 128
    split_tab = ops.split3_bf16(wn_d, 1, rows_pad=128)
    # the decomposition itself: blocks 0, 1, 3 of a side-1 row are x1, x2, x3
    Ep = split_tab.shape[1] // 6
    parts = split_tab[:V].view(V, 6, Ep)[:, [0, 1, 3], :Et].double().sum(1)
    assert float((parts - wn_d.double()).abs().max()) <= 2.0 ** -22 * float(wn_d.abs().max())
    assert float(split_tab[V:].float().abs().max() if Vp > V else 0.0) == 0.0
    cos_split = ops.cosine_scores_split(kw_d, rnorm, split_tab, Vp)[:Nk, :V].double().cpu()
    norm_T = torch.zeros(Et, Vp, device="cuda")
    norm_T[:, :V] = wn_d.t()
    cos_fp32 = ops.sgemm_mfma(kwn_T, norm_T, a_kmajor=True, b_kmajor=True)[:Nk, :V].double().cpu()
    kwn = (kw_d * rnorm[:, None]).double().cpu()              # the fp32 normalised keywords both paths start from
    ref = kwn @ wn.double().t()
    e_split, e_fp32 = float((cos_split - ref).abs().max()), float((cos_fp32 - ref).abs().max())
    print(f"max |error| vs fp64: split-bf16 GEMM {e_split:.3e}, exact-fp32 MFMA GEMM {e_fp32:.3e}")
    assert e_split <= 2.0 * e_fp32 + 1e-7, (e_split, e_fp32)
    # the decisions: identical argmax wherever the fp64 margin of the two best scores exceeds 1e-6
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-6
    assert torch.equal(cos_split.argmax(-1)[safe], ref.argmax(-1)[safe]) and float(safe.float().mean()) > 0.99


@pytest.mark.parametrize("Nk,V,Et", [(300, 1000, 64), (1600, 8112, 512), (37, 205, 48)])
def test_fused_keyword_vq_vs_fp64(Nk, V, Et):
    """SimpleVectorQuantizer.quantize_keywords (cosine in exact fp32 MFMA -> mask -> argmax -> gather; straight-through backward)
    against the fp64 statement of the reference's formulas: identical tokens, statistics, and the gradient to the keywords."""
    from speechclip_plus_amd.vector_quantizers import SimpleVectorQuantizer
    g = torch.Generator().manual_seed(Nk + V)
    table = (torch.randn(V, Et, generator=g) * 0.02)
    B = 4 if Nk % 4 == 0 else 1
    kw = torch.randn(B, Nk // B, Et, generator=g)
    gout = torch.randn(B, Nk // B, Et, generator=g)
    vq = SimpleVectorQuantizer(temp="fixed=0.1").cuda()
    table_d = table.cuda()
    for training in (False, True):
        vq.train(training)
        kw_d = kw.cuda().requires_grad_()
        res, out = vq.quantize_keywords(kw_d, table_d)
        kw_r = kw.clone().double().requires_grad_()
        ref_out, k, ent, code_ppl, prob_ppl, x = _vq_reference(kw_r.reshape(Nk, Et), table, 0.1, training)
        # near-ties of the two best scores may legitimately resolve differently in fp32: none expected at these sizes
        top2 = x.topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-6
        tok = res["targets"].reshape(-1).cpu()
        assert torch.equal(tok[safe], k[safe]) and float(safe.float().mean()) > 0.99
        assert rel(out.reshape(Nk, Et)[safe.cuda()], ref_out[safe]) < 1e-6
        assert abs(float(res["code_perplexity"]) - float(code_ppl)) < 1e-3 * float(code_ppl)
        assert abs(float(res["prob_perplexity"]) - float(prob_ppl)) < 1e-4 * float(prob_ppl)
        assert rel(res["ent_per_t"], ent.view(B, Nk // B).mean(0)) < 1e-5
        assert res["num_vars"] == V and res["temp"] == 0.1
        assert torch.equal(res["subword_prob"].argmax(-1).reshape(-1).cpu(), tok)          # lazily materialised one-hot
        if training:
            out.backward(gout.cuda())
            ref_out.backward(gout.reshape(Nk, Et).double())
            e = rel(kw_d.grad.reshape(Nk, Et)[safe.cuda()], kw_r.grad.reshape(Nk, Et)[safe])
            assert e < 2e-2, e                                    # bf16 operands in the two gradient GEMMs
        else:
            assert not out.requires_grad or out.grad_fn is not None


def test_keyword_batchnorm_kernels_vs_torch():
    from speechclip_plus_amd.vector_quantizers import Kw_BatchNorm_dynamic
    g = torch.Generator().manual_seed(3)
    E, B, N = 70, 5, 9
    bn = Kw_BatchNorm_dynamic(E, torch.randn(E, generator=g) * 0.1, torch.rand(E, generator=g) + 0.5).cuda()
    ref = torch.nn.BatchNorm1d(E).double()
    ref.load_state_dict({k: v.double().cpu() if v.is_floating_point() else v.cpu() for k, v in bn.bn_layer.state_dict().items()})
    x = torch.randn(B, N, E, generator=g)
    gy = torch.randn(B, N, E, generator=g)
    for _ in range(2):                                             # two training steps: running statistics accumulate
        xd = x.cuda().requires_grad_()
        y = bn.train()(xd)
        y.backward(gy.cuda())
        xr = x.double().requires_grad_()
        yr = ref.train()(xr.permute(0, 2, 1)).permute(0, 2, 1)
        yr.backward(gy.double())
        assert rel(y, yr) < 1e-5 and rel(xd.grad, xr.grad) < 1e-4
        assert rel(bn.bn_layer.weight.grad, ref.weight.grad) < 1e-4 and rel(bn.bn_layer.bias.grad, ref.bias.grad) < 1e-4
        bn.zero_grad()
        ref.zero_grad()
    assert rel(bn.bn_layer.running_mean, ref.running_mean) < 1e-5 and rel(bn.bn_layer.running_var, ref.running_var) < 1e-5
    assert int(bn.bn_layer.num_batches_tracked) == 2
    assert rel(bn.eval()(x.cuda()), ref.eval()(x.double().permute(0, 2, 1)).permute(0, 2, 1)) < 1e-5


@pytest.mark.parametrize("C,S", [(24, 90), (768, 499), (1024, 200)])
@pytest.mark.parametrize("scaled", [True, False])
def test_cif_bookkeeping_and_fire_vs_oracle(scaled, C, S):
    """sc_cif_prepare + sc_cif_fwd (+ sc_cif_tail) and their backward against the oracle's scatter_add_ form (autograd): outputs,
    counts, gradient to the features AND to the raw weights (through the scan, the scaling and the quantity output), multi-fire
    frames included; no host read while the targets are known on the host."""
    from oracle.cascaded_ref import integrate_and_fire
    from speechclip_plus_amd.cif import CIF, _CifFn
    g = torch.Generator().manual_seed(5 + scaled)
    B = 6
    lens = torch.tensor([S, S * 2 // 3, S // 3, S, 12, 5])
    pad = torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)
    x = torch.randn(B, S, C, generator=g)
    a_raw = torch.rand(B, S, generator=g) * 0.9
    a_raw[0, 10] = 1.0
    tgt = torch.tensor([40, 9, 3, S + 40, 0, 2])       # more keywords than frames: multi-fire frames and the 75 cap
    gout_seed = torch.Generator().manual_seed(9)
    # ---- oracle (fp32 autograd)
    xr = x.clone().requires_grad_()
    ar = a_raw.clone().requires_grad_()
    a = ar.clip(0, 1).masked_fill(pad, 0.0)
    q = a.sum(1)
    a2 = a * ((1.0 * tgt.float() + 1e-5) / q).unsqueeze(1) if scaled else a
    ref, ref_len = integrate_and_fire(xr, a2, 1.0, target_given=scaled)
    gout = torch.randn(ref.shape, generator=gout_seed)
    gq = torch.randn(B, generator=gout_seed)
    ((ref * gout).sum() + (q * gq).sum()).backward()
    # ---- kernels
    m = CIF(cif_output_dim=C, encoder_embed_dim=C).cuda()
    flags = m.consistency_flags
    xd = x.cuda().requires_grad_()
    ad = a_raw.cuda().requires_grad_()
    host = [int(t) for t in tgt]
    T_known = max(min(max(t, 1), 75) for t in host) if scaled else None
    st = {"thr": 1.0, "eps": 1e-5, "apply_scaling": scaled, "scaled": scaled, "tail": not scaled, "tail_thr": 0.5, "flags": flags,
          "T_clip": T_known if scaled else 75}
    slots, quantity, feat_len = _CifFn.apply(xd, ad, pad.cuda(), tgt.cuda() if scaled else None, st)
    Tn = T_known if scaled else int(feat_len.max())
    out = slots[:, :Tn]
    assert feat_len.cpu().tolist() == ref_len.tolist()
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-5 and rel(quantity, q) < 1e-6
    ((out * gout.cuda()).sum() + (quantity * gq.cuda()).sum()).backward()
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(ad.grad, ar.grad) < 2e-5, rel(ad.grad, ar.grad)
    pos, mism = flags.tolist()[:2]
    assert pos == B and mism == 0


@pytest.mark.parametrize("name,tol", [("mha_norm_d128_h2", 2.5e-2), ("mha_norm_d192_h2", 2.5e-2), ("mha_norm_d64_h8", 2.5e-2)])
def test_mha_norm_block_vs_reference_fixture(golden, name, tol):
    """MultiheadAttentionAndNorm on the kernel path against the reference's own module (fixtures from TransformerModels.py):
    head_dim 64, head_dim 96 (zero-padded to 128 inside the projection weights) and head_dim 8 (padded to 64)."""
    from speechclip_plus_amd.transformer_models import MultiheadAttentionAndNorm
    fx = golden(name + ".npz")
    D, H = fx["src"].shape[-1], int(fx["nhead"])
    blk = MultiheadAttentionAndNorm(d_model=D, nhead=H, dropout=0.1, layer_norm_eps=1e-5, batch_first=True)
    blk.load_state_dict(weights_from(fx), strict=True)
    blk = blk.cuda().eval()
    src = T(fx["src"]).cuda().requires_grad_()
    lens = T(np.asarray(fx["lens"])).cuda()
    kpm = torch.arange(src.shape[1], device="cuda").unsqueeze(0) >= lens.unsqueeze(1)
    out = blk(src, kpm)
    valid = (~kpm).unsqueeze(-1).cpu()
    assert rel(out.cpu() * valid, T(fx["out"]) * valid) < tol
    if "gout" in fx:
        (out * (T(fx["gout"]) * valid).cuda()).sum().backward()
        # the fixture's gradient includes the padded query rows; restrict the reference to the valid ones by linearity is not
        # possible from stored data, so compare the parameter gradients only where every row is valid (utterance 0) is not
        # separable either: use utterances whose padded rows carry zero upstream gradient -> recompute the reference here
        blk_ref = torch.nn.MultiheadAttention(D, H, dropout=0.0, batch_first=True).double()
        ln_ref = torch.nn.LayerNorm(D, eps=1e-5).double()
        sd = weights_from(fx)
        blk_ref.load_state_dict({k.replace("multihead_attn_layer.", ""): v.double() for k, v in sd.items() if k.startswith("multihead")})
        ln_ref.load_state_dict({k.replace("attentionBlock_Norm.", ""): v.double() for k, v in sd.items() if k.startswith("attentionBlock")})
        s2 = T(fx["src"]).double().requires_grad_()
        o2 = ln_ref(blk_ref(s2, s2, s2, key_padding_mask=kpm.cpu())[0] + s2)
        assert rel(o2 * valid, T(fx["out"]) * valid) < 1e-5            # the in-test reference reproduces the fixture
        (o2 * (T(fx["gout"]) * valid).double()).sum().backward()
        assert rel(src.grad, s2.grad) < 4e-2
        pairs = [(blk.multihead_attn_layer.in_proj_weight, blk_ref.in_proj_weight), (blk.multihead_attn_layer.in_proj_bias, blk_ref.in_proj_bias),
                 (blk.multihead_attn_layer.out_proj.weight, blk_ref.out_proj.weight), (blk.multihead_attn_layer.out_proj.bias, blk_ref.out_proj.bias),
                 (blk.attentionBlock_Norm.weight, ln_ref.weight), (blk.attentionBlock_Norm.bias, ln_ref.bias)]
        for got, want in pairs:
            assert rel(got.grad, want.grad) < 4e-2, (tuple(got.shape), rel(got.grad, want.grad))


@pytest.mark.parametrize("name", ["head_d64_h8", "head_d64_h1"])
def test_transformer_encoder_full_sequence_vs_reference_fixture(golden, name):
    """TransformerEncoder.forward / extract_hidden_states over every row of [CLS ; frames] (the path the reference's
    feature_extractor_s3prl and its non-plus branches use) on the kernels, against the reference's full-sequence output."""
    from speechclip_plus_amd.transformer_models import TransformerEncoder
    fx = golden(name + ".npz")
    W = weights_from(fx)
    D, H = fx["feat"].shape[-1], int(fx["nhead"])
    enc = TransformerEncoder(n_layers=1, d_model=D, nhead=H, dim_feedforward=128, dropout=0.1)
    enc.load_state_dict({k[len("self_att."):]: v for k, v in W.items() if k.startswith("self_att.")}, strict=True)
    enc = enc.cuda().eval()
    feat = T(fx["feat"]).cuda()
    B = feat.shape[0]
    src = torch.cat([W["cls"].cuda().expand(B, -1, -1), feat], dim=1).requires_grad_()
    kpm = T(fx["kpm"]).cuda()
    out = enc(src, kpm)
    valid = (~kpm).unsqueeze(-1).cpu()
    assert rel(out.cpu() * valid, T(fx["out_full"]) * valid) < 2.5e-2
    hid = enc.extract_hidden_states(src, kpm)
    assert len(hid) == 2 and rel(hid[-1].cpu() * valid, T(fx["hidden_last"]) * valid) < 2.5e-2
    (out * valid.cuda()).sum().backward()                              # the whole layer is differentiable on the kernel path
    assert torch.isfinite(src.grad).all() and float(src.grad.abs().sum()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in enc.parameters())


def test_cif_prepare_guards_all_zero_weights_and_clamps_to_the_allocated_slots():
    """an utterance whose clipped weights are all zero must not produce inf / NaN under the target-length scaling
    (ratio 0, feat_len 1), and feat_len is clamped to the slots the host allocated (T), not only to MAX_FEAT_LEN."""
    from speechclip_plus_amd import ops
    dev = "cuda:0"
    B, S = 3, 40
    g = torch.Generator().manual_seed(0)
    a = torch.rand(B, S, generator=g)
    a[1] = -0.5                                            # clips to all-zero
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[2, 30:] = True
    target = torch.tensor([4, 3, 9])
    flags = torch.zeros(8, dtype=torch.int32, device=dev)
    r = ops.cif_prepare(a.to(dev), pad.to(dev), target.to(dev), True, 1.0, 1e-5, 75, 6, flags)     # host sized the output for 6 slots
    torch.cuda.synchronize()
    for k in ("alpha", "csum", "quantity", "ratio"):
        assert bool(torch.isfinite(r[k]).all()), k
    assert r["feat_len"].tolist() == [4, 1, 6]             # utterance 2 asked for 9 keywords: clamped to the 6 allocated slots
    assert float(r["alpha"][1].abs().sum()) == 0.0 and float(r["ratio"][1]) == 0.0
    assert flags[0].item() == 2                            # two utterances with a positive weight sum
    assert flags[1].item() == 2                            # utterance 1 (1 != 3) and utterance 2 (6 != 9) disagree with their targets


@pytest.mark.parametrize("p1,p2", [(0.0, 0.0), (0.5, 0.5)])
def test_cif_weight_head_kernel(p1, p2):
    """sc_cif_head_fwd / _bwd (Dropout -> ReLU -> Dropout -> Linear(C, 1) -> Sigmoid behind the weight-generator conv, cif.py:106-129, in
    one row kernel each way) against torch autograd in fp64 with the same hash masks reconstructed on the host."""
    import numpy as np
    from speechclip_plus_amd import ops
    from test_gpu_kernels import _keep_mask
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    rows, C = 777, 768
    y = torch.randn(rows, C, generator=g)
    w = torch.randn(C, generator=g) * C ** -0.5
    b = torch.tensor([0.3])
    da = torch.randn(rows, generator=g)
    s1, s2 = 1234, 98765
    alpha = ops.cif_head_fwd(y.to(dev), w.to(dev), b.to(dev), p1, s1, p2, s2)
    idx = np.arange(rows * C, dtype=np.int64)
    m1 = torch.from_numpy(_keep_mask(idx, s1, p1)).view(rows, C).double() / (1 - p1) if p1 > 0 else torch.ones(rows, C).double()
    m2 = torch.from_numpy(_keep_mask(idx, s2, p2)).view(rows, C).double() / (1 - p2) if p2 > 0 else torch.ones(rows, C).double()
    yd, wd, bd = y.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = torch.sigmoid((torch.relu(yd * m1) * m2) @ wd + bd)
    rel = lambda a, r: float((a.double().cpu() - r.double()).norm() / (r.double().norm() + 1e-30))
    assert rel(alpha, ref.detach()) < 1e-5
    ref.backward(da.double())
    dy, dw, db = ops.cif_head_bwd(y.to(dev), w.to(dev), alpha, da.to(dev), p1, s1, p2, s2)
    assert rel(dy, yd.grad) < 1e-5 and rel(dw, wd.grad) < 1e-5 and rel(db, bd.grad) < 1e-4      # (db: a cancelling fp32 sum over the rows)


@pytest.mark.gpu
def test_prompt_assembly_kernels_match_the_elementwise_formulation():
    """csrc/prompt.hip against the torch formulation it replaces (clip_official.py:233-262 written with zeros / scatter / where / add):
    bit-identical rows, the end-of-text row index incl. a count that points behind the keyword tensor (clamped + counted), pad
    samples and rows behind the prefix zero, and the two adjoints (gather / scatter of the end-of-text rows, keyword gradient)."""
    from speechclip_plus_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, N, W, SEG, Bp = 5, 6, 512, 32, 8
    n_pos = N + 2
    kw = (torch.randn(B, N, W, generator=g) * 0.02).to(dev)
    count = torch.tensor([6, 1, 4, 0, 9], dtype=torch.int64, device=dev)           # the last one points behind the tensor
    tok = torch.randn(3, W, generator=g).to(dev)
    pos = (torch.randn(77, W, generator=g) * 0.01).to(dev)
    clamped = torch.zeros((), dtype=torch.int64, device=dev)
    X, eot_row = ops.prompt_assemble(kw, count, tok, pos, Bp, SEG, n_pos, clamped)
    torch.cuda.synchronize()
    # the element-wise formulation
    index = count + 1
    t = torch.arange(n_pos, device=dev).unsqueeze(0)
    e = tok[2].expand(B, n_pos, W).clone()
    e[:, 0] = tok[0]
    src = torch.zeros(B, n_pos, W, device=dev)
    src[:, 1: 1 + N] = kw
    e = torch.where(((t >= 1) & (t < index.unsqueeze(1))).unsqueeze(-1), src, e)
    e = torch.where((t == index.unsqueeze(1)).unsqueeze(-1), tok[1].expand(B, n_pos, W), e)
    ref = torch.zeros(Bp, SEG, W, device=dev, dtype=torch.bfloat16)
    ref[:B, :n_pos] = (e + pos[:n_pos]).to(torch.bfloat16)
    assert torch.equal(X.view(Bp, SEG, W), ref)
    want_row = torch.arange(B, device=dev) * SEG + torch.clamp(index, max=n_pos - 1)
    assert torch.equal(eot_row.long(), want_row) and int(clamped) == 1
    # gather / scatter of the selected rows
    rows = ops.rows_gather(X, eot_row)
    assert torch.equal(rows, X[eot_row.long()].float())
    d = torch.randn(B, W, generator=g).to(dev)
    dX = ops.rows_scatter(d, eot_row, Bp * SEG, SEG)
    want = torch.zeros(Bp * SEG, W, device=dev, dtype=torch.bfloat16)
    want[eot_row.long()] = d.to(torch.bfloat16)
    assert torch.equal(dX, want)
    # keyword gradient: row j + 1 of every sample for j < its count (inside the prefix), zero behind
    dXr = torch.randn(Bp * SEG, W, generator=g).to(torch.bfloat16).to(dev)
    dk = ops.prompt_assemble_bwd(dXr, count, B, N, SEG, n_pos)
    live = (torch.arange(N, device=dev).unsqueeze(0) + 1 < torch.clamp(index, max=n_pos).unsqueeze(1)).unsqueeze(-1)
    assert torch.equal(dk, torch.where(live, dXr.view(Bp, SEG, W)[:B, 1: 1 + N].float(), torch.zeros((), device=dev)))


@pytest.mark.gpu
@pytest.mark.parametrize("nseq,heads", [(64, 8), (5, 12), (1, 4)])
def test_short_sequence_attention_kernels_vs_fp32(nseq, heads):
    """sc_attn32_fwd_bf16 / sc_attn32_bwd_bf16 (one wave per (sequence, head) of the text tower's 32-row prompts) against causal
    softmax attention in fp32 on the same bf16 inputs: output within bf16 rounding of the fp32 result, gradients within 1.5e-2 of
    their norm (P and dS enter the matrix instructions as bf16), exact zeros where dout is zero for every later query; and against
    the 128-row flash kernels the tower used before (same inputs, 4 sequences to a block)."""
    from speechclip_plus_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31 + nseq)
    W = heads * 64
    M = nseq * 32
    qkv = (torch.randn(M, 3 * W, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    dout = torch.randn(M, W, generator=g).to(torch.bfloat16).to(dev)
    T = 27
    dout.view(nseq, 32, W)[:, T:] = 0                                   # scratch rows behind the prompt carry no gradient
    scale = 64 ** -0.5
    out = ops.attn32_fwd(qkv, heads, scale)
    dqkv = ops.attn32_bwd(qkv, dout, heads, scale)
    x = qkv.float().view(nseq, 32, 3, heads, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_()      # [3, nseq, heads, 32, 64]
    sc = (x[0] @ x[1].transpose(-1, -2)) * scale
    sc = sc.masked_fill(torch.ones(32, 32, device=dev, dtype=torch.bool).triu(1), float("-inf"))
    ref = (sc.softmax(-1) @ x[2]).permute(0, 2, 1, 3).reshape(M, W)
    ref.backward(dout.float())
    ref = ref.detach()
    dref = x.grad.permute(1, 3, 0, 2, 4).reshape(M, 3 * W)
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(out, ref) < 6e-3, rel(out, ref)
    assert float((out.float() - ref).abs().max()) < 3e-2
    for i, name in enumerate("qkv"):
        a, b = dqkv[:, i * W: (i + 1) * W], dref[:, i * W: (i + 1) * W]
        assert rel(a, b) < 1.5e-2, (name, rel(a, b))
    assert float(dqkv.view(nseq, 32, 3 * W)[:, T:].float().abs().max()) == 0.0
    if M % 128 == 0:
        NB = M // 128
        valid = torch.full((NB,), 128, device=dev, dtype=torch.int32)
        vt = ops.head_transpose(qkv[:, 2 * W:], NB, 128, heads)
        att = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
        lse2 = torch.empty(NB, heads, 128, device=dev, dtype=torch.float32)
        ops.attn_fwd(qkv[:, : 2 * W], vt, valid, att, NB, 128, heads, W, scale, lse2=lse2, causal=32)
        assert rel(out, att.float()) < 6e-3
        d2 = torch.empty_like(dqkv)
        ops.attn_bwd(qkv[:, :W], qkv[:, W: 2 * W], qkv[:, 2 * W:], att, dout, lse2, valid, d2[:, :W], d2[:, W: 2 * W], d2[:, 2 * W:],
                     NB, 128, heads, scale, causal=32, q_rows=128)
        assert rel(dqkv, d2.float()) < 1.5e-2


@pytest.mark.gpu
def test_weight_working_copies_in_one_launch_and_aligned_optimiser_layout():
    """sc_cast_transpose_f32_bf16 = (w.to(bf16), w.to(bf16).t()) exactly, for shapes with partial tiles and strided sources; and
    FlatAdam places every parameter (odd-sized neighbours included) on a 16-byte boundary, leaves the padding at zero through steps
    and accumulates a linear layer's weight / bias gradient straight into its buffer (no AccumulateGrad copy: same values)."""
    from speechclip_plus_amd import ops
    from speechclip_plus_amd.linear_fn import LinearBf16Fn
    from speechclip_plus_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    for N, K in ((768, 768), (2304, 768), (100, 36), (4, 260)):
        w = torch.randn(N, K + 4, generator=g).to(dev)[:, :K]
        y, yT = ops.weight_copies(w)
        assert torch.equal(y, w.to(torch.bfloat16)) and torch.equal(yT, w.to(torch.bfloat16).t().contiguous())
        assert ops.weight_copies(w, "plain")[1] is None and torch.equal(ops.weight_copies(w, "T")[1], yT)
    lin = torch.nn.Linear(256, 256).to(dev)
    odd = torch.nn.Parameter(torch.randn(13, generator=g).to(dev))
    one = torch.nn.Parameter(torch.randn(1, generator=g).to(dev))
    params = [odd, lin.weight, one, lin.bias]
    opt = FlatAdam(params, lr=1e-2, weight_decay=0.1, max_grad_norm=1.0)
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in params)
    assert opt.n == 13 + 256 * 256 + 1 + 256 and opt.size == 16 + 256 * 256 + 4 + 256
    x = torch.randn(128, 256, generator=g).to(dev)
    ref_w = lin.weight.detach().clone().requires_grad_()
    ref_b = lin.bias.detach().clone().requires_grad_()
    (LinearBf16Fn.apply(x, ref_w, ref_b).square().sum() + 0).backward()             # plain tensors: gradients returned
    for _ in range(2):                                                               # twice: accumulation (beta = 1)
        (LinearBf16Fn.apply(x, lin.weight, lin.bias).square().sum() + (odd.sum() + one.sum()) * 0.5).backward()
    assert lin.weight.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * opt.offsets[1]
    assert torch.allclose(lin.weight.grad, 2 * ref_w.grad, rtol=1e-5, atol=1e-5) and torch.allclose(lin.bias.grad, 2 * ref_b.grad, rtol=1e-5, atol=1e-4)
    pad = torch.ones(opt.size, dtype=torch.bool, device=dev)
    for p_, off in zip(opt.params, opt.offsets):
        pad[off: off + p_.numel()] = False
    opt.step()
    opt.zero_grad()
    assert int(pad.sum()) == 3 + 3 and float(opt.flat_p[pad].abs().sum()) == 0.0 and float(opt.m[pad].abs().sum()) == 0.0


@pytest.mark.gpu
def test_cif_row_kernels_match_the_fp32_tensor_kernels():
    """sc_cif_fwd_rows / sc_cif_bwd_rows (bf16 rows at the attention block's pitch, frames behind ``head`` leading rows) against
    sc_cif_fwd / sc_cif_bwd on the fp32 copy of the same frames: identical slots (fp32 accumulation of the same values), gradient =
    the fp32 one rounded to bf16 in the frames' rows and exact zeros in every other row; sc_rows_zero_pad_bf16 on the way."""
    from speechclip_plus_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    for B, P, head, S, C, Tc in ((5, 128, 1, 99, 768, 9), (3, 64, 0, 64, 1024, 5), (4, 192, 1, 150, 256, 12)):
        lead, trail = 8, 8
        flat = torch.randn(lead + B * P + trail, C, generator=g).to(torch.bfloat16).to(dev)
        ref = flat.clone()
        ops.rows_zero_pad(flat, lead, B, P, head, head + S, trail)
        keep = torch.zeros(lead + B * P + trail, dtype=torch.bool, device=dev)
        for b in range(B):
            keep[lead + b * P + head: lead + b * P + head + S] = True
        assert torch.equal(flat[keep], ref[keep]) and float(flat[~keep].float().abs().max()) == 0.0
        full = flat[lead: lead + B * P].view(B, P, C)
        x32 = full[:, head: head + S].float().contiguous()
        alpha = (torch.rand(B, S, generator=g) * (Tc / S) * 1.6).to(dev)
        csum = alpha.cumsum(1).contiguous()
        out = ops.cif_fwd_rows(full, head, S, alpha, csum, Tc, 1.0)
        assert torch.equal(out, ops.cif_fwd(x32, alpha, csum, Tc, 1.0))
        gslots = torch.randn(B, Tc + 1, C, generator=g).to(dev)
        dfull, pa, pb = ops.cif_bwd_rows(full, head, S, alpha, csum, gslots, Tc, 1.0)
        dx32, pa32, pb32 = ops.cif_bwd(x32, alpha, csum, gslots, Tc, 1.0)
        assert torch.equal(dfull[:, head: head + S], dx32.to(torch.bfloat16))
        assert float(dfull[:, :head].float().abs().sum()) == 0.0 and float(dfull[:, head + S:].float().abs().sum()) == 0.0
        assert torch.equal(pa, pa32) and torch.equal(pb, pb32)


@pytest.mark.gpu
def test_cif_slot_and_frame_parallel_kernels_equal_the_sequential_walk():
    """The integrate-and-fire kernels launched per output slot (forward) / per 8 frames (backward) against the one-wave frame walk
    (tuning switch 6): every bit equal, on weights that fire across several slots at once, run into the slot cap T, never fire at
    all, and on widths that leave a partial channel block."""
    from speechclip_plus_amd import ops
    from speechclip_plus_amd._lib import lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(14)
    try:
        for B, S, C, Tc, scale in ((6, 499, 1024, 25, 0.1), (7, 130, 768, 40, 0.5), (3, 50, 256, 8, 2.5), (2, 17, 260, 30, 3.0),
                                   (4, 70, 512, 6, 0.01), (3, 65, 768, 3, 1.0)):
            alpha = (torch.rand(B, S, generator=g) * scale).to(dev)
            alpha[0, S // 2:] = 0.0                                    # an utterance whose second half is padding
            csum = alpha.cumsum(1).contiguous()
            x32 = torch.randn(B, S, C, generator=g).to(dev)
            full = torch.randn(B, S + 13, C, generator=g).to(torch.bfloat16).to(dev)
            gs = torch.randn(B, Tc + 1, C, generator=g).to(dev)
            res = []
            for opt in (1, 0):
                assert lib().sc_set_option(6, opt) == 0
                res.append((ops.cif_fwd(x32, alpha, csum, Tc, 1.0),) + tuple(ops.cif_bwd(x32, alpha, csum, gs, Tc, 1.0))
                           + (ops.cif_fwd_rows(full, 5, S, alpha, csum, Tc, 1.0),) + tuple(ops.cif_bwd_rows(full, 5, S, alpha, csum, gs, Tc, 1.0)))
            for u, v in zip(*res):
                assert torch.equal(u, v)
    finally:
        lib().sc_set_option(6, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("head,k", [(0, 3), (1, 5)])
def test_cif_weight_head_over_resident_rows_matches_the_padded_copy_path(head, k):
    """cif._WeightHeadRowsFn (the weight conv as a strided-row GEMM over the attention block's buffer + the head kernel, one autograd
    node, gradients in the buffer's layout) against the padded-copy formulation (_ConvRowsBf16Fn + _CifHeadFn on the same frames):
    identical alpha, identical parameter gradients and input gradient (the same GEMMs on the same bf16 values)."""
    from speechclip_plus_amd import cif, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    B, P, S, C = 4, 128, 90, 256
    pd = k // 2
    lead = trail = 8
    flat = (torch.randn(lead + B * P + trail, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ops.rows_zero_pad(flat, lead, B, P, head, head + S, trail)
    full = flat[lead: lead + B * P].view(B, P, C)
    conv_w = (torch.randn(C, C, k, generator=g) * (C * k) ** -0.5).to(dev).requires_grad_()
    conv_b = (torch.randn(C, generator=g) * 0.1).to(dev).requires_grad_()
    lin_w = (torch.randn(1, C, generator=g) * C ** -0.5).to(dev).requires_grad_()
    lin_b = torch.zeros(1, device=dev).requires_grad_()
    dalpha = torch.randn(B, S, generator=g).to(dev)
    # rows path
    flat1 = flat.clone()            # the rows as mha_block hands them over: a tensor of its own over the middle of the flat buffer
    f1 = torch.empty(0, device=dev, dtype=torch.bfloat16).set_(flat1.untyped_storage(), lead * C, (B, P, C), (P * C, C, 1)).requires_grad_()
    a1 = cif._WeightHeadRowsFn.apply(f1, conv_w, conv_b, lin_w, lin_b, head, S, pd, 0.0, 0, 0.0, 0)
    a1.backward(dalpha)
    got = [t.grad.clone() for t in (f1, conv_w, conv_b, lin_w, lin_b)]
    for t in (conv_w, conv_b, lin_w, lin_b):
        t.grad = None
    # padded-copy path on the same frames
    x2 = full[:, head: head + S].clone().requires_grad_()
    y_full = cif._ConvRowsBf16Fn.apply(x2, conv_w, conv_b, pd)
    a_full = cif._CifHeadFn.apply(y_full.unsqueeze(0), lin_w, lin_b, 0.0, 0, 0.0, 0)
    a2 = a_full.view(-1)[: B * (S + 2 * pd)].view(B, S + 2 * pd)[:, :S]
    a2.backward(dalpha)
    assert torch.equal(a1, a2)
    assert torch.equal(got[0][:, head: head + S], x2.grad)
    assert float(got[0][:, :head].float().abs().sum()) == 0.0 and float(got[0][:, head + S:].float().abs().sum()) == 0.0
    for a, b in zip(got[1:], (conv_w.grad, conv_b.grad, lin_w.grad, lin_b.grad)):
        assert rel(a, b) < 1e-5, rel(a, b)
