"""GPU parity, kernel by kernel, through the C ABI (speechclip_plus_amd.ops -> libspeechclip_hip.so).

Reference for each kernel = the same op in plain fp32 torch on the same (bf16-rounded) inputs.
Tolerances are stated per test: bf16 outputs carry 2^-8 relative rounding; accumulation is fp32."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _ops():
    from speechclip_plus_amd import ops
    return ops


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-20))


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,act,use_bias,use_res,out_f32", [
    (256, 768, 768, 0, True, False, False),
    (384, 3072, 768, 1, True, False, False),
    (300, 768, 3072, 0, True, True, False),      # M tail
    (128, 512, 1536, 1, False, False, False),
    (130, 200, 64, 0, True, True, True),         # M and N tails, fp32 out
    (1024, 128, 128, 0, False, False, False),
])
def test_gemm(dev, M, N, K, act, use_bias, use_res, out_f32):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    W = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev) if use_bias else None
    res = bf(torch.randn(M, N, generator=g)).to(dev) if use_res else None
    out = ops.linear_bf16(A, W, bias, residual=res, act=act, out_f32=out_f32)
    ref = A.float() @ W.float().T
    if use_bias:
        ref = ref + bias
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + res.float()
    tol = 1e-4 if out_f32 else 6e-3
    err = rel_l2(out, ref)
    assert err < tol, err
    assert float((out.float() - ref).abs().max()) < (1e-3 if out_f32 else 0.06) * (1 + float(ref.abs().max()))


@pytest.mark.parametrize("tile", [2, 7, 8])
@pytest.mark.parametrize("M,N,K,act,use_res", [(512, 768, 768, 0, True), (1024, 512, 1536, 1, False), (700, 2408, 128, 0, True),
                                                (2048, 3072, 768, 1, False), (512, 256, 64, 0, False), (6000, 512, 1024, 1, False)])
def test_gemm_tile256(dev, M, N, K, act, use_res, tile):
    """The 256-row persistent kernel (tile = 2 by cost model, 7 / 8 = forced 256- / 192-wide) against fp32 torch and against the
    128x128 variant, incl. M / N tails, short K."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    W = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = bf(torch.randn(M, N, generator=g)).to(dev) if use_res else None
    out = ops.linear_bf16(A, W, bias, residual=res, act=act, tile=tile)
    ref = A.float() @ W.float().T + bias
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + res.float()
    assert rel_l2(out, ref) < 6e-3, rel_l2(out, ref)
    out1 = ops.linear_bf16(A, W, bias, residual=res, act=act, tile=1)
    assert torch.equal(out, out1), "both tile variants accumulate k in the same order and share the epilogue maths"


@pytest.mark.parametrize("tile", [2, 7, 8])
def test_gemm_tile256_transposed_store(dev, tile):
    ops = _ops()
    B, R, D, H = 2, 384, 768, 12
    g = torch.Generator(device="cpu").manual_seed(19)
    x = bf(torch.randn(B * R, D, generator=g)).to(dev)
    W = bf(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev)
    bias = torch.randn(3 * D, generator=g).to(dev)
    qk = torch.zeros(B * R, 2 * D, device=dev, dtype=torch.bfloat16)
    vt = torch.zeros(B, H, 64, R, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(x, D, W, D, qk, 2 * D, B * R, 3 * D, D, bias=bias, Ct=vt, n_split=2 * D, R=R, dh=64, tile=tile)
    ref = x.float() @ W.float().T + bias
    assert rel_l2(qk, ref[:, : 2 * D]) < 6e-3
    assert rel_l2(vt, ref[:, 2 * D:].view(B, R, H, 64).permute(0, 2, 3, 1)) < 6e-3


def test_gemm_identity_asymmetric(dev):
    """A = I against an asymmetric W catches a transposed C write or a wrong fragment map exactly."""
    ops = _ops()
    K = N = 256
    A = bf(torch.eye(K)).to(dev)
    W = bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).to(dev)
    out = ops.linear_bf16(A, W, out_f32=True)
    assert torch.equal(out, W.float().T.contiguous())


def test_gemm_strided_rows_conv(dev):
    """Overlapping A rows (lda < K): a channels-last Conv1d(k=3, s=2) as one GEMM."""
    ops = _ops()
    C, Tin = 512, 301
    g = torch.Generator(device="cpu").manual_seed(5)
    x = bf(torch.randn(Tin + 8, C, generator=g)).to(dev)
    w = bf(torch.randn(C, C, 3, generator=g) * (3 * C) ** -0.5).to(dev)
    wk = w.permute(0, 2, 1).reshape(C, 3 * C).contiguous()
    Tout = (Tin - 3) // 2 + 1
    out = torch.zeros(Tout, C, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(x, 2 * C, wk, 3 * C, out, C, Tout, C, 3 * C, act=1)
    ref = F.gelu(F.conv1d(x[:Tin].float().T.unsqueeze(0), w.float(), stride=2))[0].T
    assert rel_l2(out, ref) < 6e-3
    # shared-tap K order (sc_gemm_args.tap_c: per channel block tap 0, tap 2, tap 1) on both 256-row tile widths, more rows than
    # one tile: only the fp32 summation order over k changes
    Tin2 = 2 * 1500 + 1
    x2 = bf(torch.randn(Tin2 + 8, C, generator=g)).to(dev)
    Tout2 = (Tin2 - 3) // 2 + 1
    ref2 = F.gelu(F.conv1d(x2[:Tin2].float().T.unsqueeze(0), w.float(), stride=2))[0].T
    for tile in (7, 8):
        o0 = torch.zeros(Tout2, C, device=dev, dtype=torch.bfloat16)
        o1 = torch.zeros(Tout2, C, device=dev, dtype=torch.bfloat16)
        ops.gemm_raw(x2, 2 * C, wk, 3 * C, o0, C, Tout2, C, 3 * C, act=1, tile=tile)
        ops.gemm_raw(x2, 2 * C, wk, 3 * C, o1, C, Tout2, C, 3 * C, act=1, tile=tile, tap_c=C)
        assert rel_l2(o1, ref2) < 6e-3 and rel_l2(o0, ref2) < 6e-3
        assert rel_l2(o1, o0) < 3e-3 and not torch.equal(o1, torch.zeros_like(o1))
    o128 = torch.zeros(Tout2, C, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(x2, 2 * C, wk, 3 * C, o128, C, Tout2, C, 3 * C, act=1, tile=1, tap_c=C)
    assert torch.equal(o128, o1)                                     # same K order in every tile family: bitwise equal
    with pytest.raises(RuntimeError, match="tap_c"):
        ops.gemm_raw(x2, 2 * C, wk, 3 * C, o1, C, Tout2, C, 2 * C, tile=8, tap_c=C)


def test_gemm_transposed_store_and_batch(dev):
    ops = _ops()
    B, R, D, H = 2, 128, 768, 12
    g = torch.Generator(device="cpu").manual_seed(9)
    x = bf(torch.randn(B * R, D, generator=g)).to(dev)
    W = bf(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev)
    bias = torch.randn(3 * D, generator=g).to(dev)
    qk = torch.zeros(B * R, 2 * D, device=dev, dtype=torch.bfloat16)
    vt = torch.zeros(B, H, 64, R, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(x, D, W, D, qk, 2 * D, B * R, 3 * D, D, bias=bias, Ct=vt, n_split=2 * D, R=R, dh=64)
    ref = x.float() @ W.float().T + bias
    assert rel_l2(qk, ref[:, : 2 * D]) < 6e-3
    v_ref = ref[:, 2 * D:].view(B, R, H, 64).permute(0, 2, 3, 1)
    assert rel_l2(vt, v_ref) < 6e-3
    # grouped / batched narrow GEMM (pos_conv shape): N = 48 per group
    G, Dg, Kp, Rr = 4, 48, 128, 128
    Rp = Rr + Kp
    xg = bf(torch.randn(G, B, Rp, Dg, generator=g)).to(dev)
    wg = bf(torch.randn(G, Dg, Kp * Dg, generator=g) * (Kp * Dg) ** -0.5).to(dev)
    bg = torch.randn(G * Dg, generator=g).to(dev)
    resid = bf(torch.randn(B * Rr, G * Dg, generator=g)).to(dev)
    out = torch.zeros(B * Rr, G * Dg, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(xg, Dg, wg, Kp * Dg, out, G * Dg, Rr, Dg, Kp * Dg, bias=bg, residual=resid, ldr=G * Dg, act=1,
                 nb1=G, nb2=B, sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, Rr * G * Dg), sBias=(Dg, 0),
                 sR=(Dg, Rr * G * Dg))
    ref = torch.zeros(B, Rr, G * Dg, device=dev)
    for gi in range(G):
        for b in range(B):
            a = xg[gi, b].float().reshape(-1)
            rows = torch.stack([a[t * Dg: t * Dg + Kp * Dg] for t in range(Rr)])
            ref[b, :, gi * Dg:(gi + 1) * Dg] = F.gelu(rows @ wg[gi].float().T + bg[gi * Dg:(gi + 1) * Dg])
    ref = ref.reshape(B * Rr, G * Dg) + resid.float()
    assert rel_l2(out, ref) < 6e-3


def test_dropout_multiplier(dev):
    """sc_dropout_mult_f32: values in {0, 1 / (1 - p)}, the same mask bits as the bf16 dropout kernel for the same seed, a fresh
    mask per call, reproducible under torch.manual_seed."""
    ops = _ops()
    p, n = 0.1, 64 * 768
    torch.manual_seed(11)
    a = ops.dropout_mult((64, 768), p, dev)
    b = ops.dropout_mult((64, 768), p, dev)
    assert set(torch.unique(a).tolist()) == {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
    assert abs(float((a > 0).float().mean()) - (1 - p)) < 0.01 and not torch.equal(a, b)
    # same bits as sc_dropout_bf16 on a tensor of ones with the seed the call used
    calls = ops._mult_calls[0]
    seed = ((torch.initial_seed() * 0x9E3779B1) ^ (calls * 0x85EBCA6B)) & 0xffffffff
    ones = torch.ones(64, 768, device=dev, dtype=torch.bfloat16)
    ref = ops.dropout_bf16(ones, p, seed)
    assert torch.equal(ref > 0, b > 0)


@pytest.mark.parametrize("D,G,R,B", [(768, 16, 512, 3), (768, 16, 256, 2), (768, 16, 640, 2), (1024, 16, 384, 2)])
def test_posconv_slab_kernel(dev, D, G, R, B):
    """sc_posconv_bf16 (input slab resident in LDS, weights streamed) against fp32 grouped Conv1d (fairseq pos_conv + SamePad + GELU +
    residual) and, bit for bit, against the sc_gemm_bf16 formulation it replaces; full and partial 512-frame blocks, Dg = 48 and 64."""
    ops = _ops()
    Kp, Dg = 128, D // G
    halo = Kp // 2
    Rp = R + 2 * halo
    g = torch.Generator(device="cpu").manual_seed(D + R)
    x = bf(torch.randn(B, R, D, generator=g)).to(dev)
    lens = [R, max(1, R - 37), 5][:B]
    for b, n in enumerate(lens):
        x[b, n:] = 0                                                       # padded frames are zero (posconv_prep's contract)
    wconv = bf(torch.randn(D, Dg, Kp, generator=g) * (Dg * Kp) ** -0.5).to(dev)   # Conv1d weight [out, in / groups, k]
    bias = torch.randn(D, generator=g).to(dev)
    valid = torch.tensor(lens, dtype=torch.int32, device=dev)
    xz = torch.empty(B * R, D, device=dev, dtype=torch.bfloat16)
    xg = torch.zeros(G, B, Rp, Dg, device=dev, dtype=torch.bfloat16)
    ops.posconv_prep(x.view(B * R, D), valid, xz, xg, B, R, D, G, halo)
    w = wconv.reshape(G, Dg, Dg, Kp).permute(0, 1, 3, 2).reshape(G, Dg, Kp * Dg).contiguous()     # tap-major per group
    out = torch.full((B * R, D), 7.0, device=dev, dtype=torch.bfloat16)
    ops.posconv(xg, w, bias, xz, out, B, R, D, G, Kp)
    ref = F.conv1d(x.float().transpose(1, 2), wconv.float(), bias, padding=halo, groups=G)[:, :, :R].transpose(1, 2)
    ref = (F.gelu(ref) + x.float()).reshape(B * R, D)
    assert rel_l2(out, ref) < 6e-3, rel_l2(out, ref)
    out_g = torch.zeros(B * R, D, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(xg, Dg, w, Kp * Dg, out_g, D, R, Dg, Kp * Dg, bias=bias, residual=xz, ldr=D, act=1, nb1=G, nb2=B,
                 sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, R * D), sBias=(Dg, 0), sR=(Dg, R * D))
    assert torch.equal(out, out_g), "same k order, bias-initialised accumulators, GELU and rounding as the GEMM formulation"


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("R,lens", [(128, [128, 1, 37]), (256, [200, 256, 65]), (512, [499, 300, 33])])
def test_attention(dev, R, lens):
    ops = _ops()
    B, H, D = len(lens), 12, 768
    g = torch.Generator(device="cpu").manual_seed(R)
    q = bf(torch.randn(B, R, D, generator=g)).to(dev)
    k = bf(torch.randn(B, R, D, generator=g)).to(dev)
    v = bf(torch.randn(B, R, D, generator=g)).to(dev)
    # force one online-softmax rescale late in the sequence (rule 26: exercise the rescale branch)
    k[0, min(lens[0], R) - 1, :64] = 6.0 * q[0, 5, :64]
    qk = torch.cat([q, k], dim=-1).reshape(B * R, 2 * D).contiguous()
    vt = v.view(B, R, H, 64).permute(0, 2, 3, 1).contiguous()
    valid = torch.tensor(lens, dtype=torch.int32, device=dev)
    out = torch.zeros(B * R, D, device=dev, dtype=torch.bfloat16)
    scale = 64 ** -0.5
    ops.attn_fwd(qk, vt, valid, out, B, R, H, D, scale)
    qf = q.float().view(B, R, H, 64).transpose(1, 2) * scale
    kf = k.float().view(B, R, H, 64).transpose(1, 2)
    vf = v.float().view(B, R, H, 64).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2)
    mask = torch.arange(R, device=dev)[None, :] >= valid[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B * R, D)
    err = rel_l2(out, ref)
    assert err < 1.5e-2, err
    assert float((out.float() - ref).abs().max()) < 0.08


# ------------------------------------------------------------------------------------------------ row ops
@pytest.mark.parametrize("D", [512, 768, 1024])
def test_layernorm(dev, D):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(D)
    x = bf(torch.randn(301, D, generator=g) * 3 + 0.5).to(dev)
    gam = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    bet = (0.1 * torch.randn(D, generator=g)).to(dev)
    y = ops.layernorm_bf16(x, gam, bet)
    ref = F.layer_norm(x.float(), (D,), gam, bet, 1e-5)
    assert rel_l2(y, ref) < 4e-3
    y2 = ops.layernorm_bf16(x, gam, bet, act=1)
    assert rel_l2(y2, F.gelu(ref)) < 5e-3


def test_weighted_sum_golden(dev, golden):
    """WeightedSumLayer against the reference leaf's output + grad (tests/golden/wsum.npz)."""
    from speechclip_plus_amd.weighted_sum import WeightedSumLayer
    fx = golden("wsum.npz")
    layer = WeightedSumLayer(13).to(dev)
    with torch.no_grad():
        layer.weights.copy_(torch.from_numpy(fx["weights"]))
    hs = [torch.from_numpy(h).to(dev) for h in fx["hs"]]
    out = layer(hs)
    assert rel_l2(out, torch.from_numpy(fx["out"]).to(dev)) < 8e-3          # bf16 inputs + output
    (out.float() * torch.from_numpy(fx["gout"]).to(dev)).sum().backward()
    np.testing.assert_allclose(layer.weights.grad.cpu().numpy(), fx["dweights"], rtol=0.05, atol=0.02)


def test_weighted_sum_over_segments_fixed_layer_count_kernel(dev):
    """Round 5: sc_wsum_fwd_seg takes a one-chunk-per-thread kernel with the layer count as a compile-time constant for NL = 13 / 25 (base /
    large).  It must compute what the generic kernel computes (sc_set_option(5, 1) selects the generic one): the same fp32 multiply-adds
    in the same order - the compiler's contraction choices may differ, so at most one bf16 ulp on a few elements per million - and both
    within half an ulp + fp32 round-off of the exact sum; every row of the uniform output written, zeros outside the utterances."""
    from speechclip_plus_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(12)
    B, T, D, R, off = 6, 77, 768, 80, 1
    for NL in (13, 25, 7):                       # 7: no fixed instance, both calls take the generic kernel
        lens = [T, 5, 40, 77, 1, 63]
        pitch = [(l + 1 + 7) // 8 * 8 for l in lens]
        seg = ops.RowSegments(pitch, lens, dev)
        h = bf(torch.randn(NL, seg.rows, D, generator=g)).to(dev)
        w = torch.softmax(torch.randn(NL, generator=g), 0).to(dev)
        outs = []
        try:
            for opt in (0, 1):
                _lib.lib().sc_set_option(5, opt)
                out = torch.full((B, R, D), 7.0, device=dev, dtype=torch.bfloat16)
                ops.wsum_fwd(h, w, out, B, R, D, off, seg=seg)
                outs.append(out.float())
        finally:
            _lib.lib().sc_set_option(5, 0)
        ref = torch.zeros(B, R, D, device=dev, dtype=torch.float64)
        r0 = 0
        for b in range(B):
            n = min(pitch[b], R - off)
            ref[b, off: off + n] = (w.view(-1, 1, 1).double() * h[:, r0: r0 + n].double()).sum(0)
            r0 += pitch[b]
        ulp = torch.clamp(ref.abs(), min=2.0 ** -126).log2().floor().exp2() * 2.0 ** -7          # one bf16 ulp at the value
        for o in outs:
            assert bool(((o.double() - ref).abs() <= 0.5 * ulp + 1e-5).all()), NL
            assert float(o[:, :off].abs().max()) == 0.0
        d = (outs[0] - outs[1]).abs().double()
        assert bool((d <= ulp).all()) and float((d > 0).double().mean()) < 1e-3, (NL, float(d.max()))
        if NL == 7:
            assert torch.equal(outs[0], outs[1])


def test_weighted_sum_normalized(dev, golden):
    """normalize_features=True (HuBERT-large recipes): forward against the reference leaf (wsum.npz out_norm), weight
    gradient against fp32 torch autograd on the same bf16-rounded inputs; also at D = 1024."""
    from speechclip_plus_amd.weighted_sum import WeightedSumLayer
    fx = golden("wsum.npz")
    layer = WeightedSumLayer(13, normalize_features=True).to(dev)
    with torch.no_grad():
        layer.weights.copy_(torch.from_numpy(fx["weights"]))
    hs = [torch.from_numpy(h).to(dev) for h in fx["hs"]]
    out = layer(hs)
    assert rel_l2(out, torch.from_numpy(fx["out_norm"]).to(dev)) < 1e-2
    g = torch.Generator(device="cpu").manual_seed(4)
    NL, M, D = 25, 200, 1024
    h = bf(torch.randn(NL, M, D, generator=g) * 2 + 0.3).to(dev)
    layer2 = WeightedSumLayer(NL, normalize_features=True).to(dev)
    with torch.no_grad():
        layer2.weights.copy_(torch.randn(NL, generator=g))
    out2 = layer2([h[n].view(2, 100, D) for n in range(NL)])
    gout = torch.randn(2, 100, D, generator=g).to(dev)
    (out2.float() * gout).sum().backward()
    w_ref = layer2.weights.detach().clone().requires_grad_(True)
    ref = (torch.softmax(w_ref, 0).view(-1, 1, 1) * F.layer_norm(h.float(), (D,))).sum(0).view(2, 100, D)
    (ref * gout).sum().backward()
    assert rel_l2(out2, ref) < 6e-3
    assert rel_l2(layer2.weights.grad, w_ref.grad) < 2e-3


def test_frontend_conv0(dev):
    ops = _ops()
    B, L, C = 3, 4000, 512
    g = torch.Generator(device="cpu").manual_seed(1)
    wav = torch.randn(B, L, generator=g)
    lens = torch.tensor([4000, 2500, 801])
    wav = wav * (torch.arange(L)[None] < lens[:, None])
    w0 = torch.randn(C, 10, generator=g) * 0.3
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    T0 = (L - 10) // 5 + 1
    R0 = 1024
    ldw = 5 * R0 + 64
    wav_pad = torch.zeros(B, ldw, device=dev)
    ops.wav_prep(wav.to(dev), lens.to(dev), wav_pad, False)
    assert torch.equal(wav_pad[:, :L].cpu(), wav)
    out = torch.zeros(B * R0, C, device=dev, dtype=torch.bfloat16)
    ops.conv0_groupnorm_gelu(wav_pad, w0.to(dev), gam.to(dev), bet.to(dev), T0, R0, out)
    y = F.conv1d(wav.unsqueeze(1), w0.unsqueeze(1), stride=5)
    ref = F.gelu(F.group_norm(y, C, gam, bet, 1e-5)).transpose(1, 2)           # (B, T0, C)
    got = out.view(B, R0, C)[:, :T0].float().cpu()
    assert rel_l2(got, ref) < 5e-3
    # utterance layer-norm variant of wav_prep (fairseq task.cfg.normalize)
    ops.wav_prep(wav.to(dev), lens.to(dev), wav_pad, True)
    for b in range(B):
        n = int(lens[b])
        ref_n = F.layer_norm(wav[b, :n], (n,))
        np.testing.assert_allclose(wav_pad[b, :n].cpu().numpy(), ref_n.numpy(), rtol=1e-4, atol=1e-5)
        assert float(wav_pad[b, n:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("name", ["loss_b8", "loss_b32_dup", "loss_b32_dup_trainT", "loss_b256_dup", "loss_b512_cap_lifted"])
def test_loss_golden(dev, golden, name):
    """MaskedContrastiveLoss (HIP) against the reference leaf's loss / dA / dB / dtemperature."""
    from speechclip_plus_amd.losses import MaskedContrastiveLoss
    fx = golden(name + ".npz")
    trainT = "temp_param" in fx
    crit = MaskedContrastiveLoss(temperature=0.07, temperature_trainable=trainT).to(dev)
    A = torch.from_numpy(fx["A"]).to(dev).requires_grad_(True)
    Bm = torch.from_numpy(fx["B"]).to(dev).requires_grad_(True)
    ids = torch.from_numpy(fx["ids"]).to(dev)
    loss = crit(A, Bm, ids)
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"])))
    np.testing.assert_allclose(A.grad.cpu().numpy(), fx["dA"], rtol=2e-4, atol=2e-6)
    if "dB" in fx:
        np.testing.assert_allclose(Bm.grad.cpu().numpy(), fx["dB"], rtol=2e-4, atol=2e-6)
    if trainT:
        np.testing.assert_allclose(crit.temperature.grad.cpu().numpy(), fx["dtemp_param"], rtol=2e-4)
    if "loss_noindex" in fx:
        l2 = crit(A.detach(), Bm.detach(), None)
        assert abs(l2.item() - float(fx["loss_noindex"])) < 2e-5 * max(1.0, abs(float(fx["loss_noindex"])))


@pytest.mark.parametrize("name", ["loss_v_margin", "loss_v_dcl", "loss_v_a2b", "loss_v_b2a", "loss_v_margin_dcl_trainT"])
def test_loss_variants_golden(dev, golden, name):
    """margin / decoupled / one-sided MaskedContrastiveLoss (losses.py:213,226-245) against the reference leaf's vectors; two
    launches back to back share the workspace (the ticket word must come back to zero)."""
    from speechclip_plus_amd.losses import MaskedContrastiveLoss
    fx = golden(name + ".npz")
    crit = MaskedContrastiveLoss(temperature=0.07, temperature_trainable=bool(fx["trainT"]), margin=float(fx["margin"]),
                                 dcl=bool(fx["dcl"]), a2b=bool(fx["a2b"]), b2a=bool(fx["b2a"])).to(dev)
    A = torch.from_numpy(fx["A"]).to(dev).requires_grad_(True)
    Bm = torch.from_numpy(fx["B"]).to(dev).requires_grad_(True)
    ids = torch.from_numpy(fx["ids"]).to(dev)
    for _ in range(2):
        A.grad = Bm.grad = None
        crit.zero_grad()
        loss = crit(A, Bm, ids)
        loss.backward()
        assert abs(loss.item() - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"])))
        np.testing.assert_allclose(A.grad.cpu().numpy(), fx["dA"], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(Bm.grad.cpu().numpy(), fx["dB"], rtol=2e-4, atol=2e-6)
        if bool(fx["trainT"]):
            np.testing.assert_allclose(crit.temperature.grad.cpu().numpy(), fx["dtemp_param"], rtol=2e-4)
    l2 = crit(A.detach(), Bm.detach(), None)
    assert abs(l2.item() - float(fx["loss_noindex"])) < 2e-5 * max(1.0, abs(float(fx["loss_noindex"])))


@pytest.mark.parametrize("Bg,E", [(64, 512), (200, 768), (512, 512), (1000, 64)])
def test_infonce_fused_forward_vs_fp64(dev, Bg, E):
    """sc_infonce_fwd at the batch sizes of the recipes (64 per GPU ... 512 global, and a ragged one): logits, both
    log-sum-exps and the loss against fp64 torch."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(Bg)
    A = F.normalize(torch.randn(Bg, E, generator=g), dim=-1)
    Bm = F.normalize(torch.randn(Bg, E, generator=g) + 0.5 * A, dim=-1)
    ids = torch.arange(Bg) // 5
    it = torch.full((1,), 1 / 0.07)
    loss, logits, lr, lc = ops.infonce_fwd(A.to(dev), Bm.to(dev), ids.to(dev), it.to(dev))
    ref = (A.double() @ Bm.double().t()) / 0.07
    neg = (ids[:, None] != ids[None, :]) | torch.eye(Bg, dtype=torch.bool)
    rl = torch.logsumexp(ref.masked_fill(~neg, float("-inf")), dim=1)
    cl = torch.logsumexp(ref.masked_fill(~neg, float("-inf")), dim=0)
    ref_loss = 0.5 * ((rl - ref.diag()).mean() + (cl - ref.diag()).mean())
    assert float((logits.double().cpu() - ref).abs().max()) < 2e-5
    assert float((lr.double().cpu() - rl).abs().max()) < 5e-5 and float((lc.double().cpu() - cl).abs().max()) < 5e-5
    assert abs(float(loss) - float(ref_loss)) < 2e-5


# ------------------------------------------------------------------------------------------------ head
@pytest.mark.parametrize("name", ["head_d64_h8", "head_d64_h1"])
def test_parallel_branch_golden(dev, golden, name):
    """KW_ParallelBranch (CLS pooling kernels + fp32 tail) against the reference leaf composition
    (tests/golden/head_*.npz): output, grad wrt features, grads of every parameter."""
    from speechclip_plus_amd import Config, KW_ParallelBranch
    fx = golden(name + ".npz")
    nhead = int(fx["nhead"])
    D, Fd, E = 64, 128, 24
    cfg = Config({"model_settings": {"parallel_branch": {
        "transformer_type": "TransformerEncoder",
        "transformer_args": {"n_layers": 1, "d_model": D, "nhead": nhead, "dim_feedforward": Fd, "dropout": 0.1,
                             "activation": "gelu", "layer_norm_eps": 1e-5, "batch_first": True, "norm_first": False},
        "need_projection": True}}})
    br = KW_ParallelBranch(cfg, audio_dim=D, text_dim=E).to(dev).eval()
    sd = {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("W_")}
    missing = br.load_state_dict(sd, strict=True)
    feat = torch.from_numpy(fx["feat"]).to(dev).requires_grad_(True)
    out = br(audio_feat=feat, audio_feat_len=torch.from_numpy(fx["audio_len"]).to(dev))["parallel_audio_feat"]
    # features enter the kernels as bf16 -> 2^-8 relative input rounding
    assert rel_l2(out, torch.from_numpy(fx["out"]).to(dev)) < 1e-2
    (out * torch.from_numpy(fx["gout"]).to(dev)).sum().backward()
    assert rel_l2(feat.grad, torch.from_numpy(fx["g_feat"]).to(dev)) < 3e-2
    assert rel_l2(br.cls.grad, torch.from_numpy(fx["g_cls"]).to(dev)) < 3e-2
    for n, p in br.named_parameters():
        key = "g_" + n
        if key in fx and n != "cls":
            ref = torch.from_numpy(fx[key]).to(dev)
            if float(ref.norm()) < 1e-6:          # in_proj_bias k-part: exactly zero gradient
                assert float(p.grad.norm()) < 1e-4, n
            else:
                assert rel_l2(p.grad, ref) < 3e-2, (n, rel_l2(p.grad, ref))


def test_parallel_branch_train_mode_dropout_vs_oracle(dev, golden):
    """Train mode of the CLS head (the four dropout sites of nn.TransformerEncoderLayer, p = 0.1 in the recipes; p = 0.3 here)
    against the oracle's train-mode restatement fed the SAME masks: stateless hash masks drawn in a fixed order
    (head_tail.ParallelHeadFn -> ops.dropout_mult), a function of torch's seed and a call counter, so rewinding the counter
    reproduces them.  Output, feature / cls / parameter gradients."""
    import oracle
    from conftest import weights_from
    from speechclip_plus_amd import Config, KW_ParallelBranch
    fx = golden("head_d64_h8.npz")
    nhead, D, Fd, E, pd = int(fx["nhead"]), 64, 128, 24, 0.3
    cfg = Config({"model_settings": {"parallel_branch": {
        "transformer_type": "TransformerEncoder",
        "transformer_args": {"n_layers": 1, "d_model": D, "nhead": nhead, "dim_feedforward": Fd, "dropout": pd,
                             "activation": "gelu", "layer_norm_eps": 1e-5, "batch_first": True, "norm_first": False},
        "need_projection": True}}})
    br = KW_ParallelBranch(cfg, audio_dim=D, text_dim=E).to(dev).train()
    br.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("W_")}, strict=True)
    feat = torch.from_numpy(fx["feat"]).to(dev).requires_grad_(True)
    alen = torch.from_numpy(fx["audio_len"])
    B, T = feat.shape[:2]
    R = (T + 1 + 127) // 128 * 128
    torch.manual_seed(77)
    ops = _ops()
    calls0 = ops._mult_calls[0]                    # the masks are a function of torch's seed and this call counter (ops.dropout_mult)
    out = br(audio_feat=feat, audio_feat_len=alen.to(dev))["parallel_audio_feat"]
    gout = torch.from_numpy(fx["gout"]).to(dev)
    (out * gout).sum().backward()
    ops._mult_calls[0] = calls0
    mk = lambda *shape: ops.dropout_mult(shape, pd, dev).cpu()
    mult, k1, kf, k2 = mk(B, nhead, R), mk(B, D), mk(B, Fd), mk(B, D)
    assert 0.5 < float((mult > 0).float().mean()) < 0.9
    row0 = {"dropout1": k1, "dropout": kf, "dropout2": k2}

    def drop(site, layer, t):
        m = torch.ones_like(t)
        if site == "attn":
            m[:, :, 0, :] = mult[:, :, : t.shape[-1]]
        else:
            m[:, 0] = row0[site]
        return t * m

    W = {k: v.clone().requires_grad_(True) for k, v in weights_from(fx).items()}
    f_ref = torch.from_numpy(fx["feat"]).bfloat16().float().requires_grad_(True)      # the kernels read the features as bf16
    ref = oracle.parallel_branch_forward(W, f_ref, alen, nhead=nhead, drop=drop)
    (ref * gout.cpu()).sum().backward()
    assert rel_l2(out, ref.to(dev)) < 1e-2
    eval_out = oracle.parallel_branch_forward(W, f_ref, alen, nhead=nhead)
    assert rel_l2(ref, eval_out) > 0.1                       # the masks matter: train mode is far from the eval output
    assert rel_l2(feat.grad, f_ref.grad.to(dev)) < 3e-2
    assert rel_l2(br.cls.grad, W["cls"].grad.to(dev)) < 3e-2
    for n, p_ in br.named_parameters():
        if n == "cls" or W[n].grad is None:
            continue
        r = W[n].grad.to(dev)
        if float(r.norm()) < 1e-6:
            assert float(p_.grad.norm()) < 1e-4, n
        else:
            assert rel_l2(p_.grad, r) < 3e-2, (n, rel_l2(p_.grad, r))
    # eval mode is untouched by the train-mode path
    br.eval()
    with torch.no_grad():
        o2 = br(audio_feat=feat.detach(), audio_feat_len=alen.to(dev))["parallel_audio_feat"]
    assert rel_l2(o2, torch.from_numpy(fx["out"]).to(dev)) < 1e-2


def test_cls_pool_vs_torch(dev):
    """CLS pooling kernels at the shipped width (D=768, H=8, R=512) vs the same math in fp32 torch."""
    ops = _ops()
    B, R, D, H = 3, 512, 768, 8
    g = torch.Generator(device="cpu").manual_seed(3)
    X = bf(torch.randn(B, R, D, generator=g)).to(dev)
    a = (torch.randn(H, D, generator=g) * 0.05).to(dev)
    lens = torch.tensor([500, 1, 77], dtype=torch.int32, device=dev)
    scores = ops.cls_scores(X, a, False, B, R, D, H)
    p, m = ops.cls_pool_fwd(X, scores, lens, B, R, D, H)
    Xf = X.float().requires_grad_(True)
    af = a.clone().requires_grad_(True)
    s = torch.einsum("brd,hd->bhr", Xf, af)
    mask = torch.arange(R, device=dev)[None] >= lens[:, None]
    s = s.masked_fill(mask[:, None], float("-inf"))
    pr = torch.softmax(s, -1)
    mr = torch.einsum("bhr,brd->bhd", pr, Xf)
    assert rel_l2(p, pr) < 1e-4 and rel_l2(m, mr) < 1e-4
    dm = torch.randn(B, H, D, generator=g).to(dev)
    (mr * dm).sum().backward()
    dp = ops.cls_scores(X, dm.contiguous(), True, B, R, D, H)
    dX, da_part = ops.cls_pool_bwd(X, p, dp, dm, a, lens, B, R, D, H)
    assert rel_l2(dX, Xf.grad) < 1e-4
    assert rel_l2(da_part.sum(0), af.grad) < 1e-4


# ------------------------------------------------------------------------------------------------ optimiser
def test_flat_adam_vs_torch(dev):
    from speechclip_plus_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(300, 70, device=dev)), torch.nn.Parameter(torch.randn(513, device=dev))]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref = torch.optim.Adam(rs, lr=1e-3, weight_decay=1e-2)
    opt = FlatAdam(ps, lr=1e-3, weight_decay=1e-2, max_grad_norm=4.0)
    for step in range(5):
        opt.zero_grad()
        ref.zero_grad()
        gs = [torch.randn_like(p) * (3.0 if step % 2 else 0.01) for p in ps]
        for p, r, g in zip(ps, rs, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(rs, 4.0)
        ref.step()
        opt.step()
        for p, r in zip(ps, rs):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


# ---- fp32 head-tail kernels (csrc/headtail.hip) against plain torch fp32 ---------------------------------------------------
def test_sgemm_ex_batched_strided_accumulate(dev):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(3)
    Bn, H, D, dh = 37, 8, 96, 12
    m = torch.randn(Bn, H, D, generator=g).to(dev)
    Wv = torch.randn(D, D, generator=g).to(dev)
    bv = torch.randn(D, generator=g).to(dev)
    cx = torch.empty(Bn, D, device=dev)
    ops.sgemm_ex(m, (H * D, 1, D), Wv, (D, 1, dh * D), cx, D, Bn, dh, D, nbatch=H, scz=dh, bias=bv, sbiasz=dh)
    ref = torch.einsum("bhk,hdk->bhd", m, Wv.view(H, dh, D)).reshape(Bn, D) + bv
    assert torch.allclose(cx, ref, rtol=1e-4, atol=1e-4)
    # weight gradient with accumulate: gW[h*dh+d, k] = 0.5 gW + 2 sum_b dcx[b, h*dh+d] m[b, h, k]
    dcx = torch.randn(Bn, D, generator=g).to(dev)
    gW0 = torch.randn(D, D, generator=g).to(dev)
    gW = gW0.clone()
    ops.sgemm_ex(dcx, (1, D, dh), m, (1, H * D, D), gW, D, dh, D, Bn, nbatch=H, scz=dh * D, alpha=2.0, beta=0.5)
    ref = 0.5 * gW0 + 2.0 * torch.einsum("bhd,bhk->hdk", dcx.view(Bn, H, dh), m).reshape(D, D)
    assert torch.allclose(gW, ref, rtol=1e-4, atol=1e-3)
    # few tiles + long K: the split-K path (partials through the workspace, ordered reduction), with bias / alpha / beta
    x, W2, b2 = torch.randn(Bn, 3000, generator=g).to(dev), torch.randn(70, 3000, generator=g).to(dev), torch.randn(70, generator=g).to(dev)
    y0 = torch.randn(Bn, 70, generator=g).to(dev)
    y = y0.clone()
    ops.sgemm_ex(x, (3000, 1, 0), W2, (3000, 1, 0), y, 70, Bn, 70, 3000, alpha=0.5, beta=2.0, bias=b2)
    assert torch.allclose(y, 2.0 * y0 + 0.5 * (x @ W2.T) + b2, rtol=1e-4, atol=1e-3)
    cx2 = torch.empty(Bn, D, device=dev)
    m2, Wv2 = torch.randn(Bn, H, 768, generator=g).to(dev), torch.randn(D, 768, generator=g).to(dev)
    ops.sgemm_ex(m2, (H * 768, 1, 768), Wv2, (768, 1, dh * 768), cx2, D, Bn, dh, 768, nbatch=H, scz=dh, bias=bv, sbiasz=dh)
    assert torch.allclose(cx2, torch.einsum("bhk,hdk->bhd", m2, Wv2.view(H, dh, 768)).reshape(Bn, D) + bv, rtol=1e-4, atol=1e-3)
    # transposed operand: dx = dy W
    dy, W = torch.randn(Bn, 50, generator=g).to(dev), torch.randn(50, 70, generator=g).to(dev)
    dx = torch.empty(Bn, 70, device=dev)
    ops.sgemm_ex(dy, (50, 1, 0), W, (1, 70, 0), dx, 70, Bn, 70, 50)
    assert torch.allclose(dx, dy @ W, rtol=1e-4, atol=1e-4)


def test_rowln_gelu_colsum_headmask(dev):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(4)
    rows, D = 29, 200
    x = torch.randn(rows, D, generator=g).to(dev).requires_grad_()
    res = torch.randn(1, D, generator=g).to(dev)
    gam, bet = (torch.randn(D, generator=g).to(dev) + 1.0).requires_grad_(), torch.randn(D, generator=g).to(dev).requires_grad_()
    y, xhat, rstd = ops.rowln_fwd(x.detach(), res, 0, gam.detach(), bet.detach(), 1e-5)
    ref = F.layer_norm(x + res, (D,), gam, bet, 1e-5)
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5)
    dy = torch.randn(rows, D, generator=g).to(dev)
    ref.backward(dy)
    dg, db = torch.ones(D, device=dev), torch.full((D,), 2.0, device=dev)        # accumulate on top of existing values
    dx = ops.rowln_bwd(dy, xhat, gam.detach(), rstd, dg, db)
    assert torch.allclose(dx, x.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(dg - 1.0, gam.grad, rtol=1e-4, atol=1e-4) and torch.allclose(db - 2.0, bet.grad, rtol=1e-4, atol=1e-4)
    # per-row residual
    res2 = torch.randn(rows, D, generator=g).to(dev)
    y2, _, _ = ops.rowln_fwd(x.detach(), res2, D, gam.detach(), bet.detach(), 1e-5)
    assert torch.allclose(y2, F.layer_norm(x.detach() + res2, (D,), gam.detach(), bet.detach(), 1e-5), rtol=1e-4, atol=1e-5)
    # gelu fwd / bwd
    u = (torch.randn(rows, D, generator=g) * 2).to(dev).requires_grad_()
    f = ops.gelu_f32(u.detach())
    fr = F.gelu(u)
    assert torch.allclose(f, fr, rtol=1e-5, atol=1e-6)
    fr.backward(dy)
    assert torch.allclose(ops.gelu_f32(u.detach(), dy), u.grad, rtol=1e-4, atol=1e-6)
    # column sums with scale / accumulate, head mask scatter / gather
    out = torch.ones(D, device=dev)
    ops.colsum(dy, D, rows, D, out, alpha=0.5, beta=2.0)
    assert torch.allclose(out, 2.0 + 0.5 * dy.sum(0), rtol=1e-5, atol=1e-5)
    H, dh = 8, 25
    q = torch.randn(1, D, generator=g).to(dev)
    Qm = torch.empty(H, D, device=dev)
    ops.headmask(q, Qm, H, D, dh, gather=False)
    ref = torch.zeros(H, D, device=dev)
    for h in range(H):
        ref[h, h * dh: (h + 1) * dh] = q[0, h * dh: (h + 1) * dh]
    assert torch.equal(Qm, ref)
    back = torch.empty(1, D, device=dev)
    ops.headmask(back, torch.arange(H * D, device=dev, dtype=torch.float32).view(H, D), H, D, dh, gather=True)
    assert torch.equal(back[0], torch.tensor([float((j // dh) * D + j) for j in range(D)], device=dev))


# ---- attention forward (LSE, causal) and backward against fp32 torch ----------------------------------------------------------
@pytest.mark.parametrize("causal", [False, True])
def test_attention_backward(dev, causal):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(21 + causal)
    B, H, R = 2, 3, 256
    D = H * 64
    valid = [200, 256] if not causal else [77, 256]
    qkv = bf(torch.randn(B * R, 3 * D, generator=g)).to(dev)
    q, k, v = qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:]
    vl = torch.tensor(valid, dtype=torch.int32, device=dev)
    vt = ops.head_transpose(v, B, R, H)
    assert torch.equal(vt, v.view(B, R, H, 64).permute(0, 2, 3, 1).contiguous())
    out = torch.zeros(B * R, D, device=dev, dtype=torch.bfloat16)
    lse2 = torch.empty(B, H, R, device=dev, dtype=torch.float32)
    scale = 64 ** -0.5
    ops.attn_fwd(qkv[:, : 2 * D], vt, vl, out, B, R, H, D, scale, lse2=lse2, causal=causal)
    # fp32 reference with autograd
    qf, kf, vf = (t.float().view(B, R, H, 64).transpose(1, 2).detach().requires_grad_() for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    key_ok = torch.arange(R, device=dev)[None, :] < vl[:, None]                       # (B, R)
    mask = key_ok[:, None, None, :].expand(B, H, R, R).clone()
    if causal:
        mask &= torch.tril(torch.ones(R, R, dtype=torch.bool, device=dev))[None, None]
    s = s.masked_fill(~mask, float("-inf"))
    o_ref = torch.softmax(s, dim=-1) @ vf                                           # (B, H, R, 64)
    o_ref_rows = o_ref.transpose(1, 2).reshape(B * R, D)
    qrow_ok = key_ok.reshape(B * R)                                                  # only valid queries matter
    assert rel_l2(out[qrow_ok], o_ref_rows[qrow_ok]) < 1e-2
    lse_ref = torch.logsumexp(s, dim=-1) * 1.4426950408889634                        # log2 domain
    assert torch.allclose(lse2.transpose(0, 1)[:, key_ok], lse_ref.transpose(0, 1)[:, key_ok], rtol=1e-3, atol=2e-2)
    dout = bf(torch.randn(B * R, D, generator=g)).to(dev)
    dout[~qrow_ok] = 0                                                               # padded queries carry no gradient
    o_ref_rows.backward(dout.float())
    dqkv = torch.full((B * R, 3 * D), 7.0, device=dev, dtype=torch.bfloat16)
    ops.attn_bwd(q, k, v, out, dout, lse2, vl, dqkv[:, :D], dqkv[:, D: 2 * D], dqkv[:, 2 * D:], B, R, H, scale, causal=causal)
    ref = [t.grad.transpose(1, 2).reshape(B * R, D) for t in (qf, kf, vf)]
    errs = [rel_l2(dqkv[qrow_ok, i * D: (i + 1) * D], ref[i][qrow_ok]) for i in range(3)]
    assert max(errs) < 2e-2, errs
    # padded keys receive exactly zero
    assert float(dqkv[~qrow_ok][:, D:].abs().max()) == 0.0


def test_layernorm_bwd_and_act(dev):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(9)
    rows, D = 1000, 768
    x = bf(torch.randn(rows, D, generator=g) * 2 + 0.3).to(dev)
    dy = bf(torch.randn(rows, D, generator=g)).to(dev)
    dres = bf(torch.randn(rows, D, generator=g)).to(dev)
    gam = (torch.randn(D, generator=g) * 0.2 + 1).to(dev)
    xr = x.float().requires_grad_()
    gr = gam.clone().requires_grad_()
    br = torch.zeros(D, device=dev, requires_grad=True)
    F.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy.float())
    dx, dg, db = ops.layernorm_bwd(x, dy, gam, 1e-5, dres=dres, want_param_grads=True)
    assert rel_l2(dx, xr.grad + dres.float()) < 6e-3
    assert rel_l2(dg, gr.grad) < 1e-4 and rel_l2(db, br.grad) < 1e-4
    dx2 = ops.layernorm_bwd(x, dy, gam, 1e-5)
    assert rel_l2(dx2, xr.grad) < 6e-3
    # activations: erf-GELU (1) and QuickGELU (2), forward and derivative
    u = bf(torch.randn(512, 2048, generator=g) * 2).to(dev)
    df = bf(torch.randn(512, 2048, generator=g)).to(dev)
    for act, fn in ((1, F.gelu), (2, lambda t: t * torch.sigmoid(1.702 * t))):
        ur = u.float().requires_grad_()
        fr = fn(ur)
        assert rel_l2(ops.act_bf16(u, act), fr) < 4e-3
        fr.backward(df.float())
        assert rel_l2(ops.act_bf16(u, act, df=df), ur.grad) < 4e-3


def test_gelu_fit_vs_exact_build_elementwise(dev):
    """The shipped erf-GELU of the bf16-output sites (five-term logistic fit, csrc/sc_common.h gelu_bf / gelu_bf2) against the exact-GELU
    checker build (libspeechclip_hip_gelu_exact.so: A&S 7.1.28, 3e-7) on IDENTICAL inputs in one process, element by element - the
    high-power half of the paired activation check (the model-level half is test_gpu_model.py::test_large_parallel_train_step).
    4.2 M inputs ~ N(0, 1.5): the two bf16 outputs may differ only where the fit's error (<= 3.1e-6 absolute, <= 2e-5 relative for
    x >= -1, tools/fit_gelu.py) crosses a rounding boundary.  Criteria fixed before the first run:
      every difference <= one bf16 ulp (2^-7 relative) or 8e-6 absolute; at most 3 % of the elements differ;
      |mean signed difference| <= 4e-6 (rounding is unbiased over the inputs, so the mean difference IS the fit's mean error) -
      a systematic activation error of the size the rejected three-term fit had (2.5e-5) fails this by a factor of six.
    Both the row kernel (sc_act_bf16) and the FC1-shaped GEMM epilogue (gemm256 tile family, gelu_bf2)."""
    from speechclip_plus_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(123)
    u = bf(torch.randn(2048, 2048, generator=g) * 1.5).to(dev)
    x = bf(torch.randn(2048, 768, generator=g)).to(dev)
    w = bf(torch.randn(3072, 768, generator=g) * 768 ** -0.5 * 1.5).to(dev)
    b = (torch.randn(3072, generator=g) * 0.05).to(dev)
    ship = (ops.act_bf16(u, 1).float(), ops.linear_bf16(x, w, b, act=1).float())
    with _lib.using_library(_lib.GELU_EXACT_LIB_PATH):
        exact = (ops.act_bf16(u, 1).float(), ops.linear_bf16(x, w, b, act=1).float())
    plain = ops.linear_bf16(x, w, b).float()                     # the same GEMM without the activation: identical in both builds
    with _lib.using_library(_lib.GELU_EXACT_LIB_PATH):
        assert torch.equal(plain, ops.linear_bf16(x, w, b).float())
    for name, ys, ye in (("sc_act_bf16", ship[0], exact[0]), ("gemm epilogue", ship[1], exact[1])):
        d = ys - ye
        frac = float((d != 0).float().mean())
        worst = float((d.abs() - torch.maximum(ye.abs() * 2.0 ** -7, torch.full_like(ye, 8e-6))).max())
        mean = float(d.double().mean())
        print("%s: %.3f %% of %d elements differ, mean signed difference %.2e, worst excess over one ulp %.1e" % (name, 100 * frac, d.numel(), mean, worst))
        assert torch.isfinite(ys).all()
        assert worst <= 0 and frac <= 0.03 and abs(mean) <= 4e-6, (name, worst, frac, mean)


def test_transpose_and_weight_gradient(dev):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(13)
    rows, N, K = 4096, 776, 520                                          # N, K not multiples of the tile sizes
    dy = bf(torch.randn(rows, N, generator=g)).to(dev)
    x = bf(torch.randn(rows, K, generator=g)).to(dev)
    wide = bf(torch.randn(rows, N + 40, generator=g)).to(dev)
    assert torch.equal(ops.transpose_bf16(dy), dy.t().contiguous())
    assert torch.equal(ops.transpose_bf16(wide[:, 8: 8 + N]), wide[:, 8: 8 + N].t().contiguous())      # strided source
    gW0, gb0 = torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    gW, gb = gW0.clone(), gb0.clone()
    ops.wgrad_bf16(dy, x, gW, gb)
    assert rel_l2(gW - gW0, dy.float().t() @ x.float()) < 1e-3
    assert rel_l2(gb - gb0, dy.float().sum(0)) < 1e-4
    gW2 = torch.full_like(gW0, 3.0)
    ops.wgrad_bf16(dy, x, gW2, beta=0.0)
    assert rel_l2(gW2, dy.float().t() @ x.float()) < 1e-3
    # a row count without a useful divisor (B x 499 frames): the split slices are padded, the pad must contribute nothing
    rows3 = 64 * 37
    dy3, x3 = bf(torch.randn(rows3, 2304, generator=g)).to(dev), bf(torch.randn(rows3, 768, generator=g)).to(dev)
    gW3, gb3 = torch.zeros(2304, 768, device=dev), torch.zeros(2304, device=dev)
    torch.empty(64 << 20, device=dev).fill_(float("nan"))               # poison the allocator's free blocks
    ops.wgrad_bf16(dy3, x3, gW3, gb3, beta=0.0)
    assert rel_l2(gW3, dy3.float().t() @ x3.float()) < 1e-3 and rel_l2(gb3, dy3.float().sum(0)) < 1e-4


def test_gemm_random_shape_fuzz(dev):
    """Seeded fuzz of sc_gemm_bf16 over every tile variant: M from 1 row to a few thousand (tails in every tile size, fewer tiles
    than CUs and more), N any multiple of 8, K any multiple of 64, random epilogue (bias / GELU / residual / fp32 output), strided
    A and C.  The persistent 256-row kernel walks several tiles per workgroup only when tiles > CUs: forced with small grids is not
    possible, so large-M cases are included."""
    import random
    ops = _ops()
    rng = random.Random(1234)
    g = torch.Generator(device="cpu").manual_seed(99)
    cases = []
    for _ in range(40):
        M = rng.choice([1, 7, 63, 64, 129, 255, 256, 257, 511, 700, 1000, 2048, 3001, 8192, 70000])
        N = 8 * rng.randint(1, 130)
        K = 64 * rng.randint(1, 20)
        cases.append((M, N, K, rng.random() < 0.5, rng.random() < 0.4, rng.random() < 0.4, rng.random() < 0.25, rng.choice([0, 1, 2, 3, 7, 8, 2, 7, 8])))
    # narrow outputs on the 128 x 64 tile (the grouped pos_conv is 48 wide)
    cases += [(300, 48, 192, True, True, True, False, 3), (129, 40, 64, True, False, False, True, 3), (1000, 8, 128, False, True, False, False, 3),
              (257, 24, 320, True, False, True, False, 3), (640, 56, 128, True, True, False, False, 3), (512, 64, 256, False, False, True, False, 3)]
    for M, N, K, use_bias, act, use_res, out_f32, tile in cases:
        if tile == 3 and N > 64:
            tile = 0
        pad_a, pad_c = 8 * rng.randint(0, 2), 8 * rng.randint(0, 2)
        A = bf(torch.randn(M, K + pad_a, generator=g)).to(dev)
        W = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev) if use_bias else None
        res = bf(torch.randn(M, N, generator=g)).to(dev) if use_res else None
        C = torch.full((M, N + pad_c), 3.0, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
        ops.gemm_raw(A, K + pad_a, W, K, C, N + pad_c, M, N, K, bias=bias, residual=res, ldr=N, act=int(act), out_f32=out_f32, tile=tile)
        ref = A[:, :K].float() @ W.float().T
        if use_bias:
            ref = ref + bias
        if act:
            ref = F.gelu(ref)
        if use_res:
            ref = ref + res.float()
        tag = (M, N, K, use_bias, act, use_res, out_f32, tile)
        assert rel_l2(C[:, :N], ref) < (2e-4 if out_f32 else 6e-3), tag
        if pad_c:
            assert float((C[:, N:].float() - 3.0).abs().max()) == 0.0, tag          # nothing written past column N


def test_attention_fwd_bwd_fuzz(dev):
    """Seeded fuzz of the attention forward / backward kernels over batch, heads, padded length, key lengths (incl. 1 and R) and
    the causal flag, against fp32 torch."""
    import random
    ops = _ops()
    rng = random.Random(77)
    g = torch.Generator(device="cpu").manual_seed(78)
    for _ in range(10):
        B, H, R = rng.choice([1, 2, 3]), rng.choice([1, 2, 4, 12]), rng.choice([128, 256, 384, 512])
        D = 64 * H
        causal = rng.random() < 0.4
        valid = [rng.choice([1, 2, 63, 64, 65, R // 2, R - 1, R]) for _ in range(B)]
        qkv = bf(torch.randn(B * R, 3 * D, generator=g)).to(dev)
        q, k, v = qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:]
        vl = torch.tensor(valid, dtype=torch.int32, device=dev)
        vt = ops.head_transpose(v, B, R, H)
        out = torch.zeros(B * R, D, device=dev, dtype=torch.bfloat16)
        lse2 = torch.empty(B, H, R, device=dev, dtype=torch.float32)
        scale = 64 ** -0.5
        ops.attn_fwd(qkv[:, : 2 * D], vt, vl, out, B, R, H, D, scale, lse2=lse2, causal=causal)
        qf, kf, vf = (t.float().view(B, R, H, 64).transpose(1, 2).detach().requires_grad_() for t in (q, k, v))
        s = (qf @ kf.transpose(-1, -2)) * scale
        key_ok = torch.arange(R, device=dev)[None, :] < vl[:, None]
        mask = key_ok[:, None, None, :].expand(B, H, R, R).clone()
        if causal:
            mask &= torch.tril(torch.ones(R, R, dtype=torch.bool, device=dev))[None, None]
        o_ref = (torch.softmax(s.masked_fill(~mask, float("-inf")), dim=-1) @ vf).transpose(1, 2).reshape(B * R, D)
        rows_ok = key_ok.reshape(B * R)                                  # queries inside the utterance
        tag = (B, H, R, valid, causal)
        assert rel_l2(out[rows_ok], o_ref[rows_ok]) < 1e-2, tag
        dout = bf(torch.randn(B * R, D, generator=g)).to(dev)
        dout[~rows_ok] = 0
        o_ref.backward(dout.float())
        dqkv = torch.full((B * R, 3 * D), 5.0, device=dev, dtype=torch.bfloat16)
        ops.attn_bwd(q, k, v, out, dout, lse2, vl, dqkv[:, :D], dqkv[:, D: 2 * D], dqkv[:, 2 * D:], B, R, H, scale, causal=causal)
        ref = [t.grad.transpose(1, 2).reshape(B * R, D) for t in (qf, kf, vf)]
        for i in range(3):
            gi = dqkv[rows_ok, i * D: (i + 1) * D]
            if float(ref[i][rows_ok].norm()) > 1e-6:
                assert rel_l2(gi, ref[i][rows_ok]) < 2.5e-2, (tag, i)
        assert float(dqkv[~rows_ok][:, D:].abs().max() if (~rows_ok).any() else 0.0) == 0.0, tag


# ---- train-mode dropout: the stateless hash mask, reconstructed on the host ------------------------------------------------
def _hash32_np(x):
    import numpy as np
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def _keep_mask(idx, seed, p):
    """idx: int64 numpy array of linear element indices -> bool keep mask (csrc/sc_common.h: sc_keep8)."""
    import numpy as np
    h = _hash32_np((idx >> 1).astype(np.uint32) ^ np.uint32(seed))
    bits = np.where(idx & 1, h >> np.uint32(16), h & np.uint32(0xffff))
    return bits >= np.uint32(int(p * 65536 + 0.5))


def _keep_mask8(idx, seed, p):
    """Attention-probability dropout (csrc/sc_common.h sc_drop8_*): one hash word per four consecutive elements, position i of the quad
    reads byte (0, 2, 1, 3)[i]; keep iff byte >= round(256 p) -> (bool keep mask, the rate actually applied = round(256 p) / 256)."""
    import numpy as np
    h = _hash32_np((idx >> 2).astype(np.uint32) ^ np.uint32(seed))
    pos = (idx & 3).astype(np.uint32)
    shift = np.uint32(8) * ((pos & np.uint32(1)) * np.uint32(2) + (pos >> np.uint32(1)))
    thr8 = int(p * 256 + 0.5)
    return ((h >> shift) & np.uint32(0xff)) >= np.uint32(thr8), thr8 / 256.0


def test_dropout_masks_gemm_rows_attention(dev):
    import numpy as np
    from speechclip_plus_amd import _lib
    ops = _ops()
    assert int(_hash32_np(np.array([12345], dtype=np.uint32))[0]) == _lib.lib().sc_hash32(12345)
    g = torch.Generator(device="cpu").manual_seed(55)
    p, seed = 0.25, 0x1234abcd
    # rows
    x = bf(torch.randn(200, 768, generator=g)).to(dev)
    y = ops.dropout_bf16(x, p, seed)
    keep = torch.from_numpy(_keep_mask(np.arange(200 * 768, dtype=np.int64), seed, p)).view(200, 768).to(dev)
    assert torch.equal(y, bf(torch.where(keep, x.float() / (1 - p), torch.zeros((), device=dev))))
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    # GEMM epilogue: dropout after bias / GELU, before the residual; both tile families
    for M, N, K, tile in ((300, 136, 128, 1), (1000, 512, 192, 2), (700, 776, 64, 7), (700, 776, 64, 8)):
        A = bf(torch.randn(M, K, generator=g)).to(dev)
        W = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        res = bf(torch.randn(M, N, generator=g)).to(dev)
        out = ops.linear_bf16(A, W, bias, residual=res, act=1, tile=tile, drop_p=p, drop_seed=seed)
        keep = torch.from_numpy(_keep_mask(np.arange(M * N, dtype=np.int64), seed, p)).view(M, N).to(dev)
        ref = torch.where(keep, F.gelu(A.float() @ W.float().T + bias) / (1 - p), torch.zeros((), device=dev)) + res.float()
        assert rel_l2(out, ref) < 6e-3, (M, N, K, tile)
        plain = ops.linear_bf16(A, W, bias, residual=res, act=1, tile=tile)
        assert torch.equal(ops.linear_bf16(A, W, bias, residual=res, act=1, tile=tile, drop_p=0.0, drop_seed=seed), plain)
    # attention probabilities: P' = mask . P / (1 - p), row sums over the un-masked P
    B, H, R = 2, 3, 256
    D = H * 64
    qkv = bf(torch.randn(B * R, 3 * D, generator=g)).to(dev)
    vl = torch.tensor([200, 256], dtype=torch.int32, device=dev)
    vt = ops.head_transpose(qkv[:, 2 * D:], B, R, H)
    out = torch.zeros(B * R, D, device=dev, dtype=torch.bfloat16)
    ops.attn_fwd(qkv[:, : 2 * D], vt, vl, out, B, R, H, D, 64 ** -0.5, drop_p=p, drop_seed=seed)
    qf, kf, vf = (t.float().view(B, R, H, 64).transpose(1, 2) for t in (qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:]))
    s = (qf @ kf.transpose(-1, -2)) * 64 ** -0.5
    key_ok = torch.arange(R, device=dev)[None, :] < vl[:, None]
    P = torch.softmax(s.masked_fill(~key_ok[:, None, None, :], float("-inf")), dim=-1)
    keep, p_att = _keep_mask8(np.arange(B * H * R * R, dtype=np.int64), seed, p)
    keep = torch.from_numpy(keep).view(B, H, R, R).to(dev)
    assert p_att == 0.25 and abs(float(keep.float().mean()) - (1 - p_att)) < 0.005
    ref = ((P * keep / (1 - p_att)) @ vf).transpose(1, 2).reshape(B * R, D)
    rows_ok = key_ok.reshape(B * R)
    assert rel_l2(out[rows_ok], ref[rows_ok]) < 1.2e-2
    # backward through the dropped probabilities: the DROP variants of the dq / dkv kernels regenerate the same mask
    lse2 = torch.empty(B, H, R, device=dev, dtype=torch.float32)
    ops.attn_fwd(qkv[:, : 2 * D], vt, vl, out, B, R, H, D, 64 ** -0.5, lse2=lse2, drop_p=p, drop_seed=seed)
    qa, ka, va = (t.detach().clone().requires_grad_() for t in (qf, kf, vf))
    Pa = torch.softmax(((qa @ ka.transpose(-1, -2)) * 64 ** -0.5).masked_fill(~key_ok[:, None, None, :], float("-inf")), dim=-1)
    oa = ((Pa * keep / (1 - p_att)) @ va).transpose(1, 2).reshape(B * R, D)
    dout = bf(torch.randn(B * R, D, generator=g)).to(dev)
    dout[~rows_ok] = 0
    oa.backward(dout.float())
    dqkv = torch.zeros(B * R, 3 * D, device=dev, dtype=torch.bfloat16)
    ops.attn_bwd(qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:], out, dout, lse2, vl, dqkv[:, :D], dqkv[:, D: 2 * D], dqkv[:, 2 * D:],
                 B, R, H, 64 ** -0.5, drop_p=p, drop_seed=seed)
    for i, t in enumerate((qa, ka, va)):
        gref = t.grad.transpose(1, 2).reshape(B * R, D)
        assert rel_l2(dqkv[rows_ok, i * D: (i + 1) * D], gref[rows_ok]) < 3e-2, i


@pytest.mark.parametrize("D,H,S,p_drop", [(768, 1, 37, 0.0), (256, 2, 150, 0.0), (256, 2, 70, 0.25)])
def test_mha_norm_block_vs_torch(dev, D, H, S, p_drop):
    """mha_block.MhaNormFn (attention block of the cascaded+/hybrid+ branches on own kernels, head_dim 768 / 128) against
    nn.MultiheadAttention + residual + LayerNorm in fp32: output, input gradient and every parameter gradient; with dropout the
    reference is the manual fp32 composition fed the same keep mask (the stateless hash of csrc/softmax.hip, rebuilt on the
    host from the call's seed)."""
    import numpy as np
    from speechclip_plus_amd import mha_block
    from speechclip_plus_amd.mha_block import mha_norm
    B = 3
    torch.manual_seed(21)
    mha = torch.nn.MultiheadAttention(D, H, dropout=p_drop, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(D).to(dev)
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1)
        mha.out_proj.bias.normal_(0, 0.1)
        norm.weight.normal_(1, 0.1)
        norm.bias.normal_(0, 0.1)
    g = torch.Generator(device="cpu").manual_seed(8)
    x0 = bf(torch.randn(B, S, D, generator=g)).float().to(dev)
    lens = torch.tensor([S, S - 11, 5], device=dev)
    kpm = torch.arange(S, device=dev)[None, :] >= lens[:, None]
    gout = torch.randn(B, S, D, generator=g).to(dev)
    gout[kpm] = 0                                                      # padded frames are never read downstream
    x = x0.clone().requires_grad_()
    torch.manual_seed(99)
    mha_block._calls = 0
    out = mha_norm(x, mha, norm, kpm, training=p_drop > 0)
    (out * gout).sum().backward()
    got = {"x": x.grad.clone(), **{n: p.grad.clone() for n, p in list(mha.named_parameters()) + [("ln." + n, p) for n, p in norm.named_parameters()]}}
    for p in list(mha.parameters()) + list(norm.parameters()):
        p.grad = None
    # fp32 reference (manual composition so that the dropout mask can be injected)
    xr = x0.clone().requires_grad_()
    dh, Sp = D // H, (S + 63) // 64 * 64
    qkv = F.linear(xr, mha.in_proj_weight, mha.in_proj_bias)
    q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))
    sc = (q @ k.transpose(-1, -2)) * dh ** -0.5
    P = torch.softmax(sc.masked_fill(kpm[:, None, None, :], float("-inf")), dim=-1)
    if p_drop > 0:
        mha_block._calls = 0
        seed = mha_block._next_seed()
        keep = torch.from_numpy(_keep_mask(np.arange(B * H * Sp * Sp, dtype=np.int64), seed, p_drop)).view(B, H, Sp, Sp)[:, :, :S, :S].to(dev)
        assert 0.7 < float(keep.float().mean()) < 0.8
        P = P * keep / (1 - p_drop)
    cx = (P @ v).transpose(1, 2).reshape(B, S, D)
    ref = norm(F.linear(cx, mha.out_proj.weight, mha.out_proj.bias) + xr)
    (ref * gout).sum().backward()
    valid = ~kpm
    assert rel_l2(out[valid], ref[valid]) < 1e-2
    assert rel_l2(got["x"][valid], xr.grad[valid]) < 3e-2
    for n, p in list(mha.named_parameters()) + [("ln." + n, p) for n, p in norm.named_parameters()]:
        assert rel_l2(got[n], p.grad) < 3e-2, (n, rel_l2(got[n], p.grad))
    if p_drop == 0:                                                    # and against the stock module itself
        with torch.no_grad():
            stock = norm(mha(x0, x0, x0, key_padding_mask=kpm)[0] + x0)
        assert rel_l2(out[valid], stock[valid]) < 1e-2


@pytest.mark.parametrize("normalize", [False, True])
def test_weighted_sum_logit_gradient_on_a_residual_stream(dev, normalize):
    """sc_wsum_bwd on hidden states that differ little from layer to layer (a residual stream: h_n = base + 0.03 delta_n): the
    gradient of the logits, w_n (d_n - sum_m w_m d_m), is a difference of nearly equal inner products.  Against fp64 on the SAME
    bf16 states the kernel path must keep 4 digits (it subtracts the last layer before accumulating); what remains against an fp32
    oracle in the model tests is then the bf16 representation of the states, not the kernel's arithmetic."""
    from speechclip_plus_amd.weighted_sum import WeightedSumLayer
    g = torch.Generator(device="cpu").manual_seed(5 + normalize)
    NL, B, T, D = 13, 3, 200, 1024
    base = torch.randn(B, T, D, generator=g) * 2.0 + 0.5
    hs = [(base + 0.03 * torch.randn(B, T, D, generator=g)).to(torch.bfloat16) for _ in range(NL)]
    gout = torch.randn(B, T, D, generator=g).to(torch.bfloat16).float()     # the layer's output is bf16: so is its gradient
    w0 = torch.randn(NL, generator=g) * 0.5
    layer = WeightedSumLayer(NL, normalize_features=normalize).to(dev)
    with torch.no_grad():
        layer.weights.copy_(w0)
    out = layer([h.to(dev) for h in hs])
    (out.float() * gout.to(dev)).sum().backward()
    w = w0.double().requires_grad_()
    stack = torch.stack([h.double() for h in hs])
    if normalize:
        stack = torch.nn.functional.layer_norm(stack, (D,))
    ref = (torch.softmax(w, 0).view(NL, 1, 1, 1) * stack).sum(0)
    (ref * gout.double()).sum().backward()
    assert rel_l2(out.detach().float().cpu(), ref.detach().float()) < 5e-3
    e = rel_l2(layer.weights.grad.double().cpu(), w.grad)
    assert e < 1e-4, e


@pytest.mark.parametrize("n,p_drop", [(72, 0.0), (512, 0.0), (1024, 0.2), (260, 0.5)])
def test_softmax_rows_fwd_bwd_vs_torch(dev, n, p_drop):
    """csrc/softmax.hip on its own: masked row softmax (+ hash dropout) and its backward against fp32 torch, row lengths that use
    1, 2 and 4 chunks per lane, fully masked rows, the host-rebuilt dropout mask."""
    import numpy as np
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(n)
    nb, rpb, seed, scale = 3, 40, 0x51f15e, 0.37
    rows = nb * rpb
    scores = (torch.randn(rows, n, generator=g) * 3).to(dev)
    mask = torch.rand(nb, n, generator=g) < 0.3
    mask[1] = True                                                     # one batch with every key padded: probabilities all 0
    mask_d = mask.to(dev).to(torch.uint8).contiguous()
    P, Pd = ops.softmax_fwd(scores, mask_d, rpb, scale, p_drop, seed)
    mrow = mask.to(dev).repeat_interleave(rpb, dim=0)
    sr = scores.clone().requires_grad_()
    ref = torch.softmax((sr * scale).masked_fill(mrow, float("-inf")), dim=-1)
    ref = torch.nan_to_num(ref, nan=0.0)
    live = ~mrow.all(dim=1)
    assert rel_l2(P[live], ref[live]) < 4e-3 and float(P[~live].float().abs().max()) == 0.0
    keep = torch.ones(rows, n, device=dev)
    if p_drop > 0:
        keep = torch.from_numpy(_keep_mask(np.arange(rows * n, dtype=np.int64), seed, p_drop)).view(rows, n).to(dev).float() / (1 - p_drop)
        assert rel_l2(Pd[live], (ref * keep)[live]) < 4e-3
    else:
        assert Pd is P
    dPd = torch.randn(rows, n, generator=g).to(dev)
    (ref * keep * dPd)[live].sum().backward()
    dS = ops.softmax_bwd(dPd, P, scale, p_drop, seed)
    # the kernel differentiates through the bf16-rounded P it is given: compare against the fp32 gradient at bf16 tolerance
    assert rel_l2(dS[live], sr.grad[live]) < 1.5e-2


@pytest.mark.parametrize("tile", [7, 8])
def test_gemm_layernorm_folding(dev, tile):
    """LayerNorm without a LayerNorm kernel (sc_gemm_args ln_* / res_* / stats_out, csrc/gemm256_bf16.hip "LN"): a residual GEMM emits
    the row statistics of what it stores, the next GEMM multiplies the RAW rows with W diag(gamma) and finishes the normalisation in its
    epilogue, and a residual operand that is itself raw is normalised on the fly - against the explicit fp32 form, on both tile
    widths (6 and 8 column chunks per row in the producer's epilogue), with an M tail."""
    import numpy as np
    ops = _ops()
    M, D, F_, K0 = 1000, 768, 1024, 256
    g = torch.Generator(device="cpu").manual_seed(100 + tile)
    A0 = bf(torch.randn(M, K0, generator=g)).to(dev)
    W0 = bf(torch.randn(D, K0, generator=g) * K0 ** -0.5).to(dev)
    b0 = torch.randn(D, generator=g).to(dev)
    R0 = bf(torch.randn(M, D, generator=g) + 0.7).to(dev)                 # rows with a non-zero mean
    # ---- producer with a plain residual
    C0 = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
    st0 = torch.zeros(M, 8, 2, device=dev, dtype=torch.float32)
    ns0 = ops.gemm_raw(A0, K0, W0, K0, C0, D, M, D, K0, bias=b0, residual=R0, ldr=D, tile=tile, stats_out=st0, ln_eps=1e-5)
    assert ns0 == (4 if tile == 7 else 3)
    ref0 = A0.float() @ W0.float().T + b0 + R0.float()
    assert rel_l2(C0, ref0) < 6e-3
    c0 = C0.float()
    s = st0[:, :ns0].sum(1)
    assert rel_l2(s[:, 0], c0.sum(1)) < 1e-5 and rel_l2(s[:, 1], (c0 * c0).sum(1)) < 1e-5
    # ---- consumer: y = gelu(LN(C0) W1^T + b1) through the folded weights
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    W1 = (torch.randn(F_, D, generator=g) * D ** -0.5).to(dev)
    b1 = torch.randn(F_, generator=g).to(dev)
    W1f = (W1 * gam[None, :]).to(torch.bfloat16).contiguous()
    colsum = W1f.float().sum(1).contiguous()
    cvec = (W1 @ bet + b1).contiguous()
    ln0 = F.layer_norm(c0, (D,), gam, bet, 1e-5)
    for act in (0, 1):
        Y = torch.zeros(M, F_, device=dev, dtype=torch.bfloat16)
        ops.gemm_raw(C0, D, W1f, D, Y, F_, M, F_, D, bias=cvec, act=act, tile=tile, ln_stats=st0, ln_ns=ns0, ln_colsum=colsum, ln_eps=1e-5)
        ref = ln0 @ W1.T + b1
        ref = F.gelu(ref) if act else ref
        assert rel_l2(Y, ref) < 8e-3, (act, rel_l2(Y, ref))
    # ---- producer whose residual is the raw C0: C1 = A1 W2^T + b2 + LN(C0), with statistics
    A1 = bf(torch.randn(M, F_, generator=g)).to(dev)
    W2 = bf(torch.randn(D, F_, generator=g) * F_ ** -0.5).to(dev)
    b2 = torch.randn(D, generator=g).to(dev)
    C1 = torch.zeros(M, D, device=dev, dtype=torch.bfloat16)
    st1 = torch.zeros(M, 8, 2, device=dev, dtype=torch.float32)
    for p_drop in (0.0, 0.1):
        ns1 = ops.gemm_raw(A1, F_, W2, F_, C1, D, M, D, F_, bias=b2, residual=C0, ldr=D, tile=tile, stats_out=st1, res_stats=st0,
                           res_ns=ns0, res_gamma=gam, res_beta=bet, ln_eps=1e-5, drop_p=p_drop, drop_seed=77)
        prod = A1.float() @ W2.float().T + b2
        if p_drop > 0:
            keep = torch.from_numpy(_keep_mask(np.arange(M * D, dtype=np.int64), 77, p_drop)).view(M, D).to(dev)
            prod = torch.where(keep, prod / (1 - p_drop), torch.zeros_like(prod))
        ref1 = prod + ln0
        assert rel_l2(C1, ref1) < 6e-3, rel_l2(C1, ref1)
        c1 = C1.float()
        s1 = st1[:, :ns1].sum(1)
        assert rel_l2(s1[:, 0], c1.sum(1)) < 1e-5 and rel_l2(s1[:, 1], (c1 * c1).sum(1)) < 1e-5
    # the 128-row tile family has no LayerNorm folding: asking for it there is an error, not a silent plain GEMM
    with pytest.raises(RuntimeError):
        ops.gemm_raw(C0, D, W1f, D, Y, F_, M, F_, D, bias=cvec, tile=1, ln_stats=st0, ln_ns=ns0, ln_colsum=colsum, ln_eps=1e-5)


def test_rowtail_ops(dev):
    """csrc/rowtail.hip (round 3: the head's B-row tail in ~40 launches): the skinny fp32 MFMA GEMM in all its forms - split slices
    consumed by the next op, sliced A operand with row-scaled bias, epilogue (bias, GELU with kept pre-activation, hash dropout,
    accumulate), weight-gradient form with the bias gradient as a by-product, batched per-head forms - LayerNorm forward / backward
    over slices (single and chained, dropout, residual), the elementwise op and the unit-row op, against fp64 torch."""
    import numpy as np
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(321)
    R = lambda *s: torch.randn(*s, generator=g)
    B, D, F_, E, H = 50, 384, 640, 72, 4            # awkward sizes on purpose (tails everywhere)
    dh = D // H
    x, W, b = R(B, D).to(dev), (R(F_, D) * D ** -0.5).to(dev), R(F_).to(dev)
    d64 = lambda t: t.double().cpu()
    rel_l2 = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))
    # split product + elementwise consumer (GELU + dropout)
    ys = ops.rt_gemm(x, W, B, F_, D, split=True)
    assert ys.ns > 1
    ref = d64(x) @ d64(W).T
    assert rel_l2(ys.total(), ref) < 1e-6
    u, f = ops.rt_elem(ys, 1, bias=b, drop_p=0.25, drop_seed=99)
    keep = torch.from_numpy(_keep_mask(np.arange(B * F_, dtype=np.int64), 99, 0.25)).view(B, F_)
    assert rel_l2(u, ref + d64(b)) < 1e-6
    assert rel_l2(f, torch.where(keep, F.gelu(ref + d64(b)) / 0.75, torch.zeros(()).double())) < 1e-6
    # un-split epilogue: alpha, bias, GELU with U, dropout, beta
    C0 = R(B, F_).to(dev)
    C = C0.clone()
    U = torch.empty_like(C)
    ops.rt_gemm(x, W, B, F_, D, out=C, alpha=0.5, beta=2.0, bias=b, act=1, U=U, drop_p=0.25, drop_seed=99)
    pre = 0.5 * ref + d64(b)
    assert rel_l2(U, pre) < 1e-6
    assert rel_l2(C, torch.where(keep, F.gelu(pre) / 0.75, torch.zeros(()).double()) + 2 * d64(C0)) < 1e-6
    # sliced A with row-scaled bias, K-major B (the dgrad form): z = (sum_s ys + b * rs) W
    rs = R(B, 10).to(dev)                                   # group = F_ / 10 = 64
    zs = ops.rt_gemm(ys, W, B, D, F_, a_bias=b, a_rowscale=rs, a_group=64, b_kmajor=True, ldb=D, split=True)
    a_full = ref + d64(b)[None, :] * d64(rs).repeat_interleave(64, dim=1)
    assert rel_l2(zs.total(), a_full @ d64(W)) < 1e-6
    assert rel_l2(ops.rt_elem(ys, 0, bias=b, rowscale=rs, group=64), a_full) < 1e-6
    # weight-gradient form with bias gradient, accumulating
    dy = R(B, F_).to(dev)
    gW0, gb0 = R(F_, D).to(dev), R(F_).to(dev)
    gW, gb = gW0.clone(), gb0.clone()
    ops.rt_gemm(dy, x, F_, D, B, a_kmajor=True, b_kmajor=True, lda=F_, ldb=D, out=gW, beta=1.0, gb=gb)
    assert rel_l2(gW, d64(gW0) + d64(dy).T @ d64(x)) < 1e-6 and rel_l2(gb, d64(gb0) + d64(dy).sum(0)) < 1e-6
    # per-head batched forms: ctx_h = m_h Wv_h^T (split), dWv_h += dctx_h^T m_h with bias gradient, dm_h = dctx_h Wv_h
    m, Wv = R(B, H, D).to(dev), (R(D, D) * D ** -0.5).to(dev)
    cs = ops.rt_gemm(m, Wv, B, dh, D, nbatch=H, lda=H * D, a_z=D, ldb=D, b_z=dh * D, split=True, ldc=D, c_z=dh)
    ref_c = torch.einsum("bhk,hjk->bhj", d64(m), d64(Wv).view(H, dh, D)).reshape(B, D)
    assert rel_l2(cs.total(), ref_c) < 1e-6
    dc = R(B, D).to(dev)
    gWv, gbv = torch.zeros(D, D, device=dev), torch.zeros(D, device=dev)
    ops.rt_gemm(dc, m, dh, D, B, a_kmajor=True, b_kmajor=True, lda=D, ldb=H * D, nbatch=H, a_z=dh, b_z=D, out=gWv, ldc=D, c_z=dh * D,
                beta=1.0, gb=gbv, gb_z=dh)
    assert rel_l2(gWv, torch.einsum("bhj,bhk->hjk", d64(dc).view(B, H, dh), d64(m)).reshape(D, D)) < 1e-6
    assert rel_l2(gbv, d64(dc).sum(0)) < 1e-6
    dm = torch.empty(B, H, D, device=dev)
    ops.rt_gemm(dc, Wv, B, D, dh, nbatch=H, lda=D, a_z=dh, b_kmajor=True, ldb=D, b_z=dh * D, out=dm, ldc=H * D, c_z=D)
    assert rel_l2(dm, torch.einsum("bhj,hjk->bhk", d64(dc).view(B, H, dh), d64(Wv).view(H, dh, D))) < 1e-6
    # LayerNorm over slices: single (broadcast residual, dropout) and chained; backward with add, masked output, parameter gradients
    g1, b1, g2, b2 = (1 + 0.1 * R(D)).to(dev), (0.1 * R(D)).to(dev), (1 + 0.1 * R(D)).to(dev), (0.1 * R(D)).to(dev)
    bias_d, res0, resB = R(D).to(dev), R(1, D).to(dev), R(B, D).to(dev)
    keepD = torch.from_numpy(_keep_mask(np.arange(B * D, dtype=np.int64), 7, 0.1)).view(B, D)
    z_pre = torch.where(keepD, (zs.total().double().cpu() + d64(bias_d)) / 0.9, torch.zeros(()).double())
    o1, h1, r1 = ops.rt_ln_fwd(zs, bias_d, res0, 0, g1, b1, 1e-5, drop_p=0.1, drop_seed=7)
    zr = z_pre + d64(res0)
    ln = lambda t, gg, bb: F.layer_norm(t, (D,), d64(gg), d64(bb), 1e-5)
    assert rel_l2(o1, ln(zr, g1, b1)) < 1e-5 and rel_l2(h1, F.layer_norm(zr, (D,), None, None, 1e-5)) < 1e-5
    o1b, h1b, r1b, o2, h2, r2 = ops.rt_ln_fwd(zs, bias_d, resB, D, g1, b1, 1e-5, g2, b2, 1e-6, drop_p=0.1, drop_seed=7)
    z2 = (z_pre + d64(resB)).requires_grad_()
    y1 = ln(z2, g1, b1)
    y2 = F.layer_norm(y1, (D,), d64(g2), d64(b2), 1e-6)
    assert rel_l2(o1b, y1) < 1e-5 and rel_l2(o2, y2) < 1e-5
    dys = ops.Slices(R(3, B, D).to(dev))
    add = R(B, D).to(dev)
    dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx, dxm = ops.rt_ln_bwd(dys, add, h1b, g1, r1b, dgam, dbet, want_masked=True, drop_p=0.1, drop_seed=7)
    gin = d64(dys.total()) + d64(add)
    g1p, b1p = d64(g1).requires_grad_(), d64(b1).requires_grad_()
    y1p = F.layer_norm(z2, (D,), g1p, b1p, 1e-5)
    y1p.backward(gin)
    assert rel_l2(dx, z2.grad) < 1e-5 and rel_l2(dgam, g1p.grad) < 1e-5 and rel_l2(dbet, b1p.grad) < 1e-5
    assert rel_l2(dxm, torch.where(keepD, z2.grad / 0.9, torch.zeros(()).double())) < 1e-5
    # GELU backward over slices, unit rows
    du = ops.rt_elem(ys, 2, u=u, drop_p=0.25, drop_seed=99)
    up = (ref + d64(b)).requires_grad_()
    F.gelu(up).backward(torch.where(keep, ref / 0.75, torch.zeros(()).double()))
    assert rel_l2(du, up.grad) < 1e-5
    xo, e, rn = ops.rt_l2norm_fwd(zs, bias_d)
    xr = (d64(zs.total()) + d64(bias_d)).requires_grad_()
    er = xr / xr.norm(dim=-1, keepdim=True)
    assert rel_l2(xo, xr) < 1e-6 and rel_l2(e, er) < 1e-6
    ge = R(B, D).to(dev)
    er.backward(d64(ge))
    assert rel_l2(ops.rt_l2norm_bwd(ge, e, rn), xr.grad) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,rows,S", [(256, 256, 64, 1), (768, 768, 1024, 4), (512, 1536, 640, 2), (2304, 768, 512, 1)])
def test_gemm_tn_form_reads_row_major_operands(dev, M, N, rows, S):
    """sc_gemm_bf16 with tn = 1: C[m, n] = sum_r A[r, m] W[r, n] (the weight-gradient product dY^T X) straight from the row-major
    operands - against an fp64 product of the same bf16 values, un-split and split along r into fp32 partials; the second case's
    W operand is an overlapping-row (im2col) view as the conv layers' weight gradients use."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + rows)
    A = torch.randn(rows, M, generator=g).to(torch.bfloat16).to(dev)
    ldw = N if N != 1536 else 1024                      # rows overlap: row r of W starts at element r * 1024, is 1536 long
    Wflat = torch.randn(rows * ldw + N, generator=g).to(torch.bfloat16).to(dev)
    Wv = torch.as_strided(Wflat, (rows, N), (ldw, 1))
    ref = A.double().t() @ Wv.double()
    Kc = rows // S
    part = torch.empty(S, M, N, device=dev, dtype=torch.float32)
    ops.gemm_raw(A, M, Wflat, ldw, part, N, M, N, Kc, out_f32=True, nb1=S, sA=(Kc * M, 0), sW=(Kc * ldw, 0), sC=(M * N, 0), tn=True)
    got = part.double().sum(0)
    assert float((got - ref).norm() / ref.norm()) < 1e-5
    # the NT form on transposed copies gives the same numbers (same k order inside a tile): bitwise per slice
    At, Wt = A.t().contiguous(), Wv.t().contiguous()
    part2 = torch.empty_like(part)
    ops.gemm_raw(At, rows, Wt, rows, part2, N, M, N, Kc, out_f32=True, nb1=S, sA=(Kc, 0), sW=(Kc, 0), sC=(M * N, 0), tile=8)
    assert torch.equal(part, part2)
    if rows >= 512:
        # ragged slices (k_total): 3 slices of ceil(rows / 64 / 3) K-tiles, the last one shorter
        Kr = -(-(rows // 64) // 3) * 64
        Sr = -(-rows // Kr)
        part3 = torch.empty(Sr, M, N, device=dev, dtype=torch.float32)
        ops.gemm_raw(A, M, Wflat, ldw, part3, N, M, N, Kr, out_f32=True, nb1=Sr, sA=(Kr * M, 0), sW=(Kr * ldw, 0), sC=(M * N, 0), tn=True,
                     k_total=rows)
        assert float((part3.double().sum(0) - ref).norm() / ref.norm()) < 1e-5
    # ops.wgrad_bf16 on top of it, with the bias gradient
    gW = torch.zeros(M, N, device=dev)
    gb = torch.zeros(M, device=dev)
    ops.wgrad_bf16(A, Wv, gW, gb, beta=0.0)
    assert float((gW.double() - ref).norm() / ref.norm()) < 1e-5
    assert float((gb.double() - A.double().sum(0)).norm() / A.double().sum(0).norm()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("Dg,G,B,R", [(48, 2, 2, 128), (64, 3, 3, 256)])
def test_posconv_weight_gradient_kernel(dev, Dg, G, B, R):
    """sc_posconv_wgrad_bf16: gw[g][co][tap Dg + ci] = sum_m du[g][m][co] xg[g][m + tap][ci] over the halo-padded slab rows, against an
    fp64 correlation of the same bf16 values (and thereby against F.conv1d's weight gradient, which is that sum)."""
    ops = _ops()
    Kp, halo = 128, 64
    Rp = R + 2 * halo
    g = torch.Generator(device="cpu").manual_seed(Dg + R)
    xg = torch.zeros(G, B, Rp, Dg)
    xg[:, :, halo: halo + R] = torch.randn(G, B, R, Dg, generator=g)
    du_buf = torch.zeros((G * B * Rp + halo + 1) * Dg)
    dug = du_buf[: G * B * Rp * Dg].view(G, B, Rp, Dg)
    dug[:, :, halo: halo + R] = torch.randn(G, B, R, Dg, generator=g)
    xg_d, du_d = xg.to(torch.bfloat16).to(dev), du_buf.to(torch.bfloat16).to(dev)
    gw = ops.posconv_wgrad(du_d[halo * Dg:], xg_d, G, B * Rp, Dg, Kp)
    xr = xg_d.double().cpu()
    dr = du_d[: G * B * Rp * Dg].view(G, B, Rp, Dg).double().cpu()[:, :, halo: halo + R]      # du[g][b][t][co]
    ref = torch.empty(G, Dg, Kp, Dg, dtype=torch.float64)
    for tap in range(Kp):
        ref[:, :, tap] = torch.einsum("gbtc,gbti->gci", dr, xr[:, :, tap: tap + R])
    got = gw.double().cpu().view(G, Dg, Kp, Dg)
    assert float((got - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.gpu
def test_gemm_small_tile_ring_variants_agree_bitwise(dev):
    """128 x 64 tiles, 2-stage and 4-stage LDS ring (tile 3 / 13; the dispatcher picks the deep ring for small grids with long K) and the
    64 x 64 tiles (14 / 15; chosen for few rows x narrow output x long K): same K order, same epilogue - identical bits, with bias,
    GELU and residual; M with a tail in every tile size."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(4)
    for (M, N, K, act, res) in ((2048, 512, 2048, 0, True), (1000, 192, 1088, 1, False), (128, 64, 64, 0, False), (1733, 776, 1536, 1, True)):
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        R = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev) if res else None
        outs = []
        for t in (3, 13, 0, 14, 15):
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm_raw(A, K, W, K, C, N, M, N, K, bias=bias, residual=R, ldr=N, act=act, tile=t)
            outs.append(C)
        ref = A.float() @ W.float().t() + bias
        if act:
            ref = torch.nn.functional.gelu(ref)
        if res:
            ref = ref + R.float()
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        assert rel_l2(outs[0], ref) < 6e-3


@pytest.mark.gpu
@pytest.mark.parametrize("act", [1, 2])
def test_gemm_fused_activation_with_aux_operand(dev, act):
    """sc_gemm_args.aux_mode (128-row tiles): 1 = dual store (pre-activation + activation), 2 = product times act'(aux) - both must
    reproduce the two-launch sequences GEMM -> sc_act_bf16 bit for bit (erf-GELU and QuickGELU)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(act)
    M, N, K = 2048, 2048, 512
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    u_ref = ops.linear_bf16(x, w, b, tile=3)
    f_ref = ops.act_bf16(u_ref, act)
    u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = ops.linear_bf16(x, w, b, act=act, aux=u, aux_mode=1)
    assert torch.equal(u, u_ref) and torch.equal(f, f_ref)
    dy = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)           # gradient of an [M, K] output through W2 [K, N]
    w2T = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    df_ref = ops.linear_bf16(dy, w2T, tile=3)
    du_ref = ops.act_bf16(u_ref, act, df=df_ref)
    du = ops.linear_bf16(dy, w2T, act=act, aux=u_ref, aux_mode=2)
    assert torch.equal(du, du_ref)


def test_product_library_refuses_the_diagnostic_kernels():
    """VERDICT r03 "weak" 9: tile ids 32 (timing only, results wrong) / 34 (stamped) and the LayerNorm-folded GEMMs are built into
    libspeechclip_hip_diag.so only - the product library returns an error instead of launching them; ops routes the callers that ask
    for them (tools/, bench.py's clock probe, the SC_FUSED_LN experiment) to the diagnostics library."""
    import ctypes
    from speechclip_plus_amd import _lib, ops
    from speechclip_plus_amd._lib import GemmArgs
    assert _lib.lib().sc_is_diag_build() == 0 and _lib.diag_lib().sc_is_diag_build() == 1
    M, N, K = 512, 256, 128
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tile in (32, 34):
        a = GemmArgs()
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc = A.data_ptr(), K, W.data_ptr(), K, C.data_ptr(), N
        a.M, a.N, a.K, a.n_split, a.nb1, a.nb2, a.tile = M, N, K, -1, 1, 1, tile
        rc = _lib.lib().sc_gemm_bf16(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc != 0 and b"libspeechclip_hip_diag" in _lib.lib().sc_last_error()
    # the stamped kernel computes the right product (only tile 32 skips the epilogue); through ops it runs on the diagnostics library
    dbg = torch.zeros(4 * 8 * 8 * 8, device="cuda", dtype=torch.int64)
    ops.gemm_raw(A, K, W, K, C, N, M, N, K, tile=34, Ct=dbg.view(torch.bfloat16))
    assert rel_l2(C, A.float() @ W.float().T) < 1e-2


@pytest.mark.parametrize("tile", [2, 7, 8])
def test_gemm256_fused_gelu_with_aux_operand(dev, tile):
    """Round 4: sc_gemm_args.aux_mode on the 256-row tile family (erf-GELU; the fc1 / conv GEMMs of a differentiated HuBERT): the dual
    store and the product-times-gelu'(aux) epilogues reproduce GEMM -> sc_act_bf16 bit for bit, on both tile widths, also when M is not
    a multiple of the tile and for a conv-shaped (overlapping-row) A operand."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(tile)
    M, N, K = 2048 + 40, 768, 512
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    u_ref = ops.linear_bf16(x, w, b, tile=tile)
    f_ref = ops.act_bf16(u_ref, 1)
    u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = ops.linear_bf16(x, w, b, act=1, aux=u, aux_mode=1, tile=tile)
    assert torch.equal(u, u_ref) and torch.equal(f, f_ref)
    # (not equal to the plain act = 1 epilogue, which activates the UN-rounded accumulator: the dual store activates the stored bf16
    # value, as GEMM + sc_act_bf16 does - the pre-activation the backward differentiates is exactly the one that was activated)
    assert rel_l2(f, ops.linear_bf16(x, w, b, act=1, tile=tile)) < 4e-3
    dy = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w2T = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    df_ref = ops.linear_bf16(dy, w2T, tile=tile)
    du_ref = ops.act_bf16(u_ref, 1, df=df_ref)
    du = ops.linear_bf16(dy, w2T, act=1, aux=u_ref, aux_mode=2, tile=tile)
    assert torch.equal(du, du_ref)
    # conv-shaped A (k = 3, stride 2: lda = 2 C < K = 3 C) with the shared-tap K order
    C, rows = 256, 1024
    xin = torch.randn(2 * rows + 8, C, generator=g).to(torch.bfloat16).to(dev)
    wc = (torch.randn(C, 3 * C, generator=g) * (3 * C) ** -0.5).to(torch.bfloat16).to(dev)
    uc_ref = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
    ops.gemm_raw(xin, 2 * C, wc, 3 * C, uc_ref, C, rows, C, 3 * C, tap_c=C, tile=tile)
    uc, fc = torch.empty_like(uc_ref), torch.empty_like(uc_ref)
    ops.gemm_raw(xin, 2 * C, wc, 3 * C, fc, C, rows, C, 3 * C, act=1, tap_c=C, aux=uc, aux_mode=1, tile=tile)
    assert torch.equal(uc, uc_ref) and torch.equal(fc, ops.act_bf16(uc_ref, 1))


def test_conv_overlap_add_with_activation_backward(dev):
    """sc_conv_overlap_add_act_bf16 = sc_conv_overlap_add_bf16 followed by sc_act_bf16(u, dx), in one pass, bit for bit"""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(3)
    M, C = 4096, 512
    dcols = torch.randn(M, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    u = torch.randn(2 * M, C, generator=g).to(torch.bfloat16).to(dev)
    ref = ops.act_bf16(u, 1, df=ops.conv_overlap_add(dcols, C))
    assert torch.equal(ops.conv_overlap_add(dcols, C, u=u), ref)


@pytest.mark.gpu
def test_layernorm_bwd_with_dropped_copy_and_column_sums(dev):
    """sc_layernorm_bwd_drop_bf16: the same dx as the plain kernel, its dropped copy bit-identical to sc_dropout_bf16 of that dx (the
    stateless mask of element row * D + col), and the copy's column sums accumulated into a bias-gradient target (fp32, vs fp64)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(31)
    for rows, D, p in ((1000, 768, 0.1), (515, 1024, 0.25), (300, 512, 0.0)):
        x = bf(torch.randn(rows, D, generator=g)).to(dev)
        dy = bf(torch.randn(rows, D, generator=g)).to(dev)
        res = bf(torch.randn(rows, D, generator=g)).to(dev)
        gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
        ref_dx, ref_g, ref_b = ops.layernorm_bwd(x, dy, gamma, 1e-5, dres=res, want_param_grads=True)
        acc = (torch.full((D,), 2.0, device=dev), torch.full((D,), -1.0, device=dev))
        bias = torch.full((D,), 0.5, device=dev)
        if p > 0:
            dx, dxd = ops.layernorm_bwd(x, dy, gamma, 1e-5, dres=res, acc=acc, drop=(p, 777), sum_acc=bias)
            assert torch.equal(dxd, ops.dropout_bf16(dx, p, 777))
        else:
            dx = dxd = ops.layernorm_bwd(x, dy, gamma, 1e-5, dres=res, acc=acc, sum_acc=bias)
        assert torch.equal(dx, ref_dx)
        assert rel_l2(acc[0] - 2.0, ref_g) < 1e-5 and rel_l2(acc[1] + 1.0, ref_b) < 1e-5
        assert rel_l2(bias - 0.5, dxd.double().sum(0).float()) < 1e-5


@pytest.mark.gpu
def test_weighted_sum_share_kernel(dev):
    """sc_wsum_share_bf16 = the element-wise formulation it replaces: slice the frames out of dX, scale by one softmax weight, round to
    bf16, add to what came down from the layer above (bf16 add), padding rows untouched - bit-identical."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(32)
    B, R, T, D = 3, 128, 99, 768
    dX = torch.randn(B, R, D, generator=g).to(dev)
    w = torch.tensor([0.3, 0.0721, 0.6], device=dev)
    prev = bf(torch.randn(B * R, D, generator=g)).to(dev)
    dfeat = torch.zeros(B, R, D, device=dev)
    dfeat[:, :T] = dX[:, 1: T + 1]
    share = (dfeat.view(B * R, D) * w[1]).to(torch.bfloat16)
    assert torch.equal(ops.wsum_share(dX, w[1:2], None, B, R, T), share)
    assert torch.equal(ops.wsum_share(dX, w[1:2], prev, B, R, T), prev + share)


@pytest.mark.gpu
def test_attention_backward_is_bitwise_repeatable(dev):
    """The dK / dV kernel stages its tiles by LDS-DMA into a double buffer (one barrier per step): a missed wait or a buffer re-used
    too early would show as run-to-run differences.  Same inputs, five runs, several shapes incl. dropout and the causal text-tower
    form: identical bits every time (every gradient element is written once, in a fixed summation order)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(41)
    for B, H, R, T, causal, p in ((4, 12, 512, 499, 0, 0.1), (3, 8, 128, 128, 32, 0.0), (2, 12, 384, 300, 0, 0.0), (5, 4, 256, 256, 1, 0.0)):
        D = 64 * H
        qkv = bf(torch.randn(B * R, 3 * D, generator=g)).to(dev)
        dout = bf(torch.randn(B * R, D, generator=g)).to(dev)
        if T < R:
            dout.view(B, R, D)[:, T:] = 0
        valid = torch.full((B,), T if not causal else R, dtype=torch.int32, device=dev)
        vt = ops.head_transpose(qkv[:, 2 * D:], B, R, H)
        out = torch.empty(B * R, D, device=dev, dtype=torch.bfloat16)
        lse2 = torch.empty(B, H, R, device=dev, dtype=torch.float32)
        ops.attn_fwd(qkv[:, : 2 * D], vt, valid, out, B, R, H, D, 0.125, lse2=lse2, causal=causal, drop_p=p, drop_seed=5)
        ref = None
        for _ in range(5):
            dqkv = torch.full((B * R, 3 * D), 7.0, device=dev, dtype=torch.bfloat16)
            ops.attn_bwd(qkv[:, :D], qkv[:, D: 2 * D], qkv[:, 2 * D:], out, dout, lse2, valid, dqkv[:, :D], dqkv[:, D: 2 * D], dqkv[:, 2 * D:],
                         B, R, H, 0.125, causal=causal, q_rows=T if not causal else R, drop_p=p, drop_seed=5)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(dqkv.float()).all())
            if ref is None:
                ref = dqkv
            else:
                assert torch.equal(dqkv, ref), (B, H, R, T, causal, p)


def test_bf16_site_gelu_against_the_exact_erf_gelu():
    """csrc/sc_common.h gelu_bf / gelu_bf2 (round 5): the erf-GELU of every bf16-OUTPUT site is x * sigmoid(x p(t)), t = x^2 (round 5 clamped t at 36; round 6 dropped the
    clamp - p is negative and monotone beyond the fitted range, the bounds below are unchanged), p a five-term polynomial (fitted: tools/fit_gelu.py), because the FC1 / conv epilogues are VALU-bound on the activation.  Against the
    exact x Phi(x) (fairseq nn.GELU, torch erf form) on EVERY bf16 input in [-16, 16]: the fp32 error behind the bf16 store is bounded by
    3.5e-6 absolute - so the stored value is the exact result's bf16 rounding or, on a rounding boundary, its neighbour; for x >= -1 the
    fit's RELATIVE error is <= 2.1e-5, i.e. the stored value is within one bf16 rounding (<= 2^-8 relative) + 3e-5 of the exact one; the
    limits hold (gelu = x beyond +9, |gelu| < 1e-10 beyond -9)."""
    from speechclip_plus_amd import ops
    dev = torch.device("cuda:0")
    bits = torch.arange(0, 1 << 16, dtype=torch.int32)
    x = bits.to(torch.int16).view(torch.bfloat16)
    x = x[torch.isfinite(x.float()) & (x.float().abs() <= 16.0)].contiguous()
    pad = (-x.numel()) % 8
    xd = torch.cat([x, torch.zeros(pad, dtype=torch.bfloat16)]).to(dev)
    y = ops.act_bf16(xd, 1)[: x.numel()].float().cpu()
    xf = x.double()
    exact = xf * 0.5 * (1.0 + torch.erf(xf / 2.0 ** 0.5))
    ulp = torch.maximum(exact.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -7      # bf16 spacing at the exact value
    err = (y.double() - exact).abs()
    assert bool((err <= 0.5 * ulp + 3.5e-6).all()), float((err - 0.5 * ulp).max())
    sig = exact.abs() >= 4e-3
    body = (x.float() >= -1.0) & (exact.abs() >= 1e-30)            # (subnormal inputs are flushed: excluded from the RELATIVE bound)
    assert float((err[body] / exact.abs()[body]).max()) <= 2.0 ** -8 + 3e-5        # one bf16 rounding (half an ulp: 2^-9 .. 2^-8 relative) + the fit
    # the stored value is the exact result's own bf16 rounding for nearly every input, a neighbour otherwise
    same = (y == exact.float().to(torch.bfloat16).float())
    assert float(same[sig].float().mean()) > 0.995, float(same[sig].float().mean())
    big = x.float() >= 9.0
    assert torch.equal(y[big], x.float()[big])                                      # gelu(x) = x in bf16 beyond the clamp
    neg = x.float() <= -9.0
    assert float(y[neg].abs().max()) < 1e-10                                        # ... and 0 on the other side (exact: -1e-18 at x = -9)


@pytest.mark.parametrize("M,N,K,act_aux", [(2048, 1536, 512, False), (2048, 2048, 512, True), (2048, 2304, 768, False), (200, 3072, 768, True), (96, 192, 1024, False)])
def test_gemm_layernorm_prologue_against_layernorm_then_gemm(M, N, K, act_aux):
    """Round 6: LayerNorm in the prologue of the 128- / 64-row GEMM tiles (the text tower's LN -> QKV / LN -> fc pairs; csrc/gemm_bf16.hip
    ln_self, ops.fold_layernorm): y = rstd (x W'^T - mean colsum) + c against the two-launch sequence LayerNorm kernel -> GEMM and against
    fp64.  Criterion fixed before the first run: the fused result's error against fp64 is at most 1.5 x the two-launch sequence's
    (both are bf16-storage noise: the sequence rounds the normalised rows to bf16, the fused form rounds W diag(gamma)) + 1e-4 of the
    output's rms, and the two agree to 1.5e-2 rel-L2; with the QuickGELU dual store both outputs (u and act(u)) are checked."""
    from speechclip_plus_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 0.7).to(torch.bfloat16)      # rows with a mean
    w0 = torch.randn(N, K, generator=g) * K ** -0.5
    b0 = torch.randn(N, generator=g) * 0.1
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    xd, w0d, b0d, gd, bd = x.cuda(), w0.cuda(), b0.cuda(), gamma.cuda(), beta.cuda()
    w_ln, colsum, c = ops.fold_layernorm(w0d, b0d, gd, bd)
    h = ops.layernorm_bf16(xd, gd, bd, eps=1e-5)
    w_b = w0d.to(torch.bfloat16).contiguous()
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w0.double().t() + b0.double()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    if act_aux:
        u1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        u2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f1 = ops.linear_bf16(xd, w_ln, c, act=2, aux=u1, aux_mode=1, ln_colsum=colsum, ln_eps=1e-5)
        f2 = ops.linear_bf16(h, w_b, b0d, act=2, aux=u2, aux_mode=1)
        ref_f = ref * torch.sigmoid(1.702 * ref)
        pairs = ((u1, u2, ref), (f1, f2, ref_f))
    else:
        y1 = ops.linear_bf16(xd, w_ln, c, ln_colsum=colsum, ln_eps=1e-5)
        y2 = ops.linear_bf16(h, w_b, b0d)
        pairs = ((y1, y2, ref),)
    for fused, seq, r in pairs:
        e1, e2 = rel(fused, r), rel(seq, r)
        print(f"M {M} N {N} K {K}: fused vs fp64 {e1:.3e}, LayerNorm + GEMM vs fp64 {e2:.3e}, fused vs sequence {rel(fused, seq.double().cpu()):.3e}")
        assert torch.isfinite(fused.float()).all()
        assert e1 <= 1.5 * e2 + 1e-4, (e1, e2)
        assert rel(fused, seq.double().cpu()) <= 1.5e-2


@pytest.mark.parametrize("with_bias", [True, False])
def test_frontend_conv0_layer_norm_mode_closed_form_statistics(dev, with_bias):
    """Round 6: the layer_norm-mode conv layer 0 (HuBERT-large: conv -> +bias -> LayerNorm over the 512 channels -> erf-GELU,
    fairseq ConvFeatureExtractionModel in "layer_norm" mode, called at avssl/module/speech_encoder_plus.py:75) takes a row's channel mean
    and variance from closed forms of its 10-sample window (fp64 quadratic form; csrc/frontend.hip conv0_ln_gelu_stats_kernel) instead of
    two wavefront reductions.  Against the fp64 statement on rows of three kinds - noise, a quiet signal on a DC offset (the closed
    form's cancellation case: mean >> spread), silence - and against the two-pass kernel it replaces (sc_set_option(2, 1)).
    Criteria fixed before the first run: rel-L2 against fp64 < 5e-3 (the bound test_frontend_conv0 uses for the GroupNorm mode: one bf16
    store), no worse than 1.05 x the two-pass kernel's on every kind, every value finite, and the two kernels apart by at most one bf16
    ulp of the larger magnitude on every element; uniform rows with a row count that is not a multiple of the 128-row workgroup.
    (The first run passed the three accuracy criteria on every kind - 1.66e-3 / 1.75e-3 / 1.59e-3 for both kernels - and failed the
    last one as first written, without an absolute floor, on 2 of 1 536 000 elements: outputs of 4.6e-8 vs 4.7e-8 and 3.82e-7 vs 3.86e-7,
    GELU arguments that cancel to ~0, where gelu(u) ~ u / 2 carries the fp32 rounding of mean / rstd (1e-7 of O(1) terms) at full
    size whatever the output's magnitude - a one-ulp RELATIVE bound means nothing there.  The bound is therefore one bf16 ulp or 1e-6
    absolute, ten fp32 roundings of an O(1) term.)"""
    from speechclip_plus_amd._lib import lib
    ops = _ops()
    B, C, R0 = 3, 512, 1000
    L = 5 * (R0 - 1) + 10
    g = torch.Generator(device="cpu").manual_seed(5)
    wav = torch.randn(B, L, generator=g)
    wav[1] = wav[1] * 1e-3 + 0.7
    wav[2] = 0.0
    w0 = torch.randn(C, 10, generator=g) * 0.3
    b0 = torch.randn(C, generator=g) * 0.1 if with_bias else None
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wav_d = torch.zeros(B, L + 64, device=dev)
    wav_d[:, :L] = wav.to(dev)
    outs = {}
    for opt in (0, 1):
        lib().sc_set_option(2, opt)
        try:
            out = torch.zeros(B * R0, C, device=dev, dtype=torch.bfloat16)
            ops.conv0_layernorm_gelu(wav_d, w0.to(dev), None if b0 is None else b0.to(dev), gam.to(dev), bet.to(dev), R0, out)
            outs[opt] = out.view(B, R0, C).float().cpu()
        finally:
            lib().sc_set_option(2, 0)
    y = F.conv1d(wav.double().unsqueeze(1), w0.double().unsqueeze(1), b0.double() if with_bias else None, stride=5).transpose(1, 2)     # (B, R0, C)
    ref = F.gelu(F.layer_norm(y, (C,), gam.double(), bet.double(), 1e-5))
    assert bool(torch.isfinite(outs[0]).all())
    for b, kind in enumerate(("noise", "quiet + DC offset", "silence")):
        e_new, e_old = rel_l2(outs[0][b].double(), ref[b]), rel_l2(outs[1][b].double(), ref[b])
        print(f"{kind}: closed form {e_new:.3e}, two-pass {e_old:.3e}")
        assert e_new < 5e-3 and e_new <= 1.05 * e_old + 1e-6, (kind, e_new, e_old)
    ulp = torch.maximum(outs[0].abs(), outs[1].abs()).clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -7
    assert bool(((outs[0] - outs[1]).abs() <= torch.clamp(ulp, min=1e-6)).all())
