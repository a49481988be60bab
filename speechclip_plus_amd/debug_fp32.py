"""fp32 DEBUG mode of the HuBERT encoder ("fp32 kernel mode (for debugging) <= 1e-4 rel").

Same algorithm, same masks, same layouts of the reference's customHubertForward (avssl/module/speech_encoder_plus.py:29-107,
506-611) as the production path, but every tensor stays fp32 and every product runs in exact fp32 on the matrix pipe
(sc_sgemm_mfma_f32; the per-head attention products on the batched sc_sgemm_f32_ex), the norms on sc_rowln_f32_fwd, GELU on
sc_gelu_f32, the softmax and conv layer 0 on the fp32-output variants of the production kernels (sc_softmax_fwd_f32,
sc_conv0_gn_gelu_f32 / sc_conv0_ln_gelu_f32).  What it is for: separating "a kernel computes the wrong thing" from "bf16 storage
rounds" - in this mode the 13 / 25 hidden states must agree with the fp32 oracle to ~1e-5 (tests/test_gpu_model.py::
test_fp32_debug_mode_matches_the_oracle), so whatever the production path differs by beyond that is storage precision
(cf. the bf16-storage-emulated oracle, tests/test_gpu_recall.py).

Slow by design (one launch per utterance for the strided-row convs and the attention core, no fusion): a few hundred ms for a
handful of utterances.  torch is used for data movement only (views, slab / mask construction, stacking); the device does no
torch arithmetic on activations.  Not part of the product path; nothing imports it but tests and tools.
"""
from typing import Dict, List, Sequence, Tuple

import torch

from . import ops
from ._lib import lib
from .ops import _p, _stream, check


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _linear(x: torch.Tensor, w: torch.Tensor, b) -> torch.Tensor:
    return ops.sgemm_mfma(x, w, bias=b)


def _ln(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, res=None) -> torch.Tensor:
    return ops.rowln_fwd(x, res, x.shape[1] if res is not None else 0, g, b, 1e-5)[0]


def _add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b on the row kernels (sc_rt_elem adds the slices of a split product)."""
    return ops.rt_elem(ops.Slices(torch.stack([a, b]).contiguous()), 0)


def hubert_hidden_states_fp32(sd: Dict[str, torch.Tensor], arch, wavs: Sequence[torch.Tensor], device="cuda",
                              debug: dict = None) -> Tuple[List[torch.Tensor], List[int]]:
    """-> ([encoder input, layer 1 .. layer NL] each [B, T, D] fp32 on the device, feat_len per utterance).
    ``debug``: receives the intermediate stages ("conv": the 7 conv outputs [B, T_l, C], "proj": the projection [B, T, D])."""
    dev = torch.device(device)
    L_ = lib()
    B = len(wavs)
    lens = [int(w.numel()) for w in wavs]
    L = max(lens)
    ks, ss = list(arch.conv_kernels), list(arch.conv_strides)
    Ts = []
    t = L
    for k, s in zip(ks, ss):
        t = (t - k) // s + 1
        Ts.append(t)
    T0, T = Ts[0], Ts[-1]
    C, D, F_, H = arch.conv_dim, arch.embed_dim, arch.ffn_dim, arch.heads
    dh = D // H
    ln_mode = arch.extractor_mode == "layer_norm"

    # ---- waveform: zero-padded rows, per-utterance layer norm for the large checkpoint (speech_encoder_plus.py:506-518)
    ldw = (5 * (T0 - 1) + 10 + 15) // 16 * 16 + 16
    ldw = max(ldw, (L + 15) // 16 * 16)
    wav = torch.zeros(B, L, device=dev)
    for b, w in enumerate(wavs):
        wav[b, : lens[b]] = w.to(dev).float()
    wav_pad = torch.zeros(B, ldw, device=dev)
    ops.wav_prep(wav, torch.tensor(lens, dtype=torch.int64, device=dev), wav_pad, bool(arch.normalize_wav))

    # ---- conv layer 0 (+ GroupNorm over time per channel / LayerNorm over channels) + GELU, fp32 stores
    w0 = _f32(sd["feature_extractor.conv_layers.0.0.weight"].reshape(C, ks[0]), dev)
    x = torch.empty(B * T0, C, device=dev)
    if ln_mode:
        b0 = _f32(sd["feature_extractor.conv_layers.0.0.bias"], dev) if arch.conv_bias else None
        gam, bet = _f32(sd["feature_extractor.conv_layers.0.2.1.weight"], dev), _f32(sd["feature_extractor.conv_layers.0.2.1.bias"], dev)
        check(L_.sc_conv0_ln_gelu_f32(_p(wav_pad), ldw, _p(w0), _p(b0), _p(gam), _p(bet), 1e-5, _p(x), B, T0, C, _stream()),
              "sc_conv0_ln_gelu_f32")
    else:
        nchunk = 32
        partial = torch.empty(B * nchunk * 66, device=dev, dtype=torch.float64)
        scale, shift = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        gam, bet = _f32(sd["feature_extractor.conv_layers.0.2.weight"], dev), _f32(sd["feature_extractor.conv_layers.0.2.bias"], dev)
        check(L_.sc_conv0_stats(_p(wav_pad), ldw, B, T0, nchunk, _p(partial), _stream()), "sc_conv0_stats")
        check(L_.sc_conv0_finalize(_p(partial), nchunk, _p(w0), _p(gam), _p(bet), B, C, T0, 1e-5, _p(scale), _p(shift), _stream()),
              "sc_conv0_finalize")
        check(L_.sc_conv0_gn_gelu_f32(_p(wav_pad), ldw, _p(w0), _p(scale), _p(shift), _p(x), B, T0, C, _stream()), "sc_conv0_gn_gelu_f32")
    x = x.view(B, T0, C)
    if debug is not None:
        debug["conv"] = [x]

    # ---- conv layers 1..6: a strided Conv1d over channels-last rows is a GEMM with overlapping A rows (lda = s C, K = k C)
    for i in range(1, len(ks)):
        k, s, Ti = ks[i], ss[i], Ts[i]
        cw = sd[f"feature_extractor.conv_layers.{i}.0.weight"]                                  # [C_out, C_in, k]
        w = _f32(cw.permute(0, 2, 1).reshape(cw.shape[0], -1), dev)                               # tap-major [C_out, k C_in]
        bias = _f32(sd[f"feature_extractor.conv_layers.{i}.0.bias"], dev) if arch.conv_bias else None
        y = torch.empty(B, Ti, C, device=dev)
        for b in range(B):
            cols = torch.as_strided(x[b], (Ti, k * C), (s * C, 1))
            ops.sgemm_mfma(cols, w, bias=bias, out=y[b])
        y2 = y.view(B * Ti, C)
        if ln_mode:
            y2 = _ln(y2, _f32(sd[f"feature_extractor.conv_layers.{i}.2.1.weight"], dev), _f32(sd[f"feature_extractor.conv_layers.{i}.2.1.bias"], dev))
        x = ops.gelu_f32(y2).view(B, Ti, C)
        if debug is not None:
            debug["conv"].append(x)

    # ---- feature LayerNorm, projection, fairseq frame mask (:77-85, :32-33)
    f = _ln(x.reshape(B * T, C), _f32(sd["layer_norm.weight"], dev), _f32(sd["layer_norm.bias"], dev))
    xp = _linear(f, _f32(sd["post_extract_proj.weight"], dev), _f32(sd["post_extract_proj.bias"], dev)).view(B, T, D)
    if debug is not None:
        debug["proj"] = xp
    chunk = L // T
    valid = [min(T, -(-l // chunk)) for l in lens]
    feat_len = [min(round(l / arch.downsample_rate), T) for l in lens]
    frame = torch.arange(T, device=dev).unsqueeze(0)
    pad_mask = frame >= torch.tensor(valid, device=dev).unsqueeze(1)                              # [B, T] True = padded frame
    xp = xp.masked_fill(pad_mask.unsqueeze(-1), 0.0)                                               # data movement (index_put), no arithmetic

    # ---- pos_conv (grouped, k = 128, padding 64, SamePad) + GELU + residual (+ encoder LayerNorm, post-LN order) (:29-40)
    G, Kp = arch.pos_conv_groups, arch.pos_conv_kernel
    Dg = D // G
    if "encoder.pos_conv.0.weight" in sd:
        pos_w = sd["encoder.pos_conv.0.weight"].float()
    else:
        g_, v_ = sd["encoder.pos_conv.0.weight_g"].float(), sd["encoder.pos_conv.0.weight_v"].float()
        pos_w = g_ * v_ / v_.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()                          # weight_norm(dim = 2): a parameter fold
    wg = _f32(pos_w.reshape(G, Dg, Dg, Kp).permute(0, 1, 3, 2).reshape(G, Dg, Kp * Dg), dev)       # [g][co][tap ci]
    pos_b = _f32(sd["encoder.pos_conv.0.bias"], dev)
    slab = torch.zeros(G, B, T + Kp, Dg, device=dev)
    slab[:, :, Kp // 2: Kp // 2 + T] = xp.view(B, T, G, Dg).permute(2, 0, 1, 3)
    u = torch.empty(B, T, D, device=dev)
    for b in range(B):
        for g in range(G):
            cols = torch.as_strided(slab[g, b], (T, Kp * Dg), (Dg, 1))
            ops.sgemm_mfma(cols, wg[g], bias=pos_b[g * Dg: (g + 1) * Dg].contiguous(), out=u[b, :, g * Dg: (g + 1) * Dg])
    gel = ops.gelu_f32(u.view(B * T, D))
    x2 = xp.reshape(B * T, D).contiguous()
    if arch.layer_norm_first:
        h = _add(x2, gel)
    else:
        h = _ln(gel, _f32(sd["encoder.layer_norm.weight"], dev), _f32(sd["encoder.layer_norm.bias"], dev), res=x2)
    hidden = [h.view(B, T, D)]

    # ---- encoder layers (fairseq TransformerSentenceEncoderLayer, :49-53), key padding mask = the frame mask
    Tk = (T + 3) // 4 * 4
    key_mask = torch.ones(B, Tk, device=dev, dtype=torch.uint8)
    key_mask[:, :T] = pad_mask.to(torch.uint8)
    scores = torch.zeros(H, T, Tk, device=dev)
    probs = torch.empty(H, T, Tk, device=dev)
    for i in range(arch.layers):
        p = f"encoder.layers.{i}."
        Wqkv = _f32(torch.cat([sd[p + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], 0), dev)
        bqkv = _f32(torch.cat([sd[p + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], 0), dev)
        ln1 = (_f32(sd[p + "self_attn_layer_norm.weight"], dev), _f32(sd[p + "self_attn_layer_norm.bias"], dev))
        ln2 = (_f32(sd[p + "final_layer_norm.weight"], dev), _f32(sd[p + "final_layer_norm.bias"], dev))
        a_in = _ln(h, *ln1) if arch.layer_norm_first else h
        qkv = _linear(a_in, Wqkv, bqkv).view(B, T, 3 * D)
        ctx = torch.empty(B, T, D, device=dev)
        for b in range(B):
            q, k_, v = qkv[b, :, :D], qkv[b, :, D: 2 * D], qkv[b, :, 2 * D:]
            # S[h] = dh^-1/2 Q_h K_h^T  (heads = the batch dimension: stride dh inside a row)
            ops.sgemm_ex(q, (3 * D, 1, dh), k_, (3 * D, 1, dh), scores, Tk, T, T, dh, nbatch=H, scz=T * Tk, alpha=dh ** -0.5)
            check(L_.sc_softmax_fwd_f32(_p(scores), _p(key_mask[b]), _p(probs), H * T, Tk, H * T, 1.0, _stream()), "sc_softmax_fwd_f32")
            # O_h = P_h V_h  -> columns h dh .. of ctx[b]
            ops.sgemm_ex(probs, (Tk, 1, T * Tk), v, (1, 3 * D, dh), ctx[b], D, T, dh, T, nbatch=H, scz=dh)
        attn = _linear(ctx.view(B * T, D), _f32(sd[p + "self_attn.out_proj.weight"], dev), _f32(sd[p + "self_attn.out_proj.bias"], dev))
        fc1 = (_f32(sd[p + "fc1.weight"], dev), _f32(sd[p + "fc1.bias"], dev))
        fc2 = (_f32(sd[p + "fc2.weight"], dev), _f32(sd[p + "fc2.bias"], dev))
        if arch.layer_norm_first:                       # pre-LN (large): x = x + attn(LN1 x) ; x = x + ffn(LN2 x)
            h = _add(h, attn)
            ff = _linear(ops.gelu_f32(_linear(_ln(h, *ln2), *fc1)), *fc2)
            h = _add(h, ff)
        else:                                           # post-LN (base): x = LN1(x + attn(x)) ; x = LN2(x + ffn(x))
            h = _ln(attn, *ln1, res=h)
            ff = _linear(ops.gelu_f32(_linear(h, *fc1)), *fc2)
            h = _ln(ff, *ln2, res=h)
        hidden.append(h.view(B, T, D))
    return hidden, feat_len
