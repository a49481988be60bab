"""Trainable HuBERT front end (scope row f2, the part ``audio_encoder.trainable: true`` adds to hubert_train.TrainableLayers):
conv feature extractor, feature LayerNorm, post_extract_proj, the positional conv block and the encoder LayerNorm
(avssl/module/speech_encoder_plus.py:556-562: with ``trainable`` and no reinit / unfreeze list EVERY encoder parameter trains;
the arithmetic is fairseq's ConvFeatureExtractionModel / HubertModel / TransformerEncoder, :29-40, :75-87).

Forward (train mode) keeps what the backward needs; backward is a manual chain on the library's kernels, entered from
TrainableLayers.backward with the gradient of ``hidden[0]``:

    encoder LayerNorm'  ->  pos_conv: GELU', grouped Toeplitz GEMM with the flipped kernel (input gradient, + the residual path in
    the epilogue), per-group weight gradient over the halo-padded slab  ->  padded-frame mask  ->  dropout_input mask
    ->  post_extract_proj dgrad / wgrad  ->  feature LayerNorm'  ->  x feature_grad_mult (fairseq GradMultiply)
    ->  conv layers 6 .. 1: GELU' (large: + LayerNorm'), weight gradient = dy^T . (strided im2col VIEW of the layer input, no copy
        of the windows), input gradient = dy . W followed by the overlap-add of the k = 3 / stride 2 windows (k = 2 windows do
        not overlap: a reshape)
    ->  conv layer 0 (C_in = 1, k = 10) with its GroupNorm (base) / LayerNorm (large): own forward / backward kernel pairs
        (csrc/frontend.hip: sc_conv0_gn_gelu / sc_conv0_gn_bwd, sc_conv0_ln_gelu / sc_conv0_ln_bwd); parameter gradients only.

pos_conv is weight-normalised in fairseq (``weight_g``, ``weight_v``, norm over dims 0, 1): those two tensors are the
parameters; the folded weight and the chain rule back to them are a few small torch ops on 4.7 M elements.
No shipped recipe trains HuBERT; parity: tests/test_gpu_model.py::test_fully_trainable_hubert_gradients_vs_oracle.
"""
from typing import Dict

import torch
from torch import nn

from . import ops


def _key(name: str) -> str:
    return name.replace(".", "_")


class TrainableFrontend(nn.Module):
    def __init__(self, arch, sd: Dict[str, torch.Tensor], device):
        super().__init__()
        self.arch = arch
        a = arch
        names = []
        for i in range(len(a.conv_kernels)):
            names.append(f"feature_extractor.conv_layers.{i}.0.weight")
            if a.conv_bias:
                names.append(f"feature_extractor.conv_layers.{i}.0.bias")
            if a.extractor_mode == "default" and i == 0:
                names += [f"feature_extractor.conv_layers.{i}.2.weight", f"feature_extractor.conv_layers.{i}.2.bias"]
            if a.extractor_mode == "layer_norm":
                names += [f"feature_extractor.conv_layers.{i}.2.1.weight", f"feature_extractor.conv_layers.{i}.2.1.bias"]
        names += ["layer_norm.weight", "layer_norm.bias", "post_extract_proj.weight", "post_extract_proj.bias",
                  "encoder.pos_conv.0.bias", "encoder.layer_norm.weight", "encoder.layer_norm.bias"]
        self.p = nn.ParameterDict()
        self.fairseq_names = {}
        for n in names:
            self.p[_key(n)] = nn.Parameter(sd[n].detach().float().clone().to(device))
            self.fairseq_names[_key(n)] = n
        # weight-normalised positional conv: parameters weight_g / weight_v (norm over dims 0, 1 per tap: torch weight_norm dim=2)
        if "encoder.pos_conv.0.weight_g" in sd:
            g, v = sd["encoder.pos_conv.0.weight_g"].detach().float(), sd["encoder.pos_conv.0.weight_v"].detach().float()
        else:                                # a folded weight: start from g = |w|, v = w (the same function)
            v = sd["encoder.pos_conv.0.weight"].detach().float()
            g = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        for n, t in (("encoder.pos_conv.0.weight_g", g), ("encoder.pos_conv.0.weight_v", v)):
            self.p[_key(n)] = nn.Parameter(t.clone().to(device))
            self.fairseq_names[_key(n)] = n
        self._gen = None
        self._c = {}

    def P(self, name: str) -> nn.Parameter:
        return self.p[_key(name)]

    @staticmethod
    def _gacc(p: torch.Tensor) -> torch.Tensor:
        if p.grad is None:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return p.grad

    def pos_weight(self) -> torch.Tensor:
        g, v = self.P("encoder.pos_conv.0.weight_g"), self.P("encoder.pos_conv.0.weight_v")
        return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()

    def refresh(self) -> None:
        """bf16 working copies of the masters in the kernels' layouts, rebuilt after every optimiser step."""
        from .optim import param_generation
        gen = (param_generation(),) + tuple((p.data_ptr(), p._version) for p in self.p.values())
        if gen == self._gen:
            return
        a, c = self.arch, {}
        bf = ops.bf16_copy          # one copy kernel whatever the view's strides
        f32 = lambda t: ops.aligned16(t.detach().float().contiguous())
        for i in range(1, len(a.conv_kernels)):
            w = self.P(f"feature_extractor.conv_layers.{i}.0.weight").detach()           # [C_out, C_in, k]
            c[f"conv{i}_w"] = bf(w.permute(0, 2, 1)).view(w.shape[0], -1)                  # tap-major [C_out, k C_in]
            c[f"conv{i}_wT"] = bf(w.permute(2, 1, 0)).view(-1, w.shape[0])                 # its transpose [k C_in, C_out]
            if a.conv_kernels[i] == 3 and a.conv_strides[i] == 2:
                # input gradient of an even input row 2 m = window m's tap 0 + window m - 1's tap 2: ONE product over the row pair
                # [du[m - 1] | du[m]] (K = 2 C_out) against [W_2^T | W_0^T]; odd rows 2 m + 1 take tap 1 of window m (backward_frontend)
                c[f"conv{i}_wT_even"] = bf(torch.cat([w[:, :, 2].t(), w[:, :, 0].t()], dim=1))       # [C_in, 2 C_out]
                c[f"conv{i}_wT_odd"] = bf(w[:, :, 1].t())                                          # [C_in, C_out]
            c[f"conv{i}_b"] = f32(self.P(f"feature_extractor.conv_layers.{i}.0.bias")) if a.conv_bias else None
            if a.extractor_mode == "layer_norm":
                c[f"conv{i}_g"] = f32(self.P(f"feature_extractor.conv_layers.{i}.2.1.weight"))
                c[f"conv{i}_beta"] = f32(self.P(f"feature_extractor.conv_layers.{i}.2.1.bias"))
        c["ln_feat_g"], c["ln_feat_b"] = f32(self.P("layer_norm.weight")), f32(self.P("layer_norm.bias"))
        pw = self.P("post_extract_proj.weight").detach()
        (c["proj_w"], c["proj_wT"]), c["proj_b"] = ops.weight_copies(pw), f32(self.P("post_extract_proj.bias"))
        D, G, Kp = a.embed_dim, a.pos_conv_groups, a.pos_conv_kernel
        Dg = D // G
        w = self.pos_weight().detach().reshape(G, Dg, Dg, Kp)                              # [g][co][ci][tap]
        c["pos_w"] = bf(w.permute(0, 1, 3, 2)).view(G, Dg, Kp * Dg)                         # forward: [g][co][tap ci]
        c["pos_w_flip"] = bf(w.flip(3).permute(0, 2, 3, 1)).view(G, Dg, Kp * Dg)            # dgrad:   [g][ci][tap' co], tap' = K-1-tap
        c["pos_b"] = f32(self.P("encoder.pos_conv.0.bias"))
        c["ln_enc_g"], c["ln_enc_b"] = f32(self.P("encoder.layer_norm.weight")), f32(self.P("encoder.layer_norm.bias"))
        self._c, self._gen = c, gen

    # ------------------------------------------------------------------------------------------------- forward
    def forward_frontend(self, pl, L: int, p_in: float, p_res: float, seed_in: int, seed_enc: int) -> None:
        """waveform in pl.wav_pad -> pl.hidden[0]; keeps the activations of every stage in ``pl.front``."""
        a, c = self.arch, self._c
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        C, D = a.conv_dim, a.embed_dim
        dev = pl.wav_pad.device
        st = pl.front = {"p_in": p_in, "p_res": p_res, "seed_in": seed_in, "seed_enc": seed_enc}
        ln_mode = a.extractor_mode == "layer_norm"
        # ---- conv layer 0 (+ GroupNorm, GELU): the frozen path's kernels on the CURRENT parameters (analytic GroupNorm statistics from
        # the waveform's Gram matrix, activation written once); the backward (sc_conv0_gn_bwd) recomputes the pre-activation from the
        # waveform, so nothing but the per-(utterance, channel) scale / shift and the Gram partials is kept.  The "layer_norm"
        # extractor (HuBERT-large) has the same pair of kernels (sc_conv0_ln_gelu / sc_conv0_ln_bwd): statistics per row, nothing kept.
        T0 = pl.T_l[0]
        if ln_mode:
            w0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.0.weight").detach().float().reshape(C, a.conv_kernels[0]).contiguous())
            b0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.0.bias").detach().float().contiguous()) if a.conv_bias else None
            g0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.2.1.weight").detach().float().contiguous())
            be0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.2.1.bias").detach().float().contiguous())
            ops.conv0_layernorm_gelu(pl.wav_pad, w0, b0, g0, be0, pl.R_l[0], pl.conv[0])
            st["conv0_ln"] = (w0, b0, g0, be0)
        else:
            w0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.0.weight").detach().float().reshape(C, a.conv_kernels[0]).contiguous())
            g0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.2.weight").detach().float().contiguous())
            be0 = ops.aligned16(self.P("feature_extractor.conv_layers.0.2.bias").detach().float().contiguous())
            st["conv0"] = (w0, g0, be0, ops.conv0_groupnorm_gelu(pl.wav_pad, w0, g0, be0, T0, pl.R_l[0], pl.conv[0]))
        # ---- conv layers 1 .. 6 on the strided-row GEMM, pre-activations kept
        st["u"], st["n"] = {}, {}
        for i in range(1, len(a.conv_kernels)):
            k, s = a.conv_kernels[i], a.conv_strides[i]
            rows = B * pl.R_l[i]
            u = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
            st["u"][i] = u
            if ln_mode:
                ops.gemm_raw(pl.conv[i - 1], s * C, c[f"conv{i}_w"], k * C, u, C, rows, C, k * C, bias=c[f"conv{i}_b"], alg_rows=B * pl.T_l[i],
                             tap_c=C if (k == 3 and s == 2) else 0)
                n = ops.layernorm_bf16(u, c[f"conv{i}_g"], c[f"conv{i}_beta"])
                st["n"][i] = n
                ops.act_bf16(n, 1, out=pl.conv[i][:rows])
            else:     # one launch: the pre-activation u (kept) and gelu(u) (sc_gemm_args.aux_mode 1)
                ops.gemm_raw(pl.conv[i - 1], s * C, c[f"conv{i}_w"], k * C, pl.conv[i], C, rows, C, k * C, bias=c[f"conv{i}_b"], act=1,
                             alg_rows=B * pl.T_l[i], tap_c=C if (k == 3 and s == 2) else 0, aux=u, aux_mode=1)
        # ---- feature LayerNorm, projection (+ dropout_input)
        ops.layernorm_bf16(pl.conv[-1][:M], c["ln_feat_g"], c["ln_feat_b"], out=pl.feat_ln)
        ops.linear_bf16(pl.feat_ln, c["proj_w"], c["proj_b"], out=pl.x_proj, alg_rows=B * T, drop_p=p_in, drop_seed=seed_in)
        # ---- zero padded frames, pos_conv (pre-activation kept) + GELU + residual, encoder LayerNorm
        G, Kp = a.pos_conv_groups, a.pos_conv_kernel
        Dg, Rp = D // G, R + 2 * pl.halo
        ops.posconv_prep(pl.x_proj, pl.valid, pl.xz, pl.xg, B, R, D, G, pl.halo)
        u_pos = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        ops.gemm_raw(pl.xg, Dg, c["pos_w"], Kp * Dg, u_pos, D, R, Dg, Kp * Dg, bias=c["pos_b"], nb1=G, nb2=B,
                     sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, R * D), sBias=(Dg, 0), alg_rows=T)
        st["u_pos"] = u_pos
        y = ops.act_bf16(u_pos, 1)
        pre = st["pre"] = (pl.xz.float() + y.float()).to(torch.bfloat16)          # x + gelu(pos_conv(x))  (one elementwise pass)
        if a.layer_norm_first:
            pl.hidden[0].copy_(pre)
        else:
            ops.layernorm_bf16(pre, c["ln_enc_g"], c["ln_enc_b"], out=pl.hidden[0])
        if p_res > 0:
            ops.dropout_bf16(pl.hidden[0], p_res, seed_enc, out=pl.hidden[0])

    # ------------------------------------------------------------------------------------------------- backward
    def backward_frontend(self, pl, dh0: torch.Tensor) -> None:
        """dh0 [M, D] bf16: gradient of hidden[0] (rows t >= T are zero)."""
        a, c, st = self.arch, self._c, pl.front
        B, R, M, T = pl.B, pl.R, pl.M, pl.T
        C, D = a.conv_dim, a.embed_dim
        dev = dh0.device
        ln_mode = a.extractor_mode == "layer_norm"
        acc = lambda name, g: self._gacc(self.P(name)).add_(g.reshape(self.P(name).shape))
        if st["p_res"] > 0:
            dh0 = ops.dropout_bf16(dh0, st["p_res"], st["seed_enc"])
        # ---- encoder LayerNorm
        if a.layer_norm_first:
            dpre = dh0
        else:
            # (the LayerNorm parameter gradients are added by the reduction itself)
            dpre = ops.layernorm_bwd(st["pre"], dh0, c["ln_enc_g"], 1e-5,
                                     acc=(self._gacc(self.P("encoder.layer_norm.weight")), self._gacc(self.P("encoder.layer_norm.bias"))))
        # ---- pos_conv:  pre = xz + gelu(u),  u = conv(xz) + b
        G, Kp = a.pos_conv_groups, a.pos_conv_kernel
        Dg, halo = D // G, pl.halo
        Rp = R + 2 * halo
        du = ops.act_bf16(st["u_pos"], 1, df=dpre)
        # slab of du (group-major, halo-padded, frames of every utterance kept: the mask argument is "all R frames valid")
        all_valid = torch.full((B,), R, device=dev, dtype=torch.int32)
        du_z = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        # + (halo + 1) slab rows of zero slack behind the last group: the weight gradient reads the slab advanced by `halo` rows, the
        # input gradient advanced by one row
        dug_buf = torch.zeros((G * B * Rp + halo + 1) * Dg, device=dev, dtype=torch.bfloat16)
        dug = dug_buf[: G * B * Rp * Dg].view(G, B, Rp, Dg)
        ops.posconv_prep(du, all_valid, du_z, dug, B, R, D, G, halo)
        # weight gradient: gw[g][co][tap Dg + ci] = sum_m du[g][m][co] xg[g][m + tap][ci] over the slab rows m = b Rp + t (window index),
        # du laid out with the frames at row offset 0 = the du slab advanced by `halo` rows; its halo rows (zero) meet the windows that
        # straddle two utterances.  One kernel for all groups and taps (csrc/posconv_bwd.hip).
        gw = ops.posconv_wgrad(dug_buf[halo * Dg:], pl.xg, G, B * Rp, Dg, Kp)
        gb = torch.empty(D, device=dev, dtype=torch.float32)
        ops.colsum_bf16(du, gb)
        acc("encoder.pos_conv.0.bias", gb)
        # chain rule to weight_g / weight_v through the fold (small tensors: torch autograd)
        dW = gw.view(G, Dg, Kp, Dg).permute(0, 1, 3, 2).reshape(D, Dg, Kp)             # [co][ci][tap]
        with torch.enable_grad():
            w = self.pos_weight()
        gg, gv = torch.autograd.grad(w, [self.P("encoder.pos_conv.0.weight_g"), self.P("encoder.pos_conv.0.weight_v")], dW)
        acc("encoder.pos_conv.0.weight_g", gg)
        acc("encoder.pos_conv.0.weight_v", gv)
        # input gradient: dx[s] = sum_tap' du_slab[s + 1 + tap'] . W_flip[tap']  (+ the residual path dpre in the epilogue)
        dxz = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        ops.gemm_raw(dug_buf[Dg:], Dg, c["pos_w_flip"], Kp * Dg, dxz, D, R, Dg, Kp * Dg, residual=dpre, ldr=D, nb1=G, nb2=B,
                     sA=(B * Rp * Dg, Rp * Dg), sW=(Dg * Kp * Dg, 0), sC=(Dg, R * D), sR=(Dg, R * D), alg_rows=T)
        # padded frames were zeroed in the forward (x[padding_mask] = 0): no gradient to them
        dx_proj = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        ops.posconv_prep(dxz, pl.valid, dx_proj, dug, B, R, D, G, halo)
        if st["p_in"] > 0:
            dx_proj = ops.dropout_bf16(dx_proj, st["p_in"], st["seed_in"])
        # ---- post_extract_proj, feature LayerNorm
        ops.wgrad_bf16(dx_proj, pl.feat_ln, self._gacc(self.P("post_extract_proj.weight")), self._gacc(self.P("post_extract_proj.bias")))
        dfl = ops.linear_bf16(dx_proj, c["proj_wT"])
        df = ops.layernorm_bwd(pl.conv[-1][:M], dfl, c["ln_feat_g"], 1e-5,
                               acc=(self._gacc(self.P("layer_norm.weight")), self._gacc(self.P("layer_norm.bias"))))
        fgm = float(getattr(a, "feature_grad_mult", 1.0))
        if fgm != 1.0:                       # fairseq GradMultiply on the extractor's output
            df = (df.float() * fgm).to(torch.bfloat16)
        # ---- conv layers 6 .. 1.  "default" extractor (conv -> GELU): the GELU' of the layer BELOW rides on the step that produces its
        # output gradient - the epilogue of the input-gradient GEMM (k = stride: its windows are a reshape) or the overlap-add pass
        # (k = 3, stride 2) - so du of the next iteration arrives ready (no activation-sized pass of its own)
        du_ready = None

        def rows_buf(n: int) -> torch.Tensor:
            """[n, C] bf16 rows with ONE zero row allocated in front of them: the k = 3 / stride 2 layers read the row pair
            [du[m - 1] | du[m]] in place, and window -1 does not exist"""
            buf = torch.empty(1 + n, C, device=dev, dtype=torch.bfloat16)
            buf[0].zero_()
            return buf[1:]

        for i in range(len(a.conv_kernels) - 1, 0, -1):
            k, s = a.conv_kernels[i], a.conv_strides[i]
            rows = B * pl.R_l[i]
            u = st["u"][i]
            fuse_below = (not ln_mode) and i >= 2
            if ln_mode:
                dn = ops.act_bf16(st["n"][i], 1, df=df)
                du, dg, db = ops.layernorm_bwd(u, dn, c[f"conv{i}_g"], 1e-5, want_param_grads=True, out=rows_buf(rows))
                acc(f"feature_extractor.conv_layers.{i}.2.1.weight", dg)
                acc(f"feature_extractor.conv_layers.{i}.2.1.bias", db)
            elif du_ready is not None:
                du, du_ready = du_ready, None
            else:
                du = ops.act_bf16(u, 1, df=df, out=rows_buf(rows))
            # weight gradient: dy^T . im2col VIEW of the layer input (rows overlap in memory: lda = s C < K = k C)
            cols = torch.as_strided(pl.conv[i - 1], (rows, k * C), (s * C, 1))
            gw = torch.empty(C, k * C, device=dev, dtype=torch.float32)
            gbias = torch.empty(C, device=dev, dtype=torch.float32) if a.conv_bias else None
            ops.wgrad_bf16(du, cols, gw, gbias, beta=0.0)
            acc(f"feature_extractor.conv_layers.{i}.0.weight", gw.view(C, k, C).permute(0, 2, 1))
            if a.conv_bias:
                acc(f"feature_extractor.conv_layers.{i}.0.bias", gbias)
            # input gradient: the windows of du . W back onto the rows they were cut from; the GELU' of the layer below in the epilogue
            dx = rows_buf(rows * s)
            kw = dict(act=1, aux_mode=2) if fuse_below else {}
            if k == s:                                                                     # non-overlapping windows: a reshape
                ops.gemm_raw(du, C, c[f"conv{i}_wT"], C, dx.view(rows, k * C), k * C, rows, k * C, C,
                             aux=st["u"][i - 1].view(rows, k * C) if fuse_below else None, **kw)
            else:
                # k = 3, stride 2: row 2 m gets tap 0 of window m and tap 2 of window m - 1, row 2 m + 1 tap 1 of window m - two products
                # that write the interleaved rows directly (row stride 2 C), the first over the overlapping row pairs of du (lda = C,
                # K = 2 C: the conv forward's strided-row trick).  No [rows, 3 C] window gradients, no overlap-add pass (round 3 wrote
                # 3.1 GB of them for conv 1 and read them back).
                assert k == 3 and s == 2
                pair = torch.as_strided(du, (rows, 2 * C), (C, 1), du.storage_offset() - C)
                dx2 = dx.view(rows, 2 * C)
                ub = st["u"][i - 1].view(rows, 2 * C) if fuse_below else None
                ops.gemm_raw(pair, C, c[f"conv{i}_wT_even"], 2 * C, dx2[:, :C], 2 * C, rows, C, 2 * C, aux=ub[:, :C] if fuse_below else None, **kw)
                ops.gemm_raw(du, C, c[f"conv{i}_wT_odd"], C, dx2[:, C:], 2 * C, rows, C, C, aux=ub[:, C:] if fuse_below else None, **kw)
            if fuse_below:
                du_ready = dx
            else:
                df = dx
        # ---- conv layer 0: parameter gradients only (the input is the waveform)
        T0 = pl.T_l[0]
        if "conv0" in st:
            w0, g0, be0, saved = st["conv0"]
            dW0, dg0, db0 = ops.conv0_groupnorm_gelu_bwd(pl.wav_pad, w0, g0, be0, saved, df, T0, pl.R_l[0])
            acc("feature_extractor.conv_layers.0.0.weight", dW0)
            acc("feature_extractor.conv_layers.0.2.weight", dg0)
            acc("feature_extractor.conv_layers.0.2.bias", db0)
        else:                                  # "layer_norm" extractor (HuBERT-large): sc_conv0_ln_bwd
            w0, b0, g0, be0 = st["conv0_ln"]
            dW0, db0, dg0, dbe0 = ops.conv0_layernorm_gelu_bwd(pl.wav_pad, w0, b0, g0, be0, df, T0, pl.R_l[0])
            acc("feature_extractor.conv_layers.0.0.weight", dW0)
            if a.conv_bias:
                acc("feature_extractor.conv_layers.0.0.bias", db0)
            acc("feature_extractor.conv_layers.0.2.1.weight", dg0)
            acc("feature_extractor.conv_layers.0.2.1.bias", dbe0)
        pl.front = None
