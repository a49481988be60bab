"""Continuous integrate-and-fire downsampler of the cascaded+/hybrid+ branches (avssl/module/cif.py:24-311) on the library's
kernels (csrc/cif.hip).  Same constructor keywords, sub-module names (``conv.0``, ``weight_proj.1``) and result-dict keys as the
reference, including the quirks that change numbers: ``nn.Dropout()`` (p = 0.5) in the weight generator, ``MAX_FEAT_LEN = 75``,
alpha clipped to [0, 1], ``quantity_out`` taken before the scaling, train-time tail drop vs inference-time tail firing.

Everything between the weight generator and the downsampled features is device-side, one workgroup per utterance:

    sc_cif_prepare   clip / zero padded frames / quantity / target-length scaling / inclusive scan / keyword count / fired marks
    sc_cif_fwd       one deterministic pass over the frames (the reference: three atomically accumulating scatter_add_ rounds)
    sc_cif_tail      inference: tail firing, rescale, zero the rows past the final count
    sc_cif_bwd + sc_cif_prepare_bwd   gradients to the features and to the weight generator's output

Host reads: NONE while the target-length scaling is on (training before ``scaling_step``): the scaled weights sum to
``target + 1e-5`` by construction, so the keyword count equals ``clip(target, 1, 75)``, which the caller already knows on the host
(``target_lengths_host``); the kernel counts disagreements in ``consistency_flags`` (read them with ``check_flags()`` whenever a
synchronisation is acceptable).  Otherwise the count is data and the shape of the returned tensor needs ONE read (the batch
maximum); the reference reads three values per call (count maximum, extra-fire maximum, the all-zero assertion).
There is no CPU path (the CPU restatement of this maths is oracle/cascaded_ref.py).
"""
import logging
from typing import List, Optional

import torch
from torch import nn

from . import ops

logger = logging.getLogger(__name__)

MAX_FEAT_LEN = 75   # cif.py:11


class _CifFn(torch.autograd.Function):
    """(feats [B,S,C], alpha_raw [B,S]) -> (slots [B,Tc+1,C] fp32, quantity [B]); everything else rides on ``st`` (plain dict)."""

    @staticmethod
    def forward(ctx, feats, alpha_raw, pad, target, st: dict, rows=None):
        # ``rows`` = (head, S): ``feats`` is the attention block's bf16 row buffer [B, P, C] (mha_block.BranchRows.full) and the frames
        # of utterance b are its rows head .. head + S - 1 - read in place, gradient returned in the same layout (zero elsewhere)
        x = feats.detach() if rows is not None else feats.detach().float().contiguous()
        a = alpha_raw.detach().float().contiguous()
        thr, Tc = st["thr"], st["T_clip"]
        r = ops.cif_prepare(a, pad, target, st["apply_scaling"], thr, st["eps"], MAX_FEAT_LEN, Tc, st["flags"])
        out = ops.cif_fwd(x, r["alpha"], r["csum"], Tc, thr) if rows is None else ops.cif_fwd_rows(x, rows[0], rows[1], r["alpha"], r["csum"], Tc, thr)
        ctx.rows = rows
        st.update(r)
        if st["tail"]:                                         # inference-time tail handling
            st["feat_len_pre"] = r["feat_len"].clone()
            st["factor"], st["extend"] = ops.cif_tail(r["alpha"], r["csum"], r["feat_len"], out, Tc, thr, st["tail_thr"], MAX_FEAT_LEN)
        ctx.st = st
        ctx.save_for_backward(x, r["alpha"], r["csum"], r["a_clip"], r["ratio"], r["quantity"], pad)
        ctx.dtypes = (feats.dtype, alpha_raw.dtype)
        ctx.mark_non_differentiable(r["feat_len"])
        return out, r["quantity"], r["feat_len"]

    @staticmethod
    def backward(ctx, g_out, g_q, _):
        x, alpha, csum, a_clip, ratio, quantity, pad = ctx.saved_tensors
        st = ctx.st
        thr, Tc = st["thr"], st["T_clip"]
        g = g_out.float().contiguous()
        if st["tail"]:
            # the rescaled row carries its factor (a constant of the graph: cif.py:262-279 detaches it); rows past the final count
            # were overwritten with zeros
            rows = torch.arange(Tc + 1, device=g.device).unsqueeze(0)
            fl_new, fl_old = st["feat_len"].unsqueeze(1), st["feat_len_pre"].unsqueeze(1)
            scale = torch.where((rows == fl_old) & st["extend"].bool().unsqueeze(1), st["factor"].unsqueeze(1),
                                torch.ones((), device=g.device))
            g = g * (scale * (rows < fl_new)).unsqueeze(-1)
        if ctx.rows is None:
            dx, pa, pb = ops.cif_bwd(x, alpha, csum, g, Tc, thr)
        else:
            sink = st.get("grad_sink")
            dx, pa, pb = ops.cif_bwd_rows(x, ctx.rows[0], ctx.rows[1], alpha, csum, g, Tc, thr, tail_rows=ctx.rows[0] if sink is not None else 0)
        gq = g_q.float().contiguous() if g_q is not None else None
        da = ops.cif_prepare_bwd(pa, pb, a_clip, pad, ratio, quantity, gq, st["scaled"])
        if ctx.rows is not None and sink is not None and ctx.needs_input_grad[1]:
            # the weight head's backward runs after this one (its output alpha is this node's input) and returns the SUM of both
            # gradients of the block's rows: this share rides in its input-gradient GEMM's residual operand instead of an add launch
            sink["dfull"] = dx
            return None, da.to(ctx.dtypes[1]), None, None, None, None
        return dx.to(ctx.dtypes[0]), da.to(ctx.dtypes[1]), None, None, None, None


class _CifHeadFn(torch.autograd.Function):
    """Dropout -> ReLU -> Dropout -> Linear(C, 1) -> Sigmoid of the weight generator on sc_cif_head_fwd / _bwd: y [B, S, C] fp32 (the conv
    output), weight [1, C], bias [1] -> alpha [B, S]."""

    @staticmethod
    def forward(ctx, y, weight, bias, p1, seed1, p2, seed2):
        B, S, C = y.shape
        w = ops.aligned16(weight.detach().float().reshape(C).contiguous())
        b = bias.detach().float().reshape(1).contiguous()
        y2 = y.detach().reshape(B * S, C)
        alpha = ops.cif_head_fwd(y2, w, b, p1, seed1, p2, seed2)
        ctx.save_for_backward(y2, w, alpha)
        ctx.meta = (B, S, C, p1, seed1, p2, seed2, weight.shape)
        return alpha.view(B, S)

    @staticmethod
    def backward(ctx, dalpha):
        y2, w, alpha = ctx.saved_tensors
        B, S, C, p1, seed1, p2, seed2, wshape = ctx.meta
        dy, dw, db = ops.cif_head_bwd(y2, w, alpha, dalpha.float().contiguous().view(-1), p1, seed1, p2, seed2)
        return dy.view(B, S, C), dw.view(wshape), db, None, None, None, None


class _ConvRowsBf16Fn(torch.autograd.Function):
    """Conv1d(C, N, k, stride 1, padding p) over channels-last frames as ONE strided-row GEMM per pass (training: bf16 operands, fp32
    accumulation), without materialising the k shifted copies of the input.

    x [B, T, C] -> rows.  The frames are copied once into a zero-padded bf16 row buffer xp [B, T + 2p, C] (row pitch Tp = T + 2p per
    utterance); output row m = b Tp + t reads the k consecutive rows m .. m + k - 1 of xp (lda = C, K = k C, weights tap-major), so
    y_full [rows, N] carries the outputs at the same pitch and 2p rows of no meaning after every utterance (the caller slices
    [:, :T]; their incoming gradient is zero by construction).  Backward: the input gradient is the same kind of GEMM over the
    zero-padded output gradient with the tap-reversed weights (the 2p meaningless rows between utterances ARE its padding), the weight
    gradient reads dy and the overlapping-row view of xp in place (ops.wgrad_bf16, TN form).  Returns y_full fp32 [rows_pad, N]."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        B, T, C = x.shape
        N, _, k = weight.shape
        Tp = T + 2 * pad
        rows = B * Tp
        rows_pad = (rows + 63) // 64 * 64
        dev = x.device
        xp = torch.zeros(rows_pad + k, C, device=dev, dtype=torch.bfloat16)
        xp[:rows].view(B, Tp, C)[:, pad: pad + T] = x.detach()
        wt = ops.derived(weight, "conv_rows", lambda t: ops.bf16_copy(t.permute(0, 2, 1)).view(N, k * C))   # [N, k C] tap-major
        y = torch.empty(rows_pad, N, device=dev, dtype=torch.float32)
        ops.gemm_raw(xp, C, wt, k * C, y, N, rows_pad, N, k * C, bias=None if bias is None else bias.detach().float().contiguous(),
                     out_f32=True)
        ctx.save_for_backward(xp, weight)
        ctx.meta = (B, T, C, N, k, pad, Tp, rows, rows_pad, x.dtype, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        B, T, C, N, k, pad, Tp, rows, rows_pad, xdtype, has_bias = ctx.meta
        dev = dy.device
        # k zero rows in front (the first utterance's left padding), k behind (the last window of the input-gradient GEMM)
        dyb = torch.zeros(rows_pad + 2 * k, N, device=dev, dtype=torch.bfloat16)
        dyb[k: k + rows_pad] = dy
        dx = gW = gb = None
        if ctx.needs_input_grad[0]:
            # dx[m] = sum_j dy[m + p - j] W_j = sum_jj dyb_row[m + p + 1 + jj] . W_{k-1-jj}     (dyb rows are shifted by k)
            wd = ops.derived(weight, "conv_rows_T", lambda t: ops.bf16_copy(t.flip(2).permute(1, 2, 0)).view(C, k * N))   # [C, (jj, n)]
            dxf = torch.empty(rows_pad, C, device=dev, dtype=torch.bfloat16)
            ops.gemm_raw(dyb[pad + 1:], N, wd, k * N, dxf, C, rows_pad, C, k * N)
            dx = dxf[:rows].view(B, Tp, C)[:, :T].to(xdtype)
        if ctx.needs_input_grad[1]:
            cols = torch.as_strided(xp, (rows_pad, k * C), (C, 1))                                 # im2col VIEW: rows overlap
            g2 = torch.empty(N, k * C, device=dev, dtype=torch.float32)
            gb = torch.empty(N, device=dev, dtype=torch.float32) if (has_bias and ctx.needs_input_grad[2]) else None
            ops.wgrad_bf16(dyb[k: k + rows_pad], cols, g2, gb, beta=0.0)
            gW = g2.view(N, k, C).permute(0, 2, 1)
        elif has_bias and ctx.needs_input_grad[2]:
            gb = dy[:rows].sum(0)
        return dx, gW, gb, None


class _WeightHeadRowsFn(torch.autograd.Function):
    """The CIF weight generator (Conv1d(C, C, k, 'same') -> Dropout -> ReLU -> Dropout -> Linear(C, 1) -> Sigmoid, cif.py:44-60) over
    the attention block's bf16 output rows READ IN PLACE: ``full`` [B, P, C] (mha_block.BranchRows: frames of utterance b at rows
    head .. head + S - 1, every other row of the surrounding flat buffer zero - that IS the conv's zero padding) -> alpha [B, S].

    forward   y[b P + t] = sum_j x[b, t + j - p] W_j + bias : ONE strided-row GEMM (lda = C, K = k C) whose A operand starts p rows in
              front of frame 0; the weight-head row kernel on y.
    backward  the head kernel writes d y as bf16 rows between k zero rows; the conv's input gradient is the same kind of GEMM over
              them with the tap-reversed weights, stored ``head`` rows further down so that it lands in the layout of ``full``
              (rows that are not frames zeroed by one small launch); weight gradient in place (TN form) from the same views.
    No padded copy of the frames, no fp32 copy, no cast or slice of a gradient."""

    @staticmethod
    def forward(ctx, full, conv_w, conv_b, lin_w, lin_b, head, S, pd, p1, seed1, p2, seed2, sink=None):
        B, P, C = full.shape
        N, _, k = conv_w.shape
        M = B * P
        dev = full.device
        A = torch.as_strided(full.detach(), (M, C), (C, 1), full.storage_offset() + (head - pd) * C)
        wt = ops.derived(conv_w, "conv_rows", lambda t: ops.bf16_copy(t.permute(0, 2, 1)).view(N, k * C))   # [N, k C] tap-major
        y = torch.empty(M, N, device=dev, dtype=torch.float32)
        ops.gemm_raw(A, C, wt, k * C, y, N, M, N, k * C, bias=None if conv_b is None else conv_b.detach().float().contiguous(), out_f32=True)
        w = ops.aligned16(lin_w.detach().float().reshape(N).contiguous())
        b = lin_b.detach().float().reshape(1).contiguous()
        alpha = ops.cif_head_fwd(y, w, b, p1, seed1, p2, seed2)
        ctx.save_for_backward(full, conv_w, y, w, alpha)
        ctx.meta = (B, P, C, N, k, pd, head, S, p1, seed1, p2, seed2, lin_w.shape, conv_b is not None)
        ctx.params = (conv_b, lin_w, lin_b)
        ctx.sink = sink                        # dict shared with _CifFn: its gradient of ``full`` is handed over here (see its backward)
        return alpha.view(B, P)[:, :S]

    @staticmethod
    def backward(ctx, dalpha):
        full, conv_w, y, w, alpha = ctx.saved_tensors
        B, P, C, N, k, pd, head, S, p1, seed1, p2, seed2, wshape, has_bias = ctx.meta
        M = B * P
        dev = y.device
        da = torch.zeros(B, P, device=dev, dtype=torch.float32)
        da[:, :S] = dalpha
        # d y as bf16 rows, k zero rows in front (the first utterance's left padding) and behind (the last window of the GEMM below)
        dyb = torch.empty(M + 2 * k, N, device=dev, dtype=torch.bfloat16)
        ops.rows_zero_pad(dyb, k, 1, M, 0, M, k)
        # the head's and the conv bias' gradients are added straight into the optimiser's buffers where they exist (ops.grad_target)
        conv_b, lin_w, lin_b = ctx.params
        t_cb, t_lw, t_lb = (ops.grad_target(q) if q is not None else None for q in (conv_b, lin_w, lin_b))
        direct = t_lw is not None and t_lb is not None and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]
        _, dw, db = ops.cif_head_bwd(y, w, alpha, da.view(-1), p1, seed1, p2, seed2, dy_out=dyb[k: k + M],
                                     acc=(t_lw.view(-1), t_lb.view(-1)) if direct else None)
        dfull = gW = gb = None
        if ctx.needs_input_grad[0]:
            # dx[m] = sum_jj dyb_row[m + p + 1 + jj] . W_{k-1-jj}; row m = b P + t is frame t, i.e. row head + t of ``full``
            wd = ops.derived(conv_w, "conv_rows_T", lambda t: ops.bf16_copy(t.flip(2).permute(1, 2, 0)).view(C, k * N))
            flat = torch.empty(M + head, C, device=dev, dtype=torch.bfloat16)
            other = ctx.sink.pop("dfull", None) if ctx.sink is not None else None
            if other is not None:              # integrate-and-fire's share of d full (same layout, ``head`` zero rows behind it)
                res = torch.as_strided(other, (M, C), (C, 1), other.storage_offset() + head * C)
                ops.gemm_raw(dyb[pd + 1:], N, wd, k * N, flat[head:], C, M, C, k * N, residual=res, ldr=C)
            else:
                ops.gemm_raw(dyb[pd + 1:], N, wd, k * N, flat[head:], C, M, C, k * N)
            ops.rows_zero_pad(flat, 0, B, P, head, head + S, head)
            dfull = flat[:M].view(B, P, C)
        if ctx.needs_input_grad[1]:
            cols = torch.as_strided(full, (M, k * C), (C, 1), full.storage_offset() + (head - pd) * C)       # im2col VIEW: rows overlap
            g2 = torch.empty(N, k * C, device=dev, dtype=torch.float32)
            want_b = has_bias and ctx.needs_input_grad[2]
            gb = torch.empty(N, device=dev, dtype=torch.float32) if (want_b and t_cb is None) else None
            ops.wgrad_bf16(dyb[k: k + M], cols, g2, gb, beta=0.0)
            if want_b and t_cb is not None:
                ops.colsum_bf16(dyb[k: k + M], t_cb, beta=1.0)
            gW = g2.view(N, k, C).permute(0, 2, 1)
        return dfull, gW, gb, None if direct else dw.view(wshape), None if direct else db, None, None, None, None, None, None, None, None


class CIF(nn.Module):
    def __init__(self, cif_threshold=1.0, cif_output_dim=768, encoder_embed_dim=768, produce_weight_type="conv",
                 num_layer=1, conv_cif_width=3, conv_cif_dropout=0.1, apply_scaling=True, apply_tail_handling=True,
                 tail_handling_firing_threshold=0.5, scaling_step=-1, **config):
        super().__init__()
        self.cif_threshold = cif_threshold
        self.cif_output_dim = cif_output_dim
        self.encoder_embed_dim = encoder_embed_dim
        self.produce_weight_type = produce_weight_type
        self.conv_cif_width = conv_cif_width
        self.conv_cif_dropout = conv_cif_dropout
        self.apply_scaling = apply_scaling
        self.apply_tail_handling = apply_tail_handling
        self.tail_handling_firing_threshold = tail_handling_firing_threshold
        self.scaling_step = scaling_step
        self.num_layer = num_layer
        if produce_weight_type != "conv":
            raise NotImplementedError("only produce_weight_type='conv' is runnable in the reference (cif.py:116-135)")
        if cif_output_dim != encoder_embed_dim:
            raise NotImplementedError("cif_output_dim != encoder_embed_dim is not used by any shipped config")
        layers = []
        for _ in range(num_layer):
            layers += [nn.Conv1d(encoder_embed_dim, encoder_embed_dim, conv_cif_width, stride=1,
                                 padding=int(conv_cif_width / 2)), nn.Dropout(), nn.ReLU()]
        self.conv = nn.Sequential(*layers)
        self.weight_proj = nn.Sequential(nn.Dropout(), nn.Linear(encoder_embed_dim, 1), nn.Sigmoid())
        # sc_cif_prepare's counters (include/speechclip_hip.h): [0] utterances with a positive weight sum, [1] keyword counts that differ
        # from clip(target, 1, 75) while the output was sized from the host-side targets, [2] calls, [3] CALLS in which no utterance
        # had a positive weight sum (the reference asserts on every call that there is one), [6] utterances whose all-zero weights could
        # not be rescaled to the target, [4] / [5] the kernel's scratch
        self.register_buffer("consistency_flags", torch.zeros(8, dtype=torch.int32), persistent=False)

    def check_flags(self) -> dict:
        """Synchronises.  Raises like the reference's per-call assertion (avssl/module/cif.py:121) if ANY call since the last reset saw
        only all-zero weights: the kernel keeps that indicator per call (a cumulative "some utterance was positive once"
        let one early healthy call mask a collapsed weight generator for ever)."""
        pos, mism, calls, zero_calls, _, _, zero_utts, _ = self.consistency_flags.tolist()
        assert zero_calls == 0, f"alphas are all zero in {zero_calls} of {calls} CIF calls"
        return {"positive_utterances": pos, "count_mismatches": mism, "calls": calls, "all_zero_calls": zero_calls,
                "zero_quantity_utterances": zero_utts}

    def rows_usable(self, br) -> bool:
        """Can this call read the attention block's output rows (mha_block.BranchRows) in place?  Training, ONE weight-conv layer with
        'same' padding over the block's width, and enough zero rows around the buffer for the conv's reach."""
        last = self.conv[-3]
        k, pd = last.kernel_size[0], last.padding[0]
        return (br is not None and self.training and len(self.conv) == 3 and last.stride[0] == 1 and k == 2 * pd + 1 and
                last.in_channels == br.D == last.out_channels and br.D % 64 == 0 and br.lead + br.head >= pd and br.trail >= k and
                br.P - br.head - br.S >= pd and br.S <= 2048)

    def forward(self, input_dict, target_lengths=None, eps=1e-5, target_lengths_host: Optional[List[int]] = None):
        br = input_dict.get("audio_feat_rows", None)           # mha_block.BranchRows: the frames as the attention block left them
        if br is not None and not self.rows_usable(br):
            input_dict = dict(input_dict, audio_feat=br.full[:, br.head: br.head + br.S].float())
            br = None
        feats = br.full if br is not None else input_dict["audio_feat"]                       # B x S x C
        if not feats.is_cuda:
            raise RuntimeError("CIF runs on the HIP kernels: device tensors only (CPU restatement: oracle/cascaded_ref.py)")
        pad = input_dict["audio_feat_pad_mask"].bool().contiguous()      # B x S, True = padding
        B, S, C = (br.B, br.S, br.D) if br is not None else feats.shape
        if S > 2048 or C % 4 != 0:
            raise NotImplementedError(f"CIF kernels: at most 2048 frames per utterance and C % 4 == 0 (got S={S}, C={C})")
        if self.scaling_step >= 0 and self.apply_scaling and input_dict["global_step"] >= self.scaling_step:
            self.apply_scaling = False                         # cif.py:110-112: permanent once the step is reached
        # conv output of the last weight-generator layer; Dropout -> ReLU -> Dropout -> Linear(C, 1) -> Sigmoid in one row kernel
        # (sigmoid output; clip / masking: sc_cif_prepare)
        lin = self.weight_proj[1]
        p1 = float(self.conv[-2].p) if self.training else 0.0
        p2 = float(self.weight_proj[0].p) if self.training else 0.0
        seed1 = ops.next_mult_seed() if p1 > 0 else 0
        seed2 = ops.next_mult_seed() if p2 > 0 else 0
        last = self.conv[-3]
        grad_sink = None
        if br is not None:
            # the block's rows read in place: conv GEMM + weight head, forward and backward, in one autograd node
            grad_sink = {} if (torch.is_grad_enabled() and feats.requires_grad) else None
            alpha_raw = _WeightHeadRowsFn.apply(feats, last.weight, last.bias, lin.weight, lin.bias, br.head, S, last.padding[0], p1, seed1, p2,
                                                seed2, grad_sink)
        elif self.training and last.stride[0] == 1 and last.kernel_size[0] == 2 * last.padding[0] + 1:
            # training: the last conv layer as a strided-row GEMM over the zero-padded frames; the weight head runs over the GEMM's row
            # layout (pitch S + 2p per utterance, the extra rows are dropped from alpha)
            x = self._weight_conv(feats, upto=len(self.conv) - 3)
            pd = last.padding[0]
            y_full = _ConvRowsBf16Fn.apply(x, last.weight, last.bias, pd)
            a_full = _CifHeadFn.apply(y_full.unsqueeze(0), lin.weight, lin.bias, p1, seed1, p2, seed2)
            alpha_raw = a_full.view(-1)[: B * (S + 2 * pd)].view(B, S + 2 * pd)[:, :S]
        else:
            y = self._weight_conv(feats)
            alpha_raw = _CifHeadFn.apply(y.float().contiguous(), lin.weight, lin.bias, p1, seed1, p2, seed2)
        scaled = bool(self.apply_scaling and target_lengths is not None)
        tail = bool(self.apply_tail_handling and target_lengths is None)
        target = target_lengths.to(device=feats.device, dtype=torch.int64).contiguous() if target_lengths is not None else None
        T_known = None
        if scaled and target_lengths_host is not None:
            T_known = max(min(max(int(t), 1), MAX_FEAT_LEN) for t in target_lengths_host)
        st = {"thr": float(self.cif_threshold), "eps": float(eps), "apply_scaling": scaled, "scaled": scaled, "tail": tail,
              "tail_thr": float(self.tail_handling_firing_threshold), "flags": self.consistency_flags,
              "T_clip": T_known if T_known is not None else MAX_FEAT_LEN, "grad_sink": grad_sink}
        self.consistency_flags[2:3] += 1
        slots, quantity, feat_lengths = _CifFn.apply(feats, alpha_raw, pad, target, st, (br.head, S) if br is not None else None)
        if T_known is not None:
            T = T_known
        else:
            T = int(feat_lengths.max())                        # the one host read: the returned tensor's shape is data
        output = slots[:, :T].to(torch.float32 if br is not None else feats.dtype)
        fired = st["fired"].bool()
        if tail:
            # cif.py:281-283 marks, for EVERY row, the columns feat_len_j - 1 of the utterances j that fired their tail (diagnostic)
            ext = st["extend"].bool()
            cols = (feat_lengths - 1).clamp(min=0)
            add = torch.zeros(S, dtype=torch.bool, device=feats.device).index_put_((cols[ext],), torch.ones((), dtype=torch.bool,
                                                                                                       device=feats.device))
            fired = fired | add.unsqueeze(0)
        out_pad = ops.len_mask(feat_lengths, T)
        # frames per utterance: the lengths the mask was built from when it carries them (ops.len_mask), else counted
        src_lens = getattr(input_dict["audio_feat_pad_mask"], "_sc_lens", None)
        host = getattr(src_lens[0], "_sc_host", None) if src_lens is not None else None
        original_length = src_lens[0].long() if (host is not None and src_lens[1] == 0 and len(host) == B and max(host) <= S) \
            else (~pad).sum(-1).long()
        return {"quantity_out": quantity, "orig_alpha": st["a_clip"], "original_length": original_length,
                "target_len": target_lengths, "dsample_feats_pad_mask": out_pad, "dsample_feats": output,
                "dsample_feats_length": feat_lengths, "alpha": st["alpha"], "fired_marks": fired, "input_feats_pad_mask": pad}

    def _weight_conv(self, feats: torch.Tensor, upto: Optional[int] = None) -> torch.Tensor:
        """``self.conv(feats^T)^T`` (Conv1d k, stride 1, 'same' padding -> Dropout -> ReLU per layer) evaluated channels-last as ONE
        GEMM per layer over the k shifted copies of the input.  Training: bf16 operands / fp32 accumulation on the library's MFMA
        GEMM (the reference trains under precision-16 autocast, and the keyword COUNT is pinned by the target-length scaling, so
        the discrete part of CIF does not depend on this rounding).  Inference: exact fp32 on the matrix pipe
        (sc_sgemm_mfma_f32) - there the count is floor(sum alpha) and must be the fp32 reference's."""
        from .linear_fn import linear_bf16_autograd, linear_f32_autograd
        linear = linear_bf16_autograd if self.training else linear_f32_autograd
        x = feats
        for i in range(0, len(self.conv) if upto is None else upto, 3):
            conv, drop, act = self.conv[i], self.conv[i + 1], self.conv[i + 2]
            k, p = conv.kernel_size[0], conv.padding[0]
            B, T, C = x.shape
            xp = torch.nn.functional.pad(x, (0, 0, p, p))                           # (B, T + 2 p, C)
            cols = torch.cat([xp[:, j: j + T + 2 * p - k + 1] for j in range(k)], dim=-1)   # (B, T', k C), tap-major
            w = conv.weight.permute(0, 2, 1).reshape(conv.out_channels, k * C)      # [C_out, k, C_in] flattened tap-major
            x = linear(cols, w, conv.bias)
            if i + 3 < len(self.conv):                                              # not the last layer: its Dropout / ReLU follow here;
                x = act(drop(x))                                                    # the last layer's are part of the weight-head kernel
        return x
