"""Tensor-level wrappers over the C ABI (raw device pointers, current HIP stream).

torch is used here only for device memory and streams.  Every function requires device tensors and the
built extension; nothing falls back to eager PyTorch or the CPU.
"""
import ctypes
from typing import Optional

import torch

from ._lib import RtGemmArgs, RtLnArgs, RtLnBwdArgs, GemmArgs, HubertLayerArgs, Segments, check, diag_lib, lib


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("speechclip_plus_amd ops need device (HIP) tensors; there is no CPU path")
    return ctypes.c_void_p(t.data_ptr())


def aligned16(t: torch.Tensor) -> torch.Tensor:
    """``t`` itself when its storage offset is 16-byte aligned, else a fresh copy (trainable parameters are views into the
    optimiser's flat buffer at 4-byte granularity; kernels that load 16 bytes per lane need aligned bases)."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class RowSegments:
    """Ragged row layout of one batch (``sc_segments``, include/speechclip_hip.h): utterance b owns rows [row0[b], row0[b + 1]) -
    its own pitch, a multiple of 8 rows - of every activation buffer of the encoder.  Built on the host from the batch's lengths
    (``host_tables``: pure python, tested on the CPU) and uploaded in one pinned, asynchronous copy.

    ``pitch`` [B] rows per utterance; ``keys`` [B] the key count that sorts the attention work list (longest first)."""

    GRAN = 8             # SC_SEG_ROWS: pitches are multiples of this; one chunk-table entry per GRAN rows

    def __init__(self, pitch, keys, device, storage: Optional[torch.Tensor] = None, keys_known: bool = True):
        host, self.row0_host, self.n_work = self.host_tables(pitch, keys, keys_known)
        self.pitch = [int(p) for p in pitch]
        self.B, self.rows, self.max_pitch = len(self.pitch), self.row0_host[-1], max(self.pitch)
        n = host.numel()
        if storage is not None:
            assert storage.dtype == torch.int32 and storage.numel() >= n and storage.data_ptr() % 16 == 0
            dev = storage[:n]
            dev.copy_(host.pin_memory(), non_blocking=True)
        else:
            dev = host.pin_memory().to(device, non_blocking=True)
        nch = self.rows // self.GRAN
        self.chunk = dev[: 4 * nch]                              # 16-byte entries first: they stay 16-byte aligned
        self.work = dev[4 * nch: 4 * nch + 4 * self.n_work]
        self.row0 = dev[4 * nch + 4 * self.n_work:]
        self._dev = dev
        self.c = Segments()
        self.c.row0, self.c.chunk = _p(self.row0), _p(self.chunk)
        self.c.B, self.c.rows, self.c.max_pitch = self.B, self.rows, self.max_pitch

    @classmethod
    def table_ints(cls, B: int, max_rows: int) -> int:
        """upper bound of the table size in int32 (storage for a plan: B utterances, at most ``max_rows`` rows)"""
        return 4 * (max_rows // cls.GRAN) + B + 1 + 4 * B * ((max_rows + 127) // 128 + 1)

    @classmethod
    def host_tables(cls, pitch, keys, keys_known: bool = True):
        """-> (int32 tensor [chunk table | attention work list | row0], row0 as a python list, number of work items).
        chunk[c] = (first row, pitch, utterance, 0) of the utterance that owns rows 8 c .. 8 c + 7; work = one int4 item per
        128-query block, (utterance | q-block << 16, first row, pitch, key count or -1), longest utterance first (its workgroups run
        longest: start them first).  ``keys_known`` False: the key counts only order the list (the kernel reads valid_len)."""
        import numpy as np
        B, G = len(pitch), cls.GRAN
        p = np.asarray([int(x) for x in pitch], dtype=np.int64)
        assert B < 65536 and (p > 0).all() and (p % G == 0).all()
        row0 = np.concatenate([[0], np.cumsum(p)])
        per = np.stack([row0[:-1], p, np.arange(B), np.zeros(B, dtype=np.int64)], axis=1)         # one entry per utterance
        chunk = np.repeat(per, p // G, axis=0).reshape(-1)
        order = sorted(range(B), key=lambda b: (-int(keys[b]), b))
        work = [(b | (qb << 16), int(row0[b]), int(p[b]), int(keys[b]) if keys_known else -1)
                for b in order for qb in range((int(p[b]) + 127) // 128)]
        tab = np.concatenate([chunk, np.asarray(work, dtype=np.int64).reshape(-1), row0]).astype(np.int32)
        return torch.from_numpy(tab), [int(x) for x in row0], len(work)

    def ref(self):
        return ctypes.byref(self.c)


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py): events are recorded on the stream the kernel is
    launched on; ``work`` is the ALGORITHMIC flop (or byte) count the caller attributes to the launch."""

    def __init__(self):
        self.records = {}
        self.tagged = {}

    def add(self, name, start, end, work, tag=None):
        self.records.setdefault(name, []).append((start, end, work))
        if tag is not None:
            self.tagged.setdefault(tag, []).append((start, end, work))

    def summary_by_tag(self):
        return {tag: {"launches": len(r), "ms": sum(s.elapsed_time(e) for s, e, _ in r), "work": float(sum(w for _, _, w in r))}
                for tag, r in self.tagged.items()}

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[name] = {"launches": len(recs), "ms": ms, "work": float(sum(w for _, _, w in recs))}
        return out


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


def gemm_tile_name(M: int, N: int, K: int, n_split: int, batch: int, tile: int = 0) -> str:
    """Mirror of the tile selection in sc_gemm_bf16 (csrc/gemm_bf16.hip), for reporting only."""
    if tile == 0:
        if N <= 64 and n_split < 0:
            tile = 3
        elif M >= 512 and N >= 192 and ((M + 255) // 256) * ((N + 255) // 256) * batch >= 192 and (n_split < 0 or n_split % 64 == 0):
            tile = 2
        else:
            tile = 1
    return {1: "128x128", 2: "256x256", 3: "128x64", 7: "256x256", 8: "256x256", 9: "128x256_duo", 10: "128x256_duo", 11: "128x256_duo"}.get(tile, "diag")


def gemm_raw(A: torch.Tensor, lda: int, W: torch.Tensor, ldw: int, C: torch.Tensor, ldc: int, M: int, N: int, K: int,
             bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, ldr: int = 0, act: int = 0,
             out_f32: bool = False, Ct: Optional[torch.Tensor] = None, n_split: int = -1, R: int = 0, dh: int = 0,
             nb1: int = 1, nb2: int = 1, sA=(0, 0), sW=(0, 0), sC=(0, 0), sBias=(0, 0), sR=(0, 0),
             alg_rows: Optional[int] = None, tile: int = 0, drop_p: float = 0.0, drop_seed: int = 0, tap_c: int = 0,
             ln_stats: Optional[torch.Tensor] = None, ln_ns: int = 0, ln_colsum: Optional[torch.Tensor] = None,
             res_stats: Optional[torch.Tensor] = None, res_ns: int = 0, res_gamma: Optional[torch.Tensor] = None,
             res_beta: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None, ln_eps: float = 0.0,
             tn: bool = False, k_total: int = 0, aux: Optional[torch.Tensor] = None, aux_mode: int = 0,
             seg: Optional["RowSegments"] = None) -> int:
    """C = epi(A . W^T); see sc_gemm_args in include/speechclip_hip.h.  Pointers are the tensors' data_ptr()
    (pass a sliced view to offset).  ``alg_rows``: rows that are algorithmic work (excludes layout padding),
    used only by the optional KernelTimer.  ``ln_*`` / ``res_*`` / ``stats_out``: LayerNorm folded into the GEMM (row-statistics
    buffers are [M, 8, 2] fp32); returns the number of statistics strips written per row (0 without ``stats_out``)."""
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
    assert C.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    if bias is not None:
        assert bias.dtype == torch.float32
        bias = aligned16(bias)          # sc_gemm_bf16 reads the bias in 16-byte groups; trainable biases are 4-byte-aligned views
    if residual is not None:
        assert residual.dtype == torch.bfloat16
    a = GemmArgs()
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = _p(A), lda, _p(W), ldw, _p(C), ldc
    a.M, a.N, a.K = M, N, K
    a.bias, a.residual, a.ldr = _p(bias), _p(residual), ldr
    a.act, a.out_f32 = act, int(out_f32)
    a.Ct, a.n_split, a.R, a.dh = _p(Ct), n_split, R, dh
    a.nb1, a.nb2 = nb1, nb2
    a.sA1, a.sA2 = sA
    a.sW1, a.sW2 = sW
    a.sC1, a.sC2 = sC
    a.sBias1, a.sBias2 = sBias
    a.sR1, a.sR2 = sR
    a.tile = tile
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed) & 0xffffffff
    a.tap_c = int(tap_c)
    a.k_total = int(k_total)
    if aux_mode:            # activation fused with a second [M, N] bf16 operand of C's row stride (sc_gemm_args.aux_mode)
        assert aux is not None and aux.dtype == torch.bfloat16 and Ct is None and aux.stride(0) == ldc and aux.stride(1) == 1
        a.Ct, a.aux_mode = _p(aux), int(aux_mode)
    a.tn = int(tn)          # C[m, n] = sum_r A[r, m] W[r, n]: both operands row-indexed by the reduction (weight gradients)
    if seg is not None:     # ragged rows: the transposed (V^T) store looks its utterance up per 32-row chunk
        assert M == seg.rows
        a.seg_chunk = _p(seg.chunk)
    strips = 0
    if ln_colsum is not None and ln_stats is None and stats_out is None:
        # LayerNorm in the GEMM's prologue (128- / 64-row tiles, round 6): A holds raw rows, W = W0 diag(gamma), bias = c_n, the row
        # statistics are computed by the kernel (csrc/gemm_bf16.hip: ln_self)
        assert ln_colsum.dtype == torch.float32 and ln_colsum.is_contiguous() and ln_colsum.numel() == N and ln_eps > 0.0 and K <= 1024
        a.ln_colsum, a.ln_eps = _p(ln_colsum), float(ln_eps)
    if ln_stats is not None or stats_out is not None:
        for t in (ln_stats, ln_colsum, res_stats, res_gamma, res_beta, stats_out):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
        a.ln_stats, a.ln_ns, a.ln_colsum = _p(ln_stats), int(ln_ns), _p(ln_colsum)
        a.res_stats, a.res_ns, a.res_gamma, a.res_beta = _p(res_stats), int(res_ns), _p(res_gamma), _p(res_beta)
        a.stats_out, a.ln_eps = _p(stats_out), float(ln_eps)
        if stats_out is not None:
            assert stats_out.numel() >= M * 16
            strips = int(diag_lib().sc_gemm_stats_strips(ctypes.byref(a)))
    # diagnostic tile ids and the opt-in LayerNorm-folded GEMMs exist in the diagnostics library only (the product library refuses them)
    L = diag_lib() if (tile in (32, 34) or ln_stats is not None or stats_out is not None) else lib()
    if _timer is None:
        check(L.sc_gemm_bf16(ctypes.byref(a), _stream()), "sc_gemm_bf16", L)
        return strips
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    check(L.sc_gemm_bf16(ctypes.byref(a), _stream()), "sc_gemm_bf16", L)
    ev1.record()
    rows = (M if alg_rows is None else alg_rows) * nb1 * nb2
    _timer.add("gemm_bf16_" + gemm_tile_name(M, N, K, n_split, nb1 * nb2, 2 if (tile == 0 and strips + ln_ns > 0) else tile), ev0, ev1,
               2.0 * rows * N * K, tag=f"M{M} N{N} K{K} lda{lda} act{act} res{int(residual is not None)} z{nb1 * nb2}")
    return strips


def hubert_layer_fwd(x: torch.Tensor, out: torch.Tensor, valid_len: torch.Tensor, w: dict, i: int, pl, B: int, R: int, T: int, D: int,
                     F_: int, H: int, pre_ln: bool, p_attn: float = 0.0, p_res: float = 0.0, seeds=(0, 0, 0), fused=None,
                     seg: Optional["RowSegments"] = None) -> None:
    """One frozen HuBERT encoder layer in ONE C-ABI call (sc_hubert_layer_fwd: QKV -> attention -> out_proj -> LN -> FC1 -> FC2 ->
    LN on the caller's stream).  ``w``: the encoder's weight dict (keys l{i}_*), ``pl``: its plan (scratch buffers).
    ``fused`` = (x_stats or None, x_ns, out_stats): the LayerNorm-free form (the LayerNorms folded into the GEMMs; ``out`` receives
    raw rows + statistics); x_stats None = ``x`` is an ordinary, materialised input (layer 0)."""
    a = HubertLayerArgs()
    a.x, a.out, a.valid_len = _p(x), _p(out), _p(valid_len)
    a.B, a.R, a.T, a.D, a.F, a.H, a.pre_ln = B, R, T, D, F_, H, int(pre_ln)
    a.qkv_w, a.o_w, a.fc1_w, a.fc2_w = _p(w[f"l{i}_qkv_w"]), _p(w[f"l{i}_o_w"]), _p(w[f"l{i}_fc1_w"]), _p(w[f"l{i}_fc2_w"])
    a.qkv_b, a.o_b, a.fc1_b, a.fc2_b = _p(w[f"l{i}_qkv_b"]), _p(w[f"l{i}_o_b"]), _p(w[f"l{i}_fc1_b"]), _p(w[f"l{i}_fc2_b"])
    a.ln1_g, a.ln1_b, a.ln2_g, a.ln2_b = _p(w[f"l{i}_ln1_g"]), _p(w[f"l{i}_ln1_b"]), _p(w[f"l{i}_ln2_g"]), _p(w[f"l{i}_ln2_b"])
    a.eps, a.p_attn, a.p_res = 1e-5, float(p_attn), float(p_res)
    a.seed_attn, a.seed_o, a.seed_fc2 = (int(s) & 0xffffffff for s in seeds)
    a.qk, a.vt, a.ctx, a.pre, a.x1, a.ffn = _p(pl.qk), _p(pl.vt), _p(pl.ctx), _p(pl.pre), _p(pl.x1), _p(pl.ffn)
    if seg is not None:
        a.seg = ctypes.pointer(seg.c)
        a.attn_work, a.n_attn_work = _p(seg.work), seg.n_work
    if fused is not None:
        x_stats, x_ns, out_stats = fused
        a.fused_ln = 1
        a.fc1_w, a.fc1_b, a.fc1_colsum = _p(w[f"l{i}_fc1_wf"]), _p(w[f"l{i}_fc1_cf"]), _p(w[f"l{i}_fc1_sf"])
        a.stats1, a.out_stats = _p(pl.stats1), _p(out_stats)
        if x_stats is not None:
            a.x_stats, a.x_ns = _p(x_stats), int(x_ns)
            a.x_ln_g, a.x_ln_b = _p(w[f"l{i - 1}_ln2_g"]), _p(w[f"l{i - 1}_ln2_b"])
            a.qkv_w, a.qkv_b, a.qkv_colsum = _p(w[f"l{i}_qkv_wf"]), _p(w[f"l{i}_qkv_cf"]), _p(w[f"l{i}_qkv_sf"])
    L = diag_lib() if fused is not None else lib()       # the LayerNorm-free form runs on the diagnostics library's GEMMs (opt-in)
    check(L.sc_hubert_layer_fwd(ctypes.byref(a), _stream()), "sc_hubert_layer_fwd", L)


def gemm_stats_strips(M: int, N: int) -> int:
    """Row-statistics strips a producer GEMM [M, N] writes (one per N-tile of the width the dispatcher picks)."""
    a = GemmArgs()
    a.M, a.N, a.K, a.n_split, a.nb1, a.nb2 = int(M), int(N), 64, -1, 1, 1
    return int(diag_lib().sc_gemm_stats_strips(ctypes.byref(a)))


def linear_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, act: int = 0, out_f32: bool = False,
                alg_rows: Optional[int] = None, tile: int = 0, drop_p: float = 0.0, drop_seed: int = 0,
                aux: Optional[torch.Tensor] = None, aux_mode: int = 0, ln_colsum: Optional[torch.Tensor] = None, ln_eps: float = 0.0) -> torch.Tensor:
    """y[M, N] = epi(x[M, K] . w[N, K]^T) for contiguous 2-D operands.  ``aux`` / ``aux_mode`` (small problems, 128-row tiles):
    1 = also store the pre-activation into ``aux`` and return act(it); 2 = return (x . w^T) * act'(aux).
    ``ln_colsum`` / ``ln_eps``: y = epi(LayerNorm(x) . w0^T) with the LayerNorm in the GEMM's prologue - ``w`` = w0 diag(gamma),
    ``ln_colsum[n]`` = sum_k w[n, k], ``bias[n]`` = sum_k beta[k] w0[n, k] + bias0[n] (``fold_layernorm``); K <= 1024."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    gemm_raw(x, x.stride(0), w, w.stride(0), out, out.stride(0), M, N, K, bias=bias, residual=residual,
             ldr=residual.stride(0) if residual is not None else 0, act=act, out_f32=out_f32, alg_rows=alg_rows, tile=tile,
             drop_p=drop_p, drop_seed=drop_seed, aux=aux, aux_mode=aux_mode, ln_colsum=ln_colsum, ln_eps=ln_eps)
    return out


def fold_layernorm(w0: torch.Tensor, bias0: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(w, ln_colsum, bias) for linear_bf16(..., ln_colsum=...): LayerNorm(x) w0^T + bias0 = rstd (x w^T - mean ln_colsum) + bias with
    w = bf16(w0 diag(gamma)), ln_colsum = the row sums of the bf16 values the kernel multiplies, bias = bias0 + w0 beta (fp32)."""
    w0f = w0.detach().float()
    w = (w0f * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    colsum = w.float().sum(dim=1).contiguous()
    c = w0f @ beta.detach().float()
    if bias0 is not None:
        c = c + bias0.detach().float()
    return w, colsum, c.contiguous()


def attn_fwd(qk: torch.Tensor, vt: torch.Tensor, valid_len: torch.Tensor, out: torch.Tensor, B: int, R: int, H: int,
             D: int, scale: float, alg_flops: float = 0.0, lse2: Optional[torch.Tensor] = None, causal: bool = False,
             drop_p: float = 0.0, drop_seed: int = 0, seg: Optional["RowSegments"] = None, use_work: bool = True) -> None:
    """``seg``: ragged rows (B / R are then ignored; vt = per utterance [H, 64, pitch] back to back, as gemm_raw(seg=...) writes it)."""
    assert qk.dtype == torch.bfloat16 and vt.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    assert valid_len.dtype == torch.int32
    if _timer is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if seg is not None:
        check(lib().sc_attn_fwd_seg_bf16(_p(qk), qk.stride(0), _p(vt), _p(valid_len), _p(out), out.stride(0), seg.ref(),
                                         _p(seg.work) if use_work else _p(None), seg.n_work if use_work else 0, H, D, float(scale), _p(lse2),
                                         int(causal), float(drop_p), int(drop_seed) & 0xffffffff, _stream()), "sc_attn_fwd_seg_bf16")
    else:
        check(lib().sc_attn_fwd_bf16(_p(qk), qk.stride(0), _p(vt), _p(valid_len), _p(out), out.stride(0), B, R, H, D,
                                     float(scale), _p(lse2), int(causal), float(drop_p), int(drop_seed) & 0xffffffff, _stream()),
              "sc_attn_fwd_bf16")
    if _timer is not None:
        ev1.record()
        _timer.add("attn_fwd", ev0, ev1, float(alg_flops))


def head_transpose(x: torch.Tensor, B: int, R: int, H: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """xT[b, h, d, t] = x[b R + t, h 64 + d]; ``x`` may be a column slice of a wider row-major buffer."""
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    if out is None:
        out = torch.empty(B, H, 64, R, device=x.device, dtype=torch.bfloat16)
    check(lib().sc_head_transpose_bf16(_p(x), x.stride(0), _p(out), B, R, H, _stream()), "sc_head_transpose_bf16")
    return out


def attn32_fwd(qkv: torch.Tensor, heads: int, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal attention of 32-row sequences (sc_attn32_fwd_bf16): qkv [nseq 32, 3 heads 64] bf16 -> [nseq 32, heads 64]"""
    M, W3 = qkv.shape
    W = heads * 64
    assert qkv.dtype == torch.bfloat16 and qkv.stride(1) == 1 and W3 == 3 * W and M % 32 == 0
    if out is None:
        out = torch.empty(M, W, device=qkv.device, dtype=torch.bfloat16)
    assert out.dtype == torch.bfloat16 and out.stride(1) == 1 and tuple(out.shape) == (M, W)
    check(lib().sc_attn32_fwd_bf16(_p(qkv), qkv.stride(0), _p(out), out.stride(0), M // 32, heads, float(scale), _stream()), "sc_attn32_fwd_bf16")
    return out


def attn32_bwd(qkv: torch.Tensor, dout: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """-> d qkv [nseq 32, 3 heads 64] bf16 (sc_attn32_bwd_bf16: from qkv and d out alone)"""
    M, W3 = qkv.shape
    W = heads * 64
    assert qkv.dtype == torch.bfloat16 and qkv.stride(1) == 1 and W3 == 3 * W and M % 32 == 0
    assert dout.dtype == torch.bfloat16 and dout.stride(1) == 1 and tuple(dout.shape) == (M, W)
    dqkv = torch.empty(M, 3 * W, device=qkv.device, dtype=torch.bfloat16)
    check(lib().sc_attn32_bwd_bf16(_p(qkv), qkv.stride(0), _p(dout), dout.stride(0), _p(dqkv), 3 * W, M // 32, heads, float(scale), _stream()),
          "sc_attn32_bwd_bf16")
    return dqkv


def attn_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse2: torch.Tensor,
             valid_len: torch.Tensor, dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, B: int, R: int, H: int, scale: float,
             causal: bool = False, kT: Optional[torch.Tensor] = None, q_rows: Optional[int] = None, drop_p: float = 0.0,
             drop_seed: int = 0) -> None:
    """q, k, v, out, dout, dq, dk, dv: [B R, H 64] views (column slices of wider buffers allowed); query rows t >= q_rows
    (default R) must carry dout = 0.  ``drop_p`` / ``drop_seed``: those of the forward (attn_fwd) when it dropped probabilities."""
    for t in (q, k, v, out, dout, dq, dk, dv):
        assert t.dtype == torch.bfloat16 and t.stride(1) == 1
    delta = torch.empty(B, H, R, device=q.device, dtype=torch.float32)
    args = lambda qT, kT, doT: (_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0), _p(dout), dout.stride(0),
                                _p(qT), _p(kT), _p(doT), _p(lse2), _p(delta), _p(valid_len), _p(dq), dq.stride(0), _p(dk), dk.stride(0),
                                _p(dv), dv.stride(0), B, R, H, R if q_rows is None else q_rows, float(scale), int(causal), float(drop_p),
                                int(drop_seed) & 0xffffffff, _stream())
    if kT is None:          # the three per-head transposes and delta in one preparation launch inside the call
        T3 = torch.empty(3, B, H, 64, R, device=q.device, dtype=torch.bfloat16)
        check(lib().sc_attn_bwd_fused_bf16(*args(T3[0], T3[1], T3[2])), "sc_attn_bwd_fused_bf16")
        return
    qT = head_transpose(q, B, R, H)
    doT = head_transpose(dout, B, R, H)
    check(lib().sc_attn_bwd_bf16(*args(qT, kT, doT)), "sc_attn_bwd_bf16")


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float, dres: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, want_param_grads: bool = False, acc=None, drop=None, sum_acc: Optional[torch.Tensor] = None):
    """dx = LN'(x)(dy) (+ dres) for bf16 rows; with ``want_param_grads`` also returns (dgamma, dbeta) fp32.
    ``acc`` = (dgamma_target, dbeta_target): the parameter gradients are ADDED to these fp32 [D] tensors by the reduction itself (no
    temporaries, no add launches); returns dx only.
    ``drop`` = (p, seed): also returns F.dropout(dx) with the stateless mask of element row * D + col (the residual branch's operand) -
    result (dx, dx_dropped); ``sum_acc`` fp32 [D]: the column sums of that dropped copy (of dx without ``drop``) are ADDED to it (the
    branch's bias gradient; needs ``acc`` / ``want_param_grads``).  One pass (sc_layernorm_bwd_drop_bf16)."""
    want_param_grads = want_param_grads or acc is not None
    rows, D = x.shape
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and gamma.dtype == torch.float32
    if out is None:
        out = torch.empty(rows, D, device=x.device, dtype=torch.bfloat16)
    dg = db = ds = None
    n_part = 0
    if want_param_grads:
        n_part = min(1024, (rows + 3) // 4)             # one partial row per workgroup (its four waves are added in LDS)
        dg = torch.empty(n_part, D, device=x.device, dtype=torch.float32)
        db = torch.empty(n_part, D, device=x.device, dtype=torch.float32)
    ext = drop is not None or sum_acc is not None
    out_drop = None
    if ext:
        p, seed = drop if drop is not None else (0.0, 0)
        if p > 0.0:
            out_drop = torch.empty(rows, D, device=x.device, dtype=torch.bfloat16)
        if sum_acc is not None:
            assert want_param_grads, "the column sums ride on the parameter-gradient partials"
            ds = torch.empty(n_part, D, device=x.device, dtype=torch.float32)
        check(lib().sc_layernorm_bwd_drop_bf16(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(gamma), _p(dres), 0 if dres is None else dres.stride(0),
                                               _p(out), out.stride(0), rows, D, float(eps), _p(dg), _p(db), n_part, _p(out_drop),
                                               D if out_drop is not None else 0, float(p), int(seed) & 0xffffffff, _p(ds), _stream()),
              "sc_layernorm_bwd_drop_bf16")
        if sum_acc is not None:
            colsum(ds, D, n_part, D, sum_acc, beta=1.0)
        if out_drop is None:
            out_drop = out
    else:
        check(lib().sc_layernorm_bwd_bf16(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(gamma), _p(dres), 0 if dres is None else dres.stride(0),
                                          _p(out), out.stride(0), rows, D, float(eps), _p(dg), _p(db), n_part, _stream()),
              "sc_layernorm_bwd_bf16")
    res = (out, out_drop) if drop is not None else out
    if not want_param_grads:
        return res
    if acc is not None:
        colsum(dg, D, n_part, D, acc[0], beta=1.0)
        colsum(db, D, n_part, D, acc[1], beta=1.0)
        return res
    g, b = torch.empty(D, device=x.device, dtype=torch.float32), torch.empty(D, device=x.device, dtype=torch.float32)
    colsum(dg, D, n_part, D, g)
    colsum(db, D, n_part, D, b)
    return (res, g, b) if drop is not None else (out, g, b)


def transpose_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None, colsum_partial: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[rows, cols] (row stride >= cols) -> contiguous [cols, rows]; rows and cols multiples of 8.  ``colsum_partial``
    ([ceil(rows / 64), cols] fp32) receives the per-64-row-block column sums of x (first stage of a bias gradient)."""
    rows, cols = x.shape
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    if out is None:
        out = torch.empty(cols, rows, device=x.device, dtype=torch.bfloat16)
    check(lib().sc_transpose_bf16(_p(x), x.stride(0), _p(out), out.stride(0), rows, cols, _p(colsum_partial), _stream()),
          "sc_transpose_bf16")
    return out


_cus = []


def _num_cus() -> int:
    if not _cus:
        _cus.append(int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count))
    return _cus[0]


_derived = {}


def derived(param: torch.Tensor, tag: str, fn):
    """``fn(param.detach())`` cached per parameter VERSION (bf16 working copies, transposed copies of trainable weights: computed once
    per optimiser step instead of once per call).  The key carries optim.param_generation() because the fused Adam kernel writes the
    masters through a raw pointer (no ``_version`` bump).  Entries hold only a weak reference to their parameter and die with it."""
    import weakref
    from .optim import param_generation
    if not isinstance(param, torch.nn.Parameter):       # a temporary: its id / address can be recycled for different contents
        return fn(param.detach())
    key = (id(param), tag)
    ver = (param.data_ptr(), param._version, param_generation(), tuple(param.shape))
    hit = _derived.get(key)
    if hit is not None and hit[0]() is param and hit[1] == ver:
        return hit[2]
    val = fn(param.detach())
    if len(_derived) > 512:                             # drop the entries of parameters that no longer exist
        for k in [k for k, v in _derived.items() if v[0]() is None]:
            del _derived[k]
    _derived[key] = (weakref.ref(param), ver, val)
    return val


def bf16_copy(t: torch.Tensor) -> torch.Tensor:
    """contiguous bf16 copy of ``t`` (any float dtype, any strides - a permuted / transposed / flipped view) in ONE copy kernel;
    ``t.to(bfloat16).contiguous()`` on a strided view is a cast launch plus a layout launch"""
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    out.copy_(t.detach())
    return out


def weight_copies(w: torch.Tensor, want: str = "both"):
    """bf16 working copies of an fp32 [N, K] weight in ONE launch (sc_cast_transpose_f32_bf16): -> (bf16(w) [N, K], bf16(w)^T [K, N]);
    ``want`` "plain" / "T" / "both" (the entry not asked for is None).  Falls back to torch for shapes / dtypes the kernel does not take."""
    w = w.detach()
    N, K = w.shape
    if not (w.is_cuda and w.dtype == torch.float32 and w.stride(1) == 1 and w.stride(0) % 4 == 0 and N % 4 == 0 and K % 4 == 0 and w.data_ptr() % 16 == 0):
        wb = w.to(torch.bfloat16).contiguous()
        return (wb if want != "T" else None), (wb.t().contiguous() if want != "plain" else None)
    y = torch.empty(N, K, device=w.device, dtype=torch.bfloat16) if want != "T" else None
    yT = torch.empty(K, N, device=w.device, dtype=torch.bfloat16) if want != "plain" else None
    check(lib().sc_cast_transpose_f32_bf16(_p(w), w.stride(0), _p(y), K, _p(yT), N, N, K, _stream()), "sc_cast_transpose_f32_bf16")
    return y, yT


def derived_pair(param: torch.Tensor):
    """(bf16 copy, transposed bf16 copy) of a trainable 2-D weight, both cached per parameter version (``derived``) and produced by
    one launch when neither is cached."""
    pair = derived(param, "bf16_pair", lambda t: weight_copies(t, "both"))
    return pair


def len_mask(lens: torch.Tensor, n: int, add: int = 0) -> torch.Tensor:
    """bool [B, n]: True where k >= lens[b] + add (sc_len_mask_u8); the lengths ride along as ``._sc_lens`` = (lens, add) so that a
    consumer can rebuild the mask at another pitch, or read the lengths back, without a reduction over the mask"""
    assert lens.dtype == torch.int64 and lens.dim() == 1 and lens.is_contiguous()
    B = lens.shape[0]
    m = torch.empty(B, n, device=lens.device, dtype=torch.uint8)
    check(lib().sc_len_mask_u8(_p(lens), int(add), _p(m), B, int(n), _stream()), "sc_len_mask_u8")
    m = m.view(torch.bool)
    m._sc_lens = (lens, int(add))
    return m


_shared_streams = {}


def shared_stream(name: str, device=None, priority: int = 0) -> "torch.cuda.Stream":
    """ONE HIP stream per (role, device) and process.  The runtime multiplexes streams onto a handful of hardware queues; two streams
    that land on the same queue run one after the other.  Roles that are meant to run side by side ("encoder", "optimiser",
    "allreduce", "head_aux") therefore keep the streams they got first, instead of every model / trainer instance drawing new ones (the
    fourth model built in one process lost the encoder / tail overlap that way)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (name, idx)
    st = _shared_streams.get(key)
    if st is None:
        from . import warn_if_hw_queues_short
        warn_if_hw_queues_short()               # the first side stream of the process: say so once if the hardware queues cannot carry it
        # priority < 0: a high-priority stream gets a hardware queue of its own priority class - not one of the few normal-priority
        # queues the other roles share (the "h2d" copy stream: queued on the encoder's hardware queue, a batch's copy would start
        # only when the encoder in front of it has finished)
        st = _shared_streams[key] = torch.cuda.Stream(device=idx, priority=priority)
    return st


def grad_target(p):
    """``p.grad`` if a backward may ADD its result into it directly (the optimiser's flat gradient buffer: fp32, dense, p's shape) -
    the producing kernel's own reduction then accumulates (beta = 1) and the autograd node returns None for that input, which saves
    the temporary and the AccumulateGrad add launch per parameter.  None (p.grad unset / foreign): return the gradient as usual."""
    g = getattr(p, "grad", None)
    if g is None or not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
        return None
    # ONLY the optimiser's own flat gradient buffer (optim.FlatAdam tags the views it hands out): adding into any other .grad would
    # bypass tensor / post-accumulate-grad hooks and write .grad under torch.autograd.grad, which must not touch it
    if not getattr(g, "_sc_flat", False):
        return None
    return g if (g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g.device == p.device) else None


def invalidate_derived() -> None:
    """Forget every cached derived copy (after writing parameters through ``.data``, which bumps no version counter)."""
    _derived.clear()


def colsum_bf16(x: torch.Tensor, out: torch.Tensor, beta: float = 0.0) -> None:
    """out[c] = beta out[c] + sum_r x[r, c]  (bf16 rows -> fp32; bias gradients): row-block partials, then sc_colsum_f32 in block order."""
    rows, cols = x.shape
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1 and out.dtype == torch.float32
    nblk = max(1, min(512, rows // 64))
    part = torch.empty(nblk, cols, device=x.device, dtype=torch.float32)
    check(lib().sc_colsum_bf16(_p(x), x.stride(0), rows, cols, _p(part), nblk, _stream()), "sc_colsum_bf16")
    colsum(part, cols, nblk, cols, out, beta=beta)


def wgrad_bf16(dy: torch.Tensor, x: torch.Tensor, gW, gb: Optional[torch.Tensor] = None, beta: float = 1.0) -> None:
    """gW[N, K] = beta gW + dy[rows, N]^T x[rows, K]  (fp32, contiguous) ;  gb[N] = beta gb + colsum(dy).
    ``gW`` may be a LIST of fp32 contiguous tensors that together hold the N rows in order (q / k / v projection weights of one fused
    QKV product): the slice reduction writes each block straight into its tensor.

    The contraction runs over the rows.  N, K multiples of 256: the TN form of sc_gemm_bf16 reads both operands in place (row-major,
    any row stride - also the overlapping-row im2col view of a conv input), split along the rows into fp32 partials that
    sc_colsum_f32 adds in slice order.  Other shapes (grouped pos_conv: 48 columns): both operands are transposed first."""
    rows, N = dy.shape
    K = x.shape[1]
    blocks = None
    if isinstance(gW, (list, tuple)):
        blocks = list(gW)
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == K for t in blocks) and sum(t.shape[0] for t in blocks) == N
        gW = None
    else:
        assert gW.dtype == torch.float32 and gW.is_contiguous() and tuple(gW.shape) == (N, K)
    assert x.shape[0] == rows and rows % 64 == 0
    assert dy.stride(1) == 1 and x.stride(1) == 1
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    tn_ok = N % 256 == 0 and K % 256 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
    if blocks is not None and not tn_ok:                     # generic shapes: one temporary, then block adds
        tmp = torch.empty(N, K, device=dy.device, dtype=torch.float32)
        wgrad_bf16(dy, x, tmp, None, beta=0.0)
        r = 0
        for t in blocks:
            if beta == 0.0:
                t.copy_(tmp[r: r + t.shape[0]])
            else:
                t.mul_(beta).add_(tmp[r: r + t.shape[0]])
            r += t.shape[0]
        if gb is not None:
            colsum_bf16(dy, gb, beta=beta)
        return
    if tn_ok:
        kt = rows // 64
        S = max(1, min(_num_cus() // tiles, kt // 4))     # one round of workgroups; slices of whole K-tiles, the last one shorter
        Kc = -(-kt // S) * 64
        S = -(-rows // Kc)
        part = gW if (S == 1 and beta == 0.0 and gW is not None) else torch.empty(S, N, K, device=dy.device, dtype=torch.float32)
        gemm_raw(dy, dy.stride(0), x, x.stride(0), part, K, N, K, Kc, out_f32=True, nb1=S, sA=(Kc * dy.stride(0), 0),
                 sW=(Kc * x.stride(0), 0), sC=(N * K, 0), tn=True, k_total=rows)
        if blocks is not None:
            r = 0
            for t in blocks:                                   # rows r .. r + n of every slice are one contiguous run of n K floats
                n = t.shape[0]
                colsum(part.view(S, N * K)[:, r * K:], N * K, S, n * K, t, beta=beta)
                r += n
        elif part is not gW:
            colsum(part, N * K, S, N * K, gW, beta=beta)
        if gb is not None:
            colsum_bf16(dy, gb, beta=beta)
        return
    pb = torch.empty((rows + 63) // 64, N, device=dy.device, dtype=torch.float32) if gb is not None else None
    S = max(1, min(256 // max(tiles, 1), rows // 512))
    # the slices must be equal and multiples of 64 rows: the transposed operands are laid out with the row count padded up to
    # 64 * S and the pad columns zeroed (rows = B * 499 has no useful divisor; one un-split GEMM over 32k rows would leave most
    # of the chip idle)
    rows_pad = -(-rows // (64 * S)) * (64 * S)
    dyT = torch.empty(N, rows_pad, device=dy.device, dtype=torch.bfloat16)
    xT = torch.empty(K, rows_pad, device=dy.device, dtype=torch.bfloat16)
    if rows_pad != rows:
        dyT[:, rows:].zero_()
        xT[:, rows:].zero_()
    transpose_bf16(dy, out=dyT, colsum_partial=pb)
    transpose_bf16(x, out=xT)
    Kc = rows_pad // S
    part = torch.empty(S, N, K, device=dy.device, dtype=torch.float32)
    gemm_raw(dyT, rows_pad, xT, rows_pad, part, K, N, K, Kc, out_f32=True, nb1=S, sA=(Kc, 0), sW=(Kc, 0), sC=(N * K, 0))
    colsum(part, N * K, S, N * K, gW, beta=beta)
    if gb is not None:                      # second stage of the bias gradient (first stage: the transpose of dy above)
        colsum(pb, N, pb.shape[0], N, gb, beta=beta)


def posconv_wgrad(du: torch.Tensor, xg: torch.Tensor, G: int, rows: int, Dg: int, Kp: int) -> torch.Tensor:
    """gw [G, Dg, Kp Dg] fp32 = the grouped pos_conv weight gradient (sc_posconv_wgrad_bf16): du / xg are [G, rows, Dg] slabs given by
    their first element (du advanced by the halo, see the header); row slices reduced in order by sc_colsum_f32."""
    assert du.dtype == torch.bfloat16 and xg.dtype == torch.bfloat16 and rows % 128 == 0
    Z = 4
    while (rows // 128) % Z:
        Z -= 1
    n = G * Dg * Kp * Dg
    part = torch.empty(Z, n, device=du.device, dtype=torch.float32)
    check(lib().sc_posconv_wgrad_bf16(_p(du), _p(xg), _p(part), G, rows, Dg, Kp, Z, _stream()), "sc_posconv_wgrad_bf16")
    if Z == 1:
        return part.view(G, Dg, Kp * Dg)
    gw = torch.empty(G, Dg, Kp * Dg, device=du.device, dtype=torch.float32)
    colsum(part, n, Z, n, gw)
    return gw


def dropout_bf16(x: torch.Tensor, p: float, seed: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.dropout(x, p) on bf16 rows with the stateless hash mask (element row * D + col); in place when out is x."""
    rows, D = x.shape
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    if out is None:
        out = torch.empty_like(x)
    check(lib().sc_dropout_bf16(_p(x), x.stride(0), _p(out), out.stride(0), rows, D, float(p), int(seed) & 0xffffffff, _stream()),
          "sc_dropout_bf16")
    return out


_mult_calls = [0]


def next_mult_seed() -> int:
    """Seed of the next fp32 dropout site: a function of torch's seed and a call counter (reproducible under torch.manual_seed, no
    device RNG state).  A site is either materialised (``dropout_mult``) or evaluated inside its consumer kernel (rt_* ops below) -
    the keep bit of element i is the same either way."""
    _mult_calls[0] += 1
    return ((torch.initial_seed() * 0x9E3779B1) ^ (_mult_calls[0] * 0x85EBCA6B)) & 0xffffffff


def dropout_mult(shape, p: float, device, seed: Optional[int] = None) -> torch.Tensor:
    """fp32 multiplier of F.dropout(., p) for a tensor of ``shape`` (numel a multiple of 8): keep ? 1 / (1 - p) : 0, one launch."""
    n = 1
    for d in shape:
        n *= int(d)
    out = torch.empty(n, device=device, dtype=torch.float32)
    if seed is None:
        seed = next_mult_seed()
    check(lib().sc_dropout_mult_f32(_p(out), n, float(p), seed, _stream()), "sc_dropout_mult_f32")
    return out.view(*shape)


def cif_fwd(x: torch.Tensor, alpha: torch.Tensor, csum: torch.Tensor, T: int, thr: float) -> torch.Tensor:
    """integrate-and-fire accumulation: x [B,S,C] fp32, alpha / csum [B,S] fp32 -> out [B,T+1,C] (see sc_cif_fwd)."""
    B, S, C = x.shape
    for t in (x, alpha, csum):
        assert t.dtype == torch.float32 and t.is_contiguous()
    out = torch.empty(B, T + 1, C, device=x.device, dtype=torch.float32)
    check(lib().sc_cif_fwd(_p(x), _p(alpha), _p(csum), _p(out), B, S, C, T, float(thr), _stream()), "sc_cif_fwd")
    return out


def cif_bwd(x: torch.Tensor, alpha: torch.Tensor, csum: torch.Tensor, g: torch.Tensor, T: int, thr: float):
    """-> (dx [B,S,C], pa, pb [nblk,B,S]): per-channel-block partials of d alpha (direct part) and d csum (sc_cif_prepare_bwd adds
    them up)."""
    B, S, C = x.shape
    assert g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (B, T + 1, C)
    nblk = (C + 255) // 256
    dx = torch.empty_like(x)
    pa = torch.empty(nblk, B, S, device=x.device, dtype=torch.float32)
    pb = torch.empty(nblk, B, S, device=x.device, dtype=torch.float32)
    check(lib().sc_cif_bwd(_p(x), _p(alpha), _p(csum), _p(g), _p(dx), _p(pa), _p(pb), B, S, C, T, float(thr), _stream()), "sc_cif_bwd")
    return dx, pa, pb


def cif_prepare(alpha_raw: torch.Tensor, pad: torch.Tensor, target: Optional[torch.Tensor], apply_scaling: bool, thr: float, eps: float,
                max_feat: int, T: int, flags: torch.Tensor) -> dict:
    """CIF bookkeeping of one batch on the device (sc_cif_prepare): alpha_raw [B,S] fp32 (row stride free), pad [B,S] uint8 / bool."""
    B, S = alpha_raw.shape
    assert alpha_raw.dtype == torch.float32 and alpha_raw.stride(1) == 1 and pad.element_size() == 1 and pad.stride(1) == 1
    assert flags.dtype == torch.int32 and flags.numel() >= 8
    if target is not None:
        assert target.dtype == torch.int64 and target.is_contiguous() and target.numel() == B
    dev = alpha_raw.device
    f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
    r = dict(a_clip=f(B, S), alpha=f(B, S), csum=f(B, S), quantity=f(B), ratio=f(B),
             feat_len=torch.empty(B, device=dev, dtype=torch.int64), fired=torch.empty(B, S, device=dev, dtype=torch.uint8))
    check(lib().sc_cif_prepare(_p(alpha_raw), alpha_raw.stride(0), _p(pad), pad.stride(0), _p(target), int(bool(apply_scaling)), B, S,
                               float(thr), float(eps), int(max_feat), int(T), _p(r["a_clip"]), _p(r["alpha"]), _p(r["csum"]),
                               _p(r["quantity"]), _p(r["ratio"]), _p(r["feat_len"]), _p(r["fired"]), _p(flags), _stream()),
          "sc_cif_prepare")
    return r


def cif_prepare_bwd(pa: torch.Tensor, pb: torch.Tensor, a_clip: torch.Tensor, pad: torch.Tensor, ratio: torch.Tensor,
                    quantity: torch.Tensor, gq: Optional[torch.Tensor], scaled: bool) -> torch.Tensor:
    nblk, B, S = pa.shape
    da = torch.empty(B, S, device=pa.device, dtype=torch.float32)
    if gq is not None:
        assert gq.dtype == torch.float32 and gq.is_contiguous()
    check(lib().sc_cif_prepare_bwd(_p(pa), _p(pb), nblk, B, S, _p(a_clip), _p(pad), pad.stride(0), _p(ratio), _p(quantity), _p(gq),
                                   int(bool(scaled)), _p(da), _stream()), "sc_cif_prepare_bwd")
    return da


def cif_tail(alpha: torch.Tensor, csum: torch.Tensor, feat_len: torch.Tensor, out: torch.Tensor, T: int, thr: float, tail_thr: float,
             max_feat: int):
    """in place on ``feat_len`` and ``out`` [B,T+1,C]; returns (factor [B] fp32, extend [B] uint8)."""
    B, S = alpha.shape
    C = out.shape[2]
    assert tuple(out.shape) == (B, T + 1, C) and out.is_contiguous() and feat_len.dtype == torch.int64
    factor = torch.empty(B, device=alpha.device, dtype=torch.float32)
    extend = torch.empty(B, device=alpha.device, dtype=torch.uint8)
    check(lib().sc_cif_tail(_p(alpha), _p(csum), B, S, C, int(T), float(thr), float(tail_thr), int(max_feat), _p(feat_len), _p(out),
                            _p(factor), _p(extend), _stream()), "sc_cif_tail")
    return factor, extend


# ---- keyword -> sub-word vector quantiser (csrc/vq.hip) ---------------------------------------------------------------
def vq_prep(kw: torch.Tensor, eps: float = 1e-8):
    """kw [Nk, Et] fp32 -> (kwn_T [Et, Nkp] fp32 normalised + transposed, Nkp = roundup(Nk, 128); rnorm [Nk])."""
    Nk, Et = kw.shape
    assert kw.dtype == torch.float32 and kw.stride(1) == 1
    Nkp = (Nk + 127) // 128 * 128
    kwn_T = torch.empty(Et, Nkp, device=kw.device, dtype=torch.float32)
    rnorm = torch.empty(Nk, device=kw.device, dtype=torch.float32)
    check(lib().sc_vq_prep_f32(_p(kw), kw.stride(0), Nk, Et, float(eps), _p(kwn_T), Nkp, _p(rnorm), _stream()), "sc_vq_prep_f32")
    return kwn_T, rnorm


def split3_bf16(x: torch.Tensor, side: int, row_scale: Optional[torch.Tensor] = None, rows_pad: int = 128, cols_pad: int = 64) -> torch.Tensor:
    """x [R, E] fp32 (* row_scale [R]) -> its three-way bf16 split as the six K-blocks of an fp32-accurate product on the bf16 matrix
    pipe (sc_split3_bf16): [Rp, 6 Ep] bf16, Rp / Ep = R / E rounded up to ``rows_pad`` / ``cols_pad``; side 0 and side 1 pair up."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and side in (0, 1)
    R, E = x.shape
    Rp, Ep = -(-R // rows_pad) * rows_pad, -(-E // cols_pad) * cols_pad
    out = torch.empty(Rp, 6 * Ep, device=x.device, dtype=torch.bfloat16)
    check(lib().sc_split3_bf16(_p(x), x.stride(0), _p(row_scale), R, E, _p(out), Rp, Ep, side, _stream()), "sc_split3_bf16")
    return out


def cosine_scores_split(kw: torch.Tensor, rnorm: torch.Tensor, table_split: torch.Tensor, Vp: int) -> torch.Tensor:
    """cos [Nkp, Vp] fp32 = (kw * rnorm) . table_n^T to fp32 accuracy: ONE bf16 GEMM over the six K-blocks of the three-way splits
    (``table_split`` = split3_bf16(normalised table, side 1), rows padded to Vp)."""
    A = split3_bf16(kw, 0, row_scale=rnorm)
    K6 = A.shape[1]
    assert table_split.shape == (Vp, K6), (table_split.shape, Vp, K6)
    cos = torch.empty(A.shape[0], Vp, device=kw.device, dtype=torch.float32)
    gemm_raw(A, K6, table_split, K6, cos, Vp, A.shape[0], Vp, K6, out_f32=True)
    return cos


def sgemm_mfma(A: torch.Tensor, Bm: torch.Tensor, a_kmajor: bool = False, b_kmajor: bool = False, bias: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C [M, N] = A . B^T (+ bias) in exact fp32 on the matrix pipe.  A: [M, K] (or [K, M] with a_kmajor), B: [N, K] (nn.Linear
    layout; or [K, N] with b_kmajor); any M, N, K."""
    assert A.dtype == torch.float32 and Bm.dtype == torch.float32
    assert (A.stride(1) == 1 or A.shape[1] == 1) and (Bm.stride(1) == 1 or Bm.shape[1] == 1)
    K, M = A.shape if a_kmajor else A.shape[::-1]
    K2, N = Bm.shape if b_kmajor else Bm.shape[::-1]
    assert K == K2, (A.shape, Bm.shape)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.stride(1) == 1
    if _timer is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    S = int(lib().sc_sgemm_mfma_slices(M, N, K))          # few output tiles x long K: slices of the contraction, added in order
    if S > 1:
        part = torch.empty(S, M, N, device=A.device, dtype=torch.float32)
        check(lib().sc_sgemm_mfma_f32_split(_p(A), A.stride(0), int(a_kmajor), _p(Bm), Bm.stride(0), int(b_kmajor), _p(out), out.stride(0),
                                            M, N, K, _p(bias), _p(part), S, _stream()), "sc_sgemm_mfma_f32_split")
    else:
        check(lib().sc_sgemm_mfma_f32(_p(A), A.stride(0), int(a_kmajor), _p(Bm), Bm.stride(0), int(b_kmajor), _p(out), out.stride(0), M, N, K,
                                      _p(bias), _stream()), "sc_sgemm_mfma_f32")
    if _timer is not None:
        ev1.record()
        _timer.add("sgemm_mfma_f32", ev0, ev1, 2.0 * M * N * K)
    return out


def vq_rowstats(x: torch.Tensor, V: int, temp: float, mask_cols=(0, 2, 3)):
    """x [Nk, >= V] fp32 (masked columns overwritten with -inf) -> idx [Nk] int64, lse_t, lse_1, ent [Nk] fp32."""
    Nk = x.shape[0]
    assert x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] >= V
    dev = x.device
    idx = torch.empty(Nk, device=dev, dtype=torch.int64)
    lse_t, lse_1, ent = (torch.empty(Nk, device=dev, dtype=torch.float32) for _ in range(3))
    cols = [int(c) for c in mask_cols if 0 <= int(c) < V]
    assert len(cols) <= 4
    arr = (ctypes.c_int32 * 4)(*(cols + [-1] * (4 - len(cols))))
    check(lib().sc_vq_rowstats(_p(x), x.stride(0), Nk, V, float(temp), arr, len(cols), _p(idx), _p(lse_t), _p(lse_1), _p(ent), _stream()),
          "sc_vq_rowstats")
    return idx, lse_t, lse_1, ent


def vq_perplexity(x: torch.Tensor, V: int, idx: torch.Tensor, lse_1: torch.Tensor, nchunk: int = 16) -> torch.Tensor:
    """-> [code_perplexity, prob_perplexity] (fp32, device)."""
    Nk = x.shape[0]
    nchunk = max(1, min(nchunk, Nk))
    partial = torch.empty(nchunk, V, device=x.device, dtype=torch.float32)
    hist = torch.empty(V + 64, device=x.device, dtype=torch.int32)       # + 64 words: the entropy partials of the workgroups
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    check(lib().sc_vq_perplexity(_p(x), x.stride(0), Nk, V, _p(idx), _p(lse_1), _p(partial), nchunk, _p(hist), _p(out), _stream()),
          "sc_vq_perplexity")
    return out


def vq_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    Nk, Et = idx.numel(), table.shape[1]
    assert table.dtype == torch.float32 and table.stride(1) == 1 and idx.dtype == torch.int64
    out = torch.empty(Nk, Et, device=table.device, dtype=torch.float32)
    check(lib().sc_vq_gather_f32(_p(table), table.stride(0), _p(idx), _p(out), Et, Nk, Et, _stream()), "sc_vq_gather_f32")
    return out


def vq_onehot(idx: torch.Tensor, V: int) -> torch.Tensor:
    out = torch.empty(idx.numel(), V, device=idx.device, dtype=torch.float32)
    check(lib().sc_vq_onehot_f32(_p(idx), _p(out), V, idx.numel(), V, _stream()), "sc_vq_onehot_f32")
    return out


def vq_soft_bwd(x: torch.Tensor, lse_t: torch.Tensor, t: torch.Tensor, V: int, temp: float, out_bf16: bool, Vpad: Optional[int] = None):
    """dx [Nk, Vpad] = softmax(x / temp)(t - <softmax, t>) / temp; x, t [Nk, >= V] fp32."""
    Nk = x.shape[0]
    Vpad = V if Vpad is None else Vpad
    assert x.dtype == torch.float32 and t.dtype == torch.float32 and x.stride(1) == 1 and t.stride(1) == 1
    dx = torch.empty(Nk, Vpad, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    check(lib().sc_vq_soft_bwd(_p(x), x.stride(0), _p(lse_t), _p(t), t.stride(0), Nk, V, Vpad, float(temp), _p(dx), Vpad,
                               int(out_bf16), _stream()), "sc_vq_soft_bwd")
    return dx


def vq_norm_bwd(kw: torch.Tensor, rnorm: torch.Tensor, dy: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    Nk, Et = kw.shape
    assert kw.dtype == torch.float32 and dy.dtype == torch.float32 and kw.stride(1) == 1 and dy.stride(1) == 1
    dx = torch.empty(Nk, Et, device=kw.device, dtype=torch.float32)
    check(lib().sc_vq_norm_bwd_f32(_p(kw), kw.stride(0), _p(rnorm), _p(dy), dy.stride(0), float(eps), _p(dx), Et, Nk, Et, _stream()),
          "sc_vq_norm_bwd_f32")
    return dx


def bn_rows_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, run_mean: torch.Tensor, run_var: torch.Tensor,
                training: bool, momentum: float, eps: float):
    """BatchNorm over the rows of x [N, E] fp32 -> (y, save_mean, save_rstd); running estimates updated in place when training."""
    N, E = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    for t in (gamma, beta, run_mean, run_var):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == E
    y = torch.empty(N, E, device=x.device, dtype=torch.float32)
    sm = torch.empty(E, device=x.device, dtype=torch.float32) if training else None
    sr = torch.empty(E, device=x.device, dtype=torch.float32) if training else None
    check(lib().sc_bn_rows_fwd(_p(x), x.stride(0), N, E, _p(gamma), _p(beta), _p(run_mean), _p(run_var), int(training), float(momentum),
                               float(eps), _p(y), E, _p(sm), _p(sr), _stream()), "sc_bn_rows_fwd")
    return y, sm, sr


def bn_rows_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, save_mean: torch.Tensor, save_rstd: torch.Tensor):
    N, E = x.shape
    assert dy.dtype == torch.float32 and dy.stride(1) == 1 and tuple(dy.shape) == (N, E)
    dx = torch.empty(N, E, device=x.device, dtype=torch.float32)
    dg = torch.empty(E, device=x.device, dtype=torch.float32)
    db = torch.empty(E, device=x.device, dtype=torch.float32)
    check(lib().sc_bn_rows_bwd(_p(x), x.stride(0), _p(dy), dy.stride(0), N, E, _p(gamma), _p(save_mean), _p(save_rstd), _p(dx), E,
                               _p(dg), _p(db), _stream()), "sc_bn_rows_bwd")
    return dx, dg, db


def softmax_fwd(scores: torch.Tensor, key_mask: torch.Tensor, rows_per_batch: int, scale: float, drop_p: float = 0.0,
                drop_seed: int = 0):
    """P = softmax(scale * scores | key mask) as bf16 (+ the dropped copy in train mode); scores fp32 [..., n] contiguous,
    key_mask uint8 / bool [batches, n] (non-zero = padded key).  Returns (P, Pd) with Pd = P when drop_p == 0."""
    n = scores.shape[-1]
    rows = scores.numel() // n
    assert scores.dtype == torch.float32 and scores.is_contiguous() and key_mask.is_contiguous() and key_mask.element_size() == 1
    assert key_mask.shape[-1] == n and rows == key_mask.numel() // n * rows_per_batch
    P = torch.empty(scores.shape, device=scores.device, dtype=torch.bfloat16)
    Pd = torch.empty_like(P) if drop_p > 0.0 else None
    check(lib().sc_softmax_fwd(_p(scores), _p(key_mask), _p(P), _p(Pd), rows, n, rows_per_batch, float(scale), float(drop_p),
                               int(drop_seed) & 0xffffffff, _stream()), "sc_softmax_fwd")
    return P, (Pd if Pd is not None else P)


def softmax_bwd(dP: torch.Tensor, P: torch.Tensor, scale: float, drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    """dS = scale * P (dP' - rowsum(P dP')) as bf16, dP' = the dropout backward of dP (same mask as softmax_fwd)."""
    n = dP.shape[-1]
    rows = dP.numel() // n
    assert dP.dtype == torch.float32 and P.dtype == torch.bfloat16 and dP.is_contiguous() and P.is_contiguous() and P.shape == dP.shape
    dS = torch.empty_like(P)
    check(lib().sc_softmax_bwd(_p(dP), _p(P), _p(dS), rows, n, float(scale), float(drop_p), int(drop_seed) & 0xffffffff, _stream()),
          "sc_softmax_bwd")
    return dS


def act_bf16(u: torch.Tensor, act: int, df: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act 1 = erf-GELU, 2 = QuickGELU; with ``df``: df * act'(u)."""
    assert u.dtype == torch.bfloat16 and u.is_contiguous() and (df is None or (df.dtype == torch.bfloat16 and df.is_contiguous()))
    if out is None:
        out = torch.empty_like(u)
    check(lib().sc_act_bf16(_p(u), _p(df), _p(out), u.numel(), act, _stream()), "sc_act_bf16")
    return out


def layernorm_bf16(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: Optional[torch.Tensor] = None,
                   eps: float = 1e-5, act: int = 0) -> torch.Tensor:
    assert x.dim() == 2 and x.dtype == torch.bfloat16 and x.stride(1) == 1
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    check(lib().sc_layernorm_bf16(_p(x), x.stride(0), _p(gamma), _p(beta), _p(out), out.stride(0), x.shape[0],
                                  x.shape[1], float(eps), act, _stream()), "sc_layernorm_bf16")
    return out


def _wav_window(wav: torch.Tensor, L: Optional[int], wav_off: Optional[torch.Tensor]) -> int:
    """host-side shape contract of the kernels that read the caller's [B, ld] batch in place: unit sample stride, a padded length
    L <= the row width, int64 offsets on the batch's device (their range is the caller's contract: off_b + len_b <= width)"""
    assert wav.dim() == 2 and wav.stride(1) == 1 and wav.dtype == torch.float32, (wav.shape, wav.stride(), wav.dtype)
    L = wav.shape[1] if L is None else int(L)
    assert 0 < L <= wav.shape[1] and (wav.shape[0] == 1 or wav.stride(0) >= wav.shape[1]), (L, wav.shape, wav.stride())
    if wav_off is not None:
        assert wav_off.dtype == torch.int64 and wav_off.device == wav.device and wav_off.numel() == wav.shape[0] and wav_off.is_contiguous()
    return L


def wav_prep(wav: torch.Tensor, wav_len: torch.Tensor, out: torch.Tensor, normalize: bool, L: Optional[int] = None,
             wav_off: Optional[torch.Tensor] = None) -> None:
    """L / wav_off: the in-forward crop - utterance b is wav[b, off_b : off_b + len_b], padded length L (sc_wav_prep_crop)"""
    assert wav_len.dtype == torch.int64 and out.dtype == torch.float32
    L = _wav_window(wav, L, wav_off)
    check(lib().sc_wav_prep_crop(_p(wav), wav.stride(0), _p(wav_len), _p(wav_off), _p(out), out.stride(0), wav.shape[0], L, int(normalize),
                                 _stream()), "sc_wav_prep_crop")


def wav_prep_seg(wav: torch.Tensor, wav_len: torch.Tensor, out_flat: torch.Tensor, seg: "RowSegments", samples_per_row: int,
                 normalize: bool, L: Optional[int] = None, wav_off: Optional[torch.Tensor] = None) -> None:
    """waveform into the ragged layout: utterance b at sample samples_per_row * row0[b] of ONE flat fp32 buffer (sc_wav_prep_seg);
    L / wav_off as in wav_prep"""
    assert wav_len.dtype == torch.int64 and out_flat.dtype == torch.float32
    assert out_flat.numel() >= samples_per_row * seg.rows + 16
    L = _wav_window(wav, L, wav_off)
    check(lib().sc_wav_prep_seg_crop(_p(wav), wav.stride(0), _p(wav_len), _p(wav_off), _p(out_flat), seg.ref(), samples_per_row, L,
                                     int(normalize), _stream()), "sc_wav_prep_seg_crop")


def conv0_groupnorm_gelu_seg(wav: torch.Tensor, wav_len: torch.Tensor, wav_flat: torch.Tensor, seg: "RowSegments", samples_per_row: int,
                             w0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, T0: int, out: torch.Tensor, eps: float = 1e-5,
                             nchunk: int = 8, wav_off: Optional[torch.Tensor] = None) -> None:
    """conv layer 0 + GroupNorm + GELU on ragged rows.  The GroupNorm statistics run over the PADDED batch length T0 (fairseq feeds the
    zero-padded batch, speech_encoder_plus.py:75) and come straight from the caller's [B, L] batch masked by wav_len; the activation is
    written for the rows of the segment layout only."""
    B, C = wav.shape[0], w0.shape[0]
    partial = torch.empty(B * nchunk * 66, device=wav.device, dtype=torch.float64)
    scale = torch.empty(B, C, device=wav.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    L = lib()
    _wav_window(wav, None, wav_off)
    check(L.sc_conv0_stats_len_crop(_p(wav), wav.stride(0), _p(wav_len), _p(wav_off), B, T0, nchunk, _p(partial), _stream()),
          "sc_conv0_stats_len_crop")
    check(L.sc_conv0_finalize(_p(partial), nchunk, _p(w0), _p(gamma), _p(beta), B, C, T0, float(eps), _p(scale), _p(shift), _stream()),
          "sc_conv0_finalize")
    check(L.sc_conv0_gn_gelu_seg(_p(wav_flat), seg.ref(), samples_per_row, _p(w0), _p(scale), _p(shift), _p(out), C, _stream()),
          "sc_conv0_gn_gelu_seg")


def conv0_layernorm_gelu_seg(wav_flat: torch.Tensor, seg: "RowSegments", samples_per_row: int, w0: torch.Tensor, bias: Optional[torch.Tensor],
                             gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float = 1e-5) -> None:
    check(lib().sc_conv0_ln_gelu_seg(_p(wav_flat), seg.ref(), samples_per_row, _p(w0), _p(bias), _p(gamma), _p(beta), float(eps), _p(out),
                                     w0.shape[0], _stream()), "sc_conv0_ln_gelu_seg")


def conv0_groupnorm_gelu(wav_pad: torch.Tensor, w0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, T0: int,
                         R0: int, out: torch.Tensor, eps: float = 1e-5, nchunk: int = 8):
    """conv layer 0 + GroupNorm(C groups) over t < T0 + GELU -> out[B*R0, C] bf16 (channels-last).  Returns what the backward
    (conv0_groupnorm_gelu_bwd) needs: (scale, shift, Gram statistics, nchunk)."""
    B = wav_pad.shape[0]
    C = w0.shape[0]
    partial = torch.empty(B * nchunk * 66, device=wav_pad.device, dtype=torch.float64)
    scale = torch.empty(B, C, device=wav_pad.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    L = lib()
    check(L.sc_conv0_stats(_p(wav_pad), wav_pad.stride(0), B, T0, nchunk, _p(partial), _stream()), "sc_conv0_stats")
    check(L.sc_conv0_finalize(_p(partial), nchunk, _p(w0), _p(gamma), _p(beta), B, C, T0, float(eps), _p(scale),
                              _p(shift), _stream()), "sc_conv0_finalize")
    check(L.sc_conv0_gn_gelu(_p(wav_pad), wav_pad.stride(0), _p(w0), _p(scale), _p(shift), _p(out), B, R0, C, _stream()),
          "sc_conv0_gn_gelu")
    return scale, shift, partial, nchunk


def conv0_groupnorm_gelu_bwd(wav_pad: torch.Tensor, w0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, saved, dy: torch.Tensor,
                             T0: int, R0: int, eps: float = 1e-5, nwc: int = 32):
    """Parameter gradients of conv layer 0 + GroupNorm + GELU (sc_conv0_gn_bwd): dy [B*R0, C] bf16 -> (dW0 [C, 10], dgamma [C], dbeta [C])."""
    scale, shift, stats, nchunk = saved
    B, C = wav_pad.shape[0], w0.shape[0]
    dev = wav_pad.device
    partial = torch.empty(B, nwc, C, 12, device=dev, dtype=torch.float32)
    contrib = torch.empty(B, C * 12, device=dev, dtype=torch.float32)
    check(lib().sc_conv0_gn_bwd(_p(wav_pad), wav_pad.stride(0), _p(w0), _p(scale), _p(shift), _p(dy), _p(stats), nchunk, _p(gamma), _p(beta),
                                B, T0, R0, C, float(eps), _p(partial), nwc, _p(contrib), _stream()), "sc_conv0_gn_bwd")
    tot = torch.empty(C * 12, device=dev, dtype=torch.float32)
    colsum(contrib, C * 12, B, C * 12, tot)
    tot = tot.view(C, 12)
    return tot[:, :10], tot[:, 10], tot[:, 11]


def conv0_layernorm_gelu(wav_pad: torch.Tensor, w0: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                         beta: torch.Tensor, R0: int, out: torch.Tensor, eps: float = 1e-5) -> None:
    """conv layer 0 (+bias) + LayerNorm over channels + GELU ("layer_norm" extractor mode) -> out[B*R0, 512] bf16."""
    B, C = wav_pad.shape[0], w0.shape[0]
    check(lib().sc_conv0_ln_gelu(_p(wav_pad), wav_pad.stride(0), _p(w0), _p(bias), _p(gamma), _p(beta), float(eps), _p(out),
                                 B, R0, C, _stream()), "sc_conv0_ln_gelu")


def conv0_layernorm_gelu_bwd(wav_pad: torch.Tensor, w0: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                             dy: torch.Tensor, T0: int, R0: int, eps: float = 1e-5, nwc: int = 32):
    """Parameter gradients of conv layer 0 (+bias) + LayerNorm + GELU (sc_conv0_ln_bwd): dy [B*R0, 512] bf16 ->
    (dW0 [C, 10], dbias [C], dgamma [C], dbeta [C])."""
    B, C = wav_pad.shape[0], w0.shape[0]
    dev = wav_pad.device
    partial = torch.empty(B * nwc, C * 16, device=dev, dtype=torch.float32)
    check(lib().sc_conv0_ln_bwd(_p(wav_pad), wav_pad.stride(0), _p(w0), _p(bias), _p(gamma), _p(beta), float(eps), _p(dy), B, T0, R0, C,
                                _p(partial), nwc, _stream()), "sc_conv0_ln_bwd")
    tot = torch.empty(C * 16, device=dev, dtype=torch.float32)
    colsum(partial, C * 16, B * nwc, C * 16, tot)
    tot = tot.view(C, 16)
    return tot[:, :10], tot[:, 10], tot[:, 11], tot[:, 12]


def posconv_prep(x: torch.Tensor, valid_len: torch.Tensor, xz: torch.Tensor, xg: torch.Tensor, B: int, R: int, D: int,
                 G: int, halo: int) -> None:
    check(lib().sc_posconv_prep(_p(x), _p(valid_len), _p(xz), _p(xg), B, R, D, G, halo, _stream()), "sc_posconv_prep")


def posconv_prep_seg(x: torch.Tensor, valid_len: torch.Tensor, xz: torch.Tensor, xg: torch.Tensor, seg: "RowSegments", D: int, G: int,
                     halo: int) -> None:
    """ragged rows: xg = flat [G, rows + 2 halo B, D / G] slab buffer (numel at least that), utterance b at slab row row0[b] + 2 halo b"""
    assert xg.numel() >= (seg.rows + 2 * halo * seg.B) * D
    check(lib().sc_posconv_prep_seg(_p(x), _p(valid_len), _p(xz), _p(xg), seg.ref(), D, G, halo, _stream()), "sc_posconv_prep_seg")


def posconv_seg(xg: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor], out: torch.Tensor,
                seg: "RowSegments", D: int, G: int, Kp: int, alg_rows: Optional[int] = None) -> None:
    assert xg.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    if _timer is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().sc_posconv_seg_bf16(_p(xg), _p(w), _p(bias), _p(residual), _p(out), seg.ref(), D, G, Kp, _stream()), "sc_posconv_seg_bf16")
    if _timer is not None:
        ev1.record()
        _timer.add("posconv", ev0, ev1, 2.0 * (seg.rows if alg_rows is None else alg_rows) * D * Kp * (D // G))


def posconv(xg: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor], out: torch.Tensor,
            B: int, R: int, D: int, G: int, Kp: int, alg_rows: Optional[int] = None) -> None:
    """HuBERT positional convolution + bias + GELU + residual on the slab layout of ``posconv_prep`` (sc_posconv_bf16):
    xg [G, B, Rp, D/G] bf16, w [G, D/G, Kp*D/G] bf16 tap-major, out / residual [B*R, D] bf16."""
    assert xg.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    assert xg.is_contiguous() and w.is_contiguous() and out.is_contiguous() and xg.shape[0] == G and xg.shape[1] == B
    Rp = xg.shape[2]
    if _timer is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().sc_posconv_bf16(_p(xg), _p(w), _p(bias), _p(residual), _p(out), B, R, D, G, Kp, Rp, _stream()), "sc_posconv_bf16")
    if _timer is not None:
        ev1.record()
        rows = R if alg_rows is None else alg_rows
        _timer.add("posconv", ev0, ev1, 2.0 * B * rows * D * Kp * (D // G))


class LazyStates:
    """Hidden states kept RAW (in front of their LayerNorm) with row statistics (speech_encoder: LayerNorm folded into the encoder
    GEMMs): ``stats`` [NL, B*R, 8, 2] fp32, ``gamma`` / ``beta`` [NL, D] fp32, layers >= ``first_lazy`` are raw, ``ns`` valid strips."""

    def __init__(self, stats, gamma, beta, first_lazy: int, ns: int, eps: float):
        self.stats, self.gamma, self.beta, self.first_lazy, self.ns, self.eps = stats, gamma, beta, int(first_lazy), int(ns), float(eps)


def wsum_fwd(h: torch.Tensor, w_softmax: torch.Tensor, out: torch.Tensor, B: int, R: int, D: int, row_off: int,
             normalize: bool = False, lazy: Optional[LazyStates] = None, seg: Optional["RowSegments"] = None) -> None:
    """``seg``: h [NL, seg.rows, D] in the ragged layout -> out [B, R, D] uniform (every row written, zero outside the utterances)"""
    NL = h.shape[0]
    if seg is not None:
        assert lazy is None and h.shape[1] == seg.rows and B == seg.B
        assert h.dtype == torch.bfloat16 and w_softmax.dtype == torch.float32 and out.dtype == torch.bfloat16
        check(lib().sc_wsum_fwd_seg(_p(h), _p(w_softmax), NL, _p(out), seg.ref(), R, D, row_off, int(normalize), _stream()), "sc_wsum_fwd_seg")
        return
    if lazy is not None:
        assert not normalize
        check(lib().sc_wsum_lazy_fwd(_p(h), _p(w_softmax), NL, _p(out), B, R, D, row_off, _p(lazy.stats), _p(lazy.gamma), _p(lazy.beta),
                                     lazy.first_lazy, lazy.ns, lazy.eps, _stream()), "sc_wsum_lazy_fwd")
        return
    assert h.dtype == torch.bfloat16 and w_softmax.dtype == torch.float32 and out.dtype == torch.bfloat16
    check(lib().sc_wsum_fwd(_p(h), _p(w_softmax), NL, _p(out), B, R, D, row_off, int(normalize), _stream()), "sc_wsum_fwd")


def wsum_bwd(h: torch.Tensor, g: torch.Tensor, B: int, R: int, D: int, row_off: int, nblk: int = 1024,
             normalize: bool = False, lazy: Optional[LazyStates] = None) -> torch.Tensor:
    """returns d(softmaxed weights)[NL] up to a common shift: <g, h_n - h_last> (sc_wsum_bwd: the callers' softmax projection
    w_n (d_n - sum_m w_m d_m) does not see the shift, and the fp32 sums keep the digits the projection needs)."""
    NL = h.shape[0]
    assert g.dtype == torch.float32
    part = torch.empty(nblk, NL, device=h.device, dtype=torch.float32)
    if lazy is not None:
        assert not normalize
        check(lib().sc_wsum_lazy_bwd(_p(h), _p(g), NL, _p(part), nblk, B, R, D, row_off, _p(lazy.stats), _p(lazy.gamma), _p(lazy.beta),
                                     lazy.first_lazy, lazy.ns, lazy.eps, _stream()), "sc_wsum_lazy_bwd")
        return part.sum(0)
    check(lib().sc_wsum_bwd(_p(h), _p(g), NL, _p(part), nblk, B, R, D, row_off, int(normalize), _stream()), "sc_wsum_bwd")
    return part.sum(0)


def cls_scores(X: torch.Tensor, vec: torch.Tensor, per_batch: bool, B: int, R: int, D: int, H: int) -> torch.Tensor:
    scores = torch.empty(B, H, R, device=X.device, dtype=torch.float32)
    assert vec.dtype == torch.float32 and vec.is_contiguous()
    check(lib().sc_cls_scores(_p(X), _p(vec), H * D if per_batch else 0, _p(scores), B, R, D, H, _stream()), "sc_cls_scores")
    return scores


def cls_pool_fwd(X: torch.Tensor, scores: torch.Tensor, lens: torch.Tensor, B: int, R: int, D: int, H: int,
                 mult: Optional[torch.Tensor] = None, want_psum: bool = False):
    """``mult`` [B,H,R] fp32 = dropout multipliers of the attention weights (0 or 1/(1-p)); p is returned un-masked.  ``want_psum``:
    also sum_s p mult [B,H] (the weight of the value bias when dropped weights no longer sum to 1)."""
    p = torch.empty(B, H, R, device=X.device, dtype=torch.float32)
    m = torch.empty(B, H, D, device=X.device, dtype=torch.float32)
    psum = torch.empty(B, H, device=X.device, dtype=torch.float32) if want_psum else None
    if mult is not None:
        assert mult.shape == (B, H, R) and mult.dtype == torch.float32 and mult.is_contiguous()
    check(lib().sc_cls_pool_fwd(_p(X), _p(scores), _p(lens), _p(p), _p(m), B, R, D, H, _p(mult), _p(psum), _stream()), "sc_cls_pool_fwd")
    return (p, m, psum) if want_psum else (p, m)


def cls_pool_bwd(X: torch.Tensor, p: torch.Tensor, dp: torch.Tensor, dm: torch.Tensor, a: torch.Tensor, lens: torch.Tensor,
                 B: int, R: int, D: int, H: int, mult: Optional[torch.Tensor] = None, cbias: Optional[torch.Tensor] = None):
    """``cbias`` [B,H]: ``dp`` is the raw X . dm and the kernel forms (dp + cbias) * mult itself; otherwise dp arrives in that form."""
    dX = torch.empty(B, R, D, device=X.device, dtype=torch.float32)
    da = torch.empty(B, H, D, device=X.device, dtype=torch.float32)
    check(lib().sc_cls_pool_bwd(_p(X), _p(p), _p(dp), _p(dm), _p(a), _p(lens), _p(dX), _p(da), B, R, D, H, _p(mult), _p(cbias), _stream()),
          "sc_cls_pool_bwd")
    return dX, da


def sgemm(A: torch.Tensor, sai: int, sak: int, Bm: torch.Tensor, sbj: int, sbk: int, M: int, N: int, K: int,
          alpha: float = 1.0, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[i, j] = alpha sum_k A[i sai + k sak] Bm[j sbj + k sbk] (+ bias[j]): the loss's logits / gradient products.  Routed through
    sc_sgemm_f32_ex so that the few-tile case (Bg = 64 per GPU: one 64 x 64 tile) is split along K over the chip."""
    assert A.dtype == torch.float32 and Bm.dtype == torch.float32
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    sgemm_ex(A, (sai, sak, 0), Bm, (sbj, sbk, 0), out, out.stride(0), M, N, K, alpha=alpha, bias=bias)
    return out


_INFONCE_WS = {}


def _infonce_workspace(Bg: int, device) -> torch.Tensor:
    """Per (device, stream, Bg) workspace of sc_infonce_fwd; zeroed once (it holds the ticket word, which every launch resets)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, Bg)
    ws = _INFONCE_WS.get(key)
    if ws is None:
        ws = _INFONCE_WS[key] = torch.zeros(int(lib().sc_infonce_workspace_floats(Bg)), device=device, dtype=torch.float32)
    return ws


def infonce_fwd(A: torch.Tensor, Bm: torch.Tensor, ids: Optional[torch.Tensor], inv_temp: torch.Tensor, margin: float = 0.0,
                dcl: bool = False, a2b: bool = True, b2a: bool = True):
    """-> (loss [1], logits [Bg, Bg], lse_row [Bg], lse_col [Bg]); A, B [Bg, E] fp32 contiguous, inv_temp a device scalar tensor."""
    Bg, E = A.shape
    assert A.dtype == torch.float32 and Bm.dtype == torch.float32 and A.is_contiguous() and Bm.is_contiguous() and Bm.shape == A.shape
    assert inv_temp.dtype == torch.float32 and inv_temp.is_cuda and inv_temp.numel() == 1
    if ids is not None:
        assert ids.dtype == torch.int64 and ids.is_contiguous()
    A, Bm = aligned16(A), aligned16(Bm)
    dev = A.device
    logits = torch.empty(Bg, Bg, device=dev, dtype=torch.float32)
    lse_row, lse_col = torch.empty(Bg, device=dev, dtype=torch.float32), torch.empty(Bg, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    ws = _infonce_workspace(Bg, dev)
    check(lib().sc_infonce_fwd(_p(A), _p(Bm), Bg, E, _p(ids), _p(inv_temp), float(margin), int(dcl), int(a2b), int(b2a), _p(logits),
                               _p(lse_row), _p(lse_col), _p(loss), _p(ws), _stream()), "sc_infonce_fwd")
    return loss, logits, lse_row, lse_col


def infonce_grad(logits: torch.Tensor, ids: Optional[torch.Tensor], lse_row: torch.Tensor, lse_col: torch.Tensor,
                 gscale: torch.Tensor, inv_temp: torch.Tensor, margin: float = 0.0, dcl: bool = False, a2b: bool = True,
                 b2a: bool = True):
    """-> (G [Bg, Bg] = inv_temp * gscale * dloss/dlogits, dot [Bg] with sum(dot) = d loss / d inv_temp * gscale)."""
    Bg = logits.shape[0]
    G = torch.empty_like(logits)
    dot = torch.empty(Bg, device=logits.device, dtype=torch.float32)
    check(lib().sc_infonce_grad(_p(logits), _p(ids), _p(lse_row), _p(lse_col), Bg, _p(gscale), _p(inv_temp), float(margin), int(dcl),
                                int(a2b), int(b2a), _p(G), _p(dot), _stream()), "sc_infonce_grad")
    return G, dot


def sumsq(x: torch.Tensor, nblk: int = 1024) -> torch.Tensor:
    part = torch.empty(nblk, device=x.device, dtype=torch.float32)
    check(lib().sc_sumsq_f32(_p(x), x.numel(), _p(part), nblk, _stream()), "sc_sumsq_f32")
    return part


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float,
              eps: float, weight_decay: float, step: int, gn_partial: Optional[torch.Tensor], max_norm: float) -> None:
    nblk = gn_partial.numel() if gn_partial is not None else 0
    check(lib().sc_adam_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                            _p(gn_partial), nblk, max_norm, _stream()), "sc_adam_f32")


# ---- fp32 one-row-per-utterance head tail (csrc/headtail.hip) ------------------------------------------------------
_SPLITK_WS = {}


def _splitk_workspace(device) -> torch.Tensor:
    """Scratch for split-K partial sums, one buffer per (device, stream): calls on one stream are serialised, calls on different
    streams (the head's parameter-only backward half, the trainer's prefetch) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(8 << 20, device=device, dtype=torch.float32)
    return ws


def sgemm_ex(A: torch.Tensor, sa, Bm: torch.Tensor, sb, C: torch.Tensor, ldc: int, M: int, N: int, K: int, nbatch: int = 1,
             scz: int = 0, alpha: float = 1.0, beta: float = 0.0, bias: Optional[torch.Tensor] = None, sbiasz: int = 0) -> None:
    """C[z][i,j] = alpha sum_k A[z][i*sa0 + k*sa1] Bm[z][j*sb0 + k*sb1] (+ bias[z][j]) + beta C[z][i,j];
    sa / sb = (row stride, k stride, batch stride) in elements; pointers are the tensors' data_ptr()."""
    assert A.dtype == torch.float32 and Bm.dtype == torch.float32 and C.dtype == torch.float32
    ws = _splitk_workspace(A.device)
    check(lib().sc_sgemm_f32_ex(_p(A), sa[0], sa[1], sa[2], _p(Bm), sb[0], sb[1], sb[2], _p(C), ldc, scz, M, N, K, nbatch,
                                float(alpha), float(beta), _p(bias), sbiasz, _p(ws), ws.numel(), _stream()), "sc_sgemm_f32_ex")


def rowln_fwd(x: torch.Tensor, res: Optional[torch.Tensor], res_stride: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    rows, D = x.shape
    y, xhat = torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib().sc_rowln_f32_fwd(_p(x), _p(res), res_stride, _p(gamma), _p(beta), _p(y), _p(xhat), _p(rstd), rows, D, float(eps),
                                 _stream()), "sc_rowln_f32_fwd")
    return y, xhat, rstd


def rowln_bwd(dy: torch.Tensor, xhat: torch.Tensor, gamma: torch.Tensor, rstd: torch.Tensor, dgamma_acc: torch.Tensor,
              dbeta_acc: torch.Tensor) -> torch.Tensor:
    rows, D = dy.shape
    dx = torch.empty_like(dy)
    check(lib().sc_rowln_f32_bwd(_p(dy), _p(xhat), _p(gamma), _p(rstd), _p(dx), _p(dgamma_acc), _p(dbeta_acc), rows, D, _stream()),
          "sc_rowln_f32_bwd")
    return dx


def gelu_f32(u: torch.Tensor, df: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(u)
    check(lib().sc_gelu_f32(_p(u), _p(df), _p(out), u.numel(), _stream()), "sc_gelu_f32")
    return out


def colsum(x: torch.Tensor, ld: int, rows: int, cols: int, out: torch.Tensor, alpha: float = 1.0, beta: float = 0.0) -> None:
    check(lib().sc_colsum_f32(_p(x), ld, rows, cols, _p(out), float(alpha), float(beta), _stream()), "sc_colsum_f32")


def headmask(q: torch.Tensor, Qm: torch.Tensor, H: int, D: int, dh: int, gather: bool) -> None:
    check(lib().sc_headmask_f32(_p(q), _p(Qm), H, D, dh, int(gather), _stream()), "sc_headmask_f32")


# ---------------------------------------------------------------------------------------------- row tail of the head (csrc/rowtail.hip)
class Slices:
    """A [rows, cols] fp32 matrix given as ``ns`` partial slices [ns, rows, cols] that its consumer adds in order (a split product of
    sc_rt_gemm; ns = 1: an ordinary matrix)."""

    def __init__(self, t: torch.Tensor):
        assert t.dim() == 3 and t.dtype == torch.float32 and t.is_contiguous()
        self.t, self.ns, self.rows, self.cols = t, t.shape[0], t.shape[1], t.shape[2]

    def total(self) -> torch.Tensor:          # (tests / debugging: the consumer kernels never materialise this)
        return self.t.sum(0)


def rt_gemm(A, Bm: torch.Tensor, M: int, N: int, K: int, *, a_kmajor: bool = False, b_kmajor: bool = False, lda: Optional[int] = None,
            ldb: Optional[int] = None, a_bias: Optional[torch.Tensor] = None, a_rowscale: Optional[torch.Tensor] = None, a_group: int = 1,
            split: bool = False, out: Optional[torch.Tensor] = None, ldc: Optional[int] = None, alpha: float = 1.0, beta: float = 0.0,
            bias: Optional[torch.Tensor] = None, act: int = 0, U: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
            gb: Optional[torch.Tensor] = None, nbatch: int = 1, a_z: int = 0, b_z: int = 0, c_z: int = 0, bias_z: int = 0, gb_z: int = 0):
    """C = epi(alpha A . B^T) (sc_rt_gemm).  ``A``: a tensor or ``Slices`` (row-major [M, K]; or [K, M] with a_kmajor), ``Bm`` [N, K] (or
    [K, N] with b_kmajor).  ``split`` = True: the contraction is split over the chip and a ``Slices`` of raw partial products comes back
    (its consumer adds them, with the bias); otherwise the epilogue runs and the [M, N] result (``out`` or a fresh tensor) is returned."""
    a = RtGemmArgs()
    if isinstance(A, Slices):
        assert not a_kmajor
        a.A, a.a_ns, a.a_slice = _p(A.t), A.ns, A.rows * A.cols
        lda = A.cols if lda is None else lda
    else:
        assert A.dtype == torch.float32
        a.A, a.a_ns, a.a_slice = _p(A), 1, 0
        lda = A.stride(0) if lda is None else lda
    assert Bm.dtype == torch.float32
    a.lda, a.a_z, a.a_kmajor = int(lda), int(a_z), int(a_kmajor)
    a.a_bias, a.a_rowscale, a.a_group = _p(a_bias), _p(a_rowscale), int(a_group)
    a.a_nscale = int(a_rowscale.shape[-1]) if a_rowscale is not None else 0
    a.B, a.ldb, a.b_z, a.b_kmajor, a.nbatch = _p(Bm), int(Bm.stride(0) if ldb is None else ldb), int(b_z), int(b_kmajor), int(nbatch)
    a.M, a.N, a.K = int(M), int(N), int(K)
    a.alpha, a.beta = float(alpha), float(beta)
    dev = Bm.device
    if split:
        S = int(lib().sc_rt_gemm_slices(M, N, K, nbatch))
        assert nbatch == 1 or c_z > 0
        res = torch.empty(S, M, N if nbatch == 1 else ldc, device=dev, dtype=torch.float32)
        a.C, a.ldc, a.c_slice, a.c_z, a.S = _p(res), res.shape[2], M * res.shape[2], int(c_z), S
        check(lib().sc_rt_gemm(ctypes.byref(a), _stream()), "sc_rt_gemm")
        return Slices(res)
    if out is None:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
    a.C, a.ldc, a.c_slice, a.c_z, a.S = _p(out), int(out.stride(0) if ldc is None else ldc), 0, int(c_z), 1
    a.U = _p(U)
    a.bias, a.bias_z, a.act = _p(bias), int(bias_z), int(act)
    a.drop_p, a.drop_seed = float(drop_p), int(drop_seed) & 0xffffffff
    a.gb, a.gb_z = _p(gb), int(gb_z)
    check(lib().sc_rt_gemm(ctypes.byref(a), _stream()), "sc_rt_gemm")
    return out


def rt_ln_fwd(y: Slices, bias, res, res_stride: int, g1, b1, eps1: float, g2=None, b2=None, eps2: float = 0.0, drop_p: float = 0.0,
              drop_seed: int = 0):
    """z = (sum_s y_s + bias) * dropout + res ; LN1 [; LN2] -> (out1, xhat1, rstd1[, out2, xhat2, rstd2])."""
    rows, D = y.rows, y.cols
    dev = y.t.device
    f = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
    a = RtLnArgs()
    a.y, a.y_slice, a.ns, a.rows, a.D = _p(y.t), rows * D, y.ns, rows, D
    a.bias, a.drop_p, a.drop_seed = _p(bias), float(drop_p), int(drop_seed) & 0xffffffff
    a.res, a.res_stride = _p(res), int(res_stride)
    o1, h1, r1 = f(rows, D), f(rows, D), f(rows)
    a.g1, a.b1, a.eps1, a.out1, a.xhat1, a.rstd1 = _p(g1), _p(b1), float(eps1), _p(o1), _p(h1), _p(r1)
    outs = (o1, h1, r1)
    if g2 is not None:
        o2, h2, r2 = f(rows, D), f(rows, D), f(rows)
        a.g2, a.b2, a.eps2, a.out2, a.xhat2, a.rstd2 = _p(g2), _p(b2), float(eps2), _p(o2), _p(h2), _p(r2)
        outs = outs + (o2, h2, r2)
    check(lib().sc_rt_ln_fwd(ctypes.byref(a), _stream()), "sc_rt_ln_fwd")
    return outs


def rt_ln_bwd(dy: Slices, add, xhat, gamma, rstd, dgamma_acc, dbeta_acc, want_masked: bool = False, drop_p: float = 0.0, drop_seed: int = 0):
    """LayerNorm backward over a sliced incoming gradient (+ ``add``); returns dx (and dx * dropout multiplier when asked)."""
    rows, D = dy.rows, dy.cols
    a = RtLnBwdArgs()
    dx = torch.empty(rows, D, device=dy.t.device, dtype=torch.float32)
    dxm = torch.empty_like(dx) if want_masked else None
    a.dy, a.dy_slice, a.ns, a.rows, a.D = _p(dy.t), rows * D, dy.ns, rows, D
    a.add, a.xhat, a.gamma, a.rstd, a.dx, a.dx_masked = _p(add), _p(xhat), _p(gamma), _p(rstd), _p(dx), _p(dxm)
    a.drop_p, a.drop_seed, a.dgamma, a.dbeta = float(drop_p), int(drop_seed) & 0xffffffff, _p(dgamma_acc), _p(dbeta_acc)
    check(lib().sc_rt_ln_bwd(ctypes.byref(a), _stream()), "sc_rt_ln_bwd")
    return (dx, dxm) if want_masked else dx


def rt_l2norm_fwd(y: Slices, bias, keep_x: bool = True):
    """x = sum_s y_s + bias ; e = x / |x| -> (x or None, e, 1 / |x|)."""
    rows, D = y.rows, y.cols
    dev = y.t.device
    x = torch.empty(rows, D, device=dev, dtype=torch.float32) if keep_x else None
    e = torch.empty(rows, D, device=dev, dtype=torch.float32)
    rn = torch.empty(rows, device=dev, dtype=torch.float32)
    check(lib().sc_rt_l2norm_fwd(_p(y.t), rows * D, y.ns, _p(bias), _p(x), _p(e), _p(rn), rows, D, _stream()), "sc_rt_l2norm_fwd")
    return x, e, rn


def rt_l2norm_bwd(g: torch.Tensor, e: torch.Tensor, rn: torch.Tensor) -> torch.Tensor:
    rows, D = e.shape
    dx = torch.empty_like(e)
    g = g.float().contiguous()
    check(lib().sc_rt_l2norm_bwd(_p(g), _p(e), _p(rn), _p(dx), rows, D, _stream()), "sc_rt_l2norm_bwd")
    return dx


def rt_elem(y: Slices, mode: int, bias=None, rowscale=None, group: int = 1, u: Optional[torch.Tensor] = None, drop_p: float = 0.0,
            drop_seed: int = 0):
    """mode 0: sum_s y_s + bias * rowscale -> out ; 1: -> (u, gelu(u) * dropout) ; 2: (sum_s y_s) * dropout * gelu'(u) -> out."""
    rows, D = y.rows, y.cols
    out = torch.empty(rows, D, device=y.t.device, dtype=torch.float32)
    if mode == 1:
        u = torch.empty_like(out)
    nscale = int(rowscale.shape[-1]) if rowscale is not None else 0
    check(lib().sc_rt_elem(_p(y.t), rows * D, y.ns, _p(bias), _p(rowscale), int(group), nscale, _p(out), _p(u), rows, D, int(mode), float(drop_p),
                           int(drop_seed) & 0xffffffff, _stream()), "sc_rt_elem")
    return (u, out) if mode == 1 else out


def rt_value_bias_bwd(dctx: torch.Tensor, bv: torch.Tensor, psum: torch.Tensor, gbv_acc: torch.Tensor, H: int) -> torch.Tensor:
    """-> cbias [B,H] = sum_j dctx[b,h,j] bv[h,j] ; gbv_acc[h,j] += sum_b dctx[b,h,j] psum[b,h]  (one launch)"""
    B, D = dctx.shape
    cb = torch.empty(B, H, device=dctx.device, dtype=torch.float32)
    check(lib().sc_rt_value_bias_bwd(_p(dctx), _p(bv), _p(psum), _p(cb), _p(gbv_acc), B, D, H, _stream()), "sc_rt_value_bias_bwd")
    return cb


def wsum_bwd_logits(h: torch.Tensor, g: torch.Tensor, w_soft: torch.Tensor, B: int, R: int, D: int, row_off: int, nblk: int = 1024,
                    normalize: bool = False, lazy: Optional[LazyStates] = None, seg: Optional["RowSegments"] = None) -> torch.Tensor:
    """Gradient of the weighted-sum LOGITS in two launches: the partial sums <g, h_n - h_last> per block (sc_wsum_bwd) and their
    reduction fused with the softmax backward w_n (d_n - sum_m w_m d_m) (sc_rt_softmax_bwd_reduce)."""
    NL = h.shape[0]
    assert g.dtype in (torch.float32, torch.bfloat16) and g.is_contiguous()      # bf16: as the attention block's GEMM wrote it
    flags = int(normalize) | (2 if g.dtype == torch.bfloat16 else 0)
    part = torch.empty(nblk, NL, device=h.device, dtype=torch.float32)
    if seg is not None:
        assert lazy is None and h.shape[1] == seg.rows and B == seg.B
        check(lib().sc_wsum_bwd_seg(_p(h), _p(g), NL, _p(part), nblk, seg.ref(), R, D, row_off, flags, _stream()), "sc_wsum_bwd_seg")
    elif lazy is not None:
        assert not normalize
        if g.dtype != torch.float32:
            g = g.float()
        check(lib().sc_wsum_lazy_bwd(_p(h), _p(g), NL, _p(part), nblk, B, R, D, row_off, _p(lazy.stats), _p(lazy.gamma), _p(lazy.beta),
                                     lazy.first_lazy, lazy.ns, lazy.eps, _stream()), "sc_wsum_lazy_bwd")
    else:
        check(lib().sc_wsum_bwd(_p(h), _p(g), NL, _p(part), nblk, B, R, D, row_off, flags, _stream()), "sc_wsum_bwd")
    out = torch.empty(NL, device=h.device, dtype=torch.float32)
    check(lib().sc_rt_softmax_bwd_reduce(_p(part), nblk, NL, _p(w_soft), _p(out), _stream()), "sc_rt_softmax_bwd_reduce")
    return out


def cif_head_fwd(y: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, p1: float, seed1: int, p2: float, seed2: int) -> torch.Tensor:
    """alpha [rows] = sigmoid(bias + sum_c w[c] m2 relu(m1 y[row, c]))  (sc_cif_head_fwd; y [rows, C] fp32 contiguous)"""
    rows, C = y.shape
    alpha = torch.empty(rows, device=y.device, dtype=torch.float32)
    check(lib().sc_cif_head_fwd(_p(y), y.stride(0), _p(w), _p(bias), _p(alpha), rows, C, float(p1), int(seed1) & 0xffffffff, float(p2),
                                int(seed2) & 0xffffffff, _stream()), "sc_cif_head_fwd")
    return alpha


def cif_head_bwd(y: torch.Tensor, w: torch.Tensor, alpha: torch.Tensor, dalpha: torch.Tensor, p1: float, seed1: int, p2: float, seed2: int,
                 nblk: int = 512, dy_out: Optional[torch.Tensor] = None, acc=None):
    """-> dy [rows, C], dw [C], db [1]; ``acc`` = (gw, gb) fp32 gradient buffers of C and 1 elements the two reductions ADD into
    (-> dy, None, None)"""
    rows, C = y.shape
    dy = torch.empty_like(y) if dy_out is None else dy_out             # dy_out: bf16 [rows, C] rows (what the conv's dgrad GEMM reads)
    assert dy.dtype in (torch.float32, torch.bfloat16) and dy.stride(1) == 1 and tuple(dy.shape) == (rows, C)
    pw = torch.empty(nblk, C, device=y.device, dtype=torch.float32)
    pb = torch.empty(nblk, device=y.device, dtype=torch.float32)
    check(lib().sc_cif_head_bwd_rows(_p(y), y.stride(0), _p(w), _p(alpha), _p(dalpha), _p(dy), int(dy.dtype == torch.bfloat16), dy.stride(0), _p(pw),
                                     _p(pb), nblk, rows, C, float(p1), int(seed1) & 0xffffffff, float(p2), int(seed2) & 0xffffffff, _stream()),
          "sc_cif_head_bwd")
    if acc is not None:
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in acc) and acc[0].numel() == C and acc[1].numel() == 1
        colsum(pw, C, nblk, C, acc[0], beta=1.0)
        colsum(pb.view(nblk, 1), 1, nblk, 1, acc[1], beta=1.0)
        return dy, None, None
    dw = torch.empty(C, device=y.device, dtype=torch.float32)
    db = torch.empty(1, device=y.device, dtype=torch.float32)
    colsum(pw, C, nblk, C, dw)
    colsum(pb.view(nblk, 1), 1, nblk, 1, db)
    return dy, dw, db


def conv_overlap_add(dcols: torch.Tensor, C: int, u: Optional[torch.Tensor] = None, act: int = 1) -> torch.Tensor:
    """dcols [M, 3C] bf16 (window gradients of a k = 3 / stride 2 conv) -> input-row gradients [2M, C] bf16 (sc_conv_overlap_add_bf16);
    ``u`` [2M, C] bf16: also through the activation below, dx * act'(u) (sc_conv_overlap_add_act_bf16)"""
    M = dcols.shape[0]
    assert dcols.dtype == torch.bfloat16 and dcols.is_contiguous() and dcols.shape[1] == 3 * C
    dx = torch.empty(2 * M, C, device=dcols.device, dtype=torch.bfloat16)
    if u is not None:
        assert u.dtype == torch.bfloat16 and u.is_contiguous() and u.numel() >= 2 * M * C
        check(lib().sc_conv_overlap_add_act_bf16(_p(dcols), _p(u), _p(dx), M, C, int(act), _stream()), "sc_conv_overlap_add_act_bf16")
    else:
        check(lib().sc_conv_overlap_add_bf16(_p(dcols), _p(dx), M, C, _stream()), "sc_conv_overlap_add_bf16")
    return dx


# ---- keyword prompt of the cascaded branches in the text tower's packed rows (csrc/prompt.hip) ------------------------------------------
def prompt_assemble(keywords: torch.Tensor, count: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor, Bp: int, SEG: int, n_pos: int,
                    clamped: Optional[torch.Tensor] = None):
    """keywords [B, N, W] fp32 (last two dims dense), count [B] int64, tok [3, W] (SOT, EOT, token 0), pos [>= n_pos, W] ->
    X [Bp * SEG, W] bf16 (prefix rows of [SOT, kw.., EOT, 0..] + pos; zero elsewhere), eot_row [B] int32 (rows of X)"""
    B, N, W = keywords.shape
    assert keywords.dtype == torch.float32 and keywords.stride(2) == 1 and (N == 0 or keywords.stride(1) == W)
    assert count.dtype == torch.int64 and count.is_contiguous() and tok.dtype == torch.float32 and tok.is_contiguous() and tuple(tok.shape) == (3, W)
    assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[0] >= n_pos and pos.shape[1] == W
    X = torch.empty(Bp * SEG, W, device=keywords.device, dtype=torch.bfloat16)
    eot_row = torch.empty(B, device=keywords.device, dtype=torch.int32)
    check(lib().sc_prompt_assemble(_p(keywords), keywords.stride(0) if B > 1 else N * W, _p(count), _p(tok), _p(pos), _p(X), _p(eot_row),
                                   _p(clamped), B, Bp, N, W, SEG, n_pos, _stream()), "sc_prompt_assemble")
    return X, eot_row


def prompt_assemble_bwd(dX: torch.Tensor, count: torch.Tensor, B: int, N: int, SEG: int, n_pos: int) -> torch.Tensor:
    """dX [>= B * SEG, W] bf16 -> dkeywords [B, N, W] fp32 (rows behind a sample's keyword count: zero)"""
    W = dX.shape[1]
    assert dX.dtype == torch.bfloat16 and dX.is_contiguous()
    dk = torch.empty(B, N, W, device=dX.device, dtype=torch.float32)
    if N > 0:
        check(lib().sc_prompt_assemble_bwd(_p(dX), _p(count), _p(dk), N * W, B, N, W, SEG, n_pos, _stream()), "sc_prompt_assemble_bwd")
    return dk


def rows_gather(X: torch.Tensor, row: torch.Tensor) -> torch.Tensor:
    """out[b] = float(X[row[b]]): X [M, W] bf16 contiguous, row [B] int32"""
    assert X.dtype == torch.bfloat16 and X.is_contiguous() and row.dtype == torch.int32
    out = torch.empty(row.numel(), X.shape[1], device=X.device, dtype=torch.float32)
    check(lib().sc_rows_gather_bf16(_p(X), _p(row), _p(out), row.numel(), X.shape[1], _stream()), "sc_rows_gather_bf16")
    return out


def rows_scatter(d: torch.Tensor, row: torch.Tensor, M: int, SEG: int) -> torch.Tensor:
    """dX [M, W] bf16: zero except dX[row[b]] = bf16(d[b]) (row[b] inside segment b of SEG rows)"""
    B, W = d.shape
    assert d.dtype == torch.float32 and d.is_contiguous() and row.dtype == torch.int32
    dX = torch.empty(M, W, device=d.device, dtype=torch.bfloat16)
    check(lib().sc_rows_scatter_bf16(_p(d), _p(row), _p(dX), M, B, W, SEG, _stream()), "sc_rows_scatter_bf16")
    return dX


# ---- CIF on the rows of the attention block in front of it (bf16, the block's pitch; csrc/cif.hip) ------------------------------------
def cif_fwd_rows(full: torch.Tensor, head: int, S: int, alpha: torch.Tensor, csum: torch.Tensor, T: int, thr: float) -> torch.Tensor:
    """full [B, P, C] bf16 contiguous, frames of utterance b = its rows head .. head + S - 1 -> out [B, T + 1, C] fp32"""
    B, P, C = full.shape
    assert full.dtype == torch.bfloat16 and full.is_contiguous() and head + S <= P
    assert alpha.dtype == torch.float32 and alpha.is_contiguous() and csum.is_contiguous() and tuple(alpha.shape) == (B, S)
    out = torch.empty(B, T + 1, C, device=full.device, dtype=torch.float32)
    x0 = full.view(-1)[head * C:]
    check(lib().sc_cif_fwd_rows(_p(x0), 1, P * C, _p(alpha), _p(csum), _p(out), B, S, C, T, float(thr), _stream()), "sc_cif_fwd_rows")
    return out


def cif_bwd_rows(full: torch.Tensor, head: int, S: int, alpha: torch.Tensor, csum: torch.Tensor, g: torch.Tensor, T: int, thr: float,
                 tail_rows: int = 0):
    """-> (d full [B, P, C] bf16: the frames' gradient, zero in every other row, pa, pb [nblk, B, S]); ``tail_rows``: d full is the
    head of a flat buffer with that many more (zero) rows behind it"""
    B, P, C = full.shape
    assert g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (B, T + 1, C)
    nblk = (C + 255) // 256
    if tail_rows:
        flat = torch.empty(B * P + tail_rows, C, device=full.device, dtype=full.dtype)
        flat[B * P:].zero_()
        dfull = flat[: B * P].view(B, P, C)
    else:
        dfull = torch.empty_like(full)
    pa = torch.empty(nblk, B, S, device=full.device, dtype=torch.float32)
    pb = torch.empty(nblk, B, S, device=full.device, dtype=torch.float32)
    check(lib().sc_cif_bwd_rows(_p(full.view(-1)[head * C:]), 1, P * C, _p(alpha), _p(csum), _p(g), _p(dfull.view(-1)[head * C:]), 1, P * C, head,
                                P - head - S, _p(pa), _p(pb), B, S, C, T, float(thr), _stream()), "sc_cif_bwd_rows")
    return dfull, pa, pb


def rows_zero_pad(flat: torch.Tensor, lead: int, B: int, P: int, head: int, stop: int, trail: int) -> None:
    """flat [lead + B P + trail, D] bf16: zero the rows that are not frames (see sc_rows_zero_pad_bf16)"""
    assert flat.dtype == torch.bfloat16 and flat.is_contiguous() and flat.shape[0] == lead + B * P + trail
    check(lib().sc_rows_zero_pad_bf16(_p(flat), lead, B, P, head, stop, trail, flat.shape[1], _stream()), "sc_rows_zero_pad_bf16")


def wsum_share(dX: torch.Tensor, w: torch.Tensor, prev: Optional[torch.Tensor], B: int, R: int, T: int, row_off: int = 1) -> torch.Tensor:
    """[B R, D] bf16 = (prev or 0) + bf16(w * dX[b, t + row_off]) for the frames t < T (sc_wsum_share_bf16): dX fp32 [B, R, D], w a
    one-element fp32 device tensor (one softmax weight of the weighted sum)"""
    D = dX.shape[-1]
    assert dX.dtype == torch.float32 and dX.is_contiguous() and w.dtype == torch.float32 and w.numel() == 1
    assert prev is None or (prev.dtype == torch.bfloat16 and prev.is_contiguous() and prev.numel() == B * R * D)
    out = torch.empty(B * R, D, device=dX.device, dtype=torch.bfloat16)
    check(lib().sc_wsum_share_bf16(_p(dX), _p(w), _p(prev), _p(out), B, R, T, D, row_off, _stream()), "sc_wsum_share_bf16")
    return out


def transpose_batched_bf16(x0: torch.Tensor, sx: int, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """n equally shaped [rows, cols] bf16 matrices, matrix z = ``sx`` elements after matrix z - 1 (``x0`` = the first, dense rows) ->
    out [n, cols, rows] (one launch, sc_transpose_batched_bf16)"""
    rows, cols = x0.shape
    assert x0.dtype == torch.bfloat16 and x0.stride(1) == 1
    if out is None:
        out = torch.empty(n, cols, rows, device=x0.device, dtype=torch.bfloat16)
    check(lib().sc_transpose_batched_bf16(_p(x0), x0.stride(0), sx, _p(out), rows, cols * rows, rows, cols, n, _stream()), "sc_transpose_batched_bf16")
    return out
