// bf16 MFMA GEMM  C = epi(A . W^T)  for gfx950.
//
// Tile BM x BN x 64, 256 threads = 4 waves (2 x 2), v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// Staging: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 tile rows of 128 B), two LDS
// buffers; the next K-tile's DMA is issued before the current tile's MFMAs, one barrier per K-tile.
// LDS image: row-major 128-B rows, 16-B chunk index XOR ((row >> 1) & 7): with two rows per 256-B bank row
// every ds_read_b128 lane group of the 16x16x32 fragment read touches 16 distinct 16-B slots (conflict
// free); the DMA writes LDS linearly, so the swizzle is applied on the per-lane GLOBAL source address.
// Epilogue: fp32 accumulators (+bias, GELU) -> LDS tile -> coalesced 16-B rows (+residual) -> HBM;
// column blocks >= n_split are written transposed per head (V^T for the attention kernel).
// Block -> tile map: XCD-aware (blocks b, b+8, ... share an XCD and therefore an L2) and banded
// (8 M-tiles x all N-tiles per band) so that co-resident blocks share A rows and W rows in L2.
#include "sc_common.h"

namespace {

constexpr int BK = 64;
constexpr int ROWB = BK * 2;   // bytes per LDS tile row

// NS = LDS stages.  2: the next K-tile is staged while the current one is multiplied, one __syncthreads (which drains the DMA) per
// K-tile - two workgroups per CU cover each other's waits.  4 (small problems, at most one workgroup per CU anyway: the text tower's
// 2048 packed rows): K-tiles are staged THREE ahead, raw barrier + one counted s_waitcnt vmcnt per K-tile: 23.3 -> 18.2 us on the
// K = 2048 products (tools/bench_small_gemm.py).  Six stages measured the same: past the round-trip latency a 128 x 64 tile is bound by
// the LDS-DMA rate of its CU (24 KiB per K-tile at ~21 B/clk, the same per-CU rate the 256-row kernel stages at).
template <int BM, int BN, int NS = 2>
struct GemmCfg {
    static constexpr int WM = 2, WN = 2;
    static constexpr int TM = BM / WM, TN = BN / WN;
    static constexpr int FM = TM / 16, FN = TN / 16;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = NS * (A_BYTES + B_BYTES);
    static constexpr int EPI_BYTES = BM * BN * 4;
    static constexpr int RING_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    static constexpr int LDS_BYTES = RING_BYTES + BM * 8;       // + (mean, rstd) of the tile's rows: the LayerNorm-prologue mode
};

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int NS = 2>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const sc_gemm_args p) {
    using Cfg = GemmCfg<BM, BN, NS>;
    constexpr int FM = Cfg::FM, FN = Cfg::FN, TM = Cfg::TM, TN = Cfg::TN;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> (m_tile, n_tile): XCD remap (bijective) then banded order --------------------------
    const int nM = (p.M + BM - 1) / BM, nN = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GM = 8;
    const int band = L / (GM * nN);
    const int first_m = band * GM;
    const int gm = min(GM, nM - first_m);
    const int within = L - band * GM * nN;
    const int n_tile = within / gm, m_tile = first_m + within % gm;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    // ---- batch offsets ----------------------------------------------------------------------------------
    const int z = blockIdx.z, z1 = z / p.nb2, z2 = z % p.nb2;
    const uint16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const uint16_t* W = p.W + z1 * p.sW1 + z2 * p.sW2;
    const float* bias = p.bias ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    const uint16_t* Rs = p.residual ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    const int64_t coff = z1 * p.sC1 + z2 * p.sC2;

    // ---- DMA source pointers (per lane; swizzle lives in the source address) ----------------------------
    constexpr int A_INST = BM / 32, B_INST = BN / 32;   // wave-instructions per wave per K-tile
    const uint16_t* a_src[A_INST];
    const uint16_t* b_src[B_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int gm_row = min(m0 + row, p.M - 1);
        a_src[i] = A + (int64_t)gm_row * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int gn_row = min(n0 + row, p.N - 1);
        b_src[i] = W + (int64_t)gn_row * p.ldw + c * 8;
    }
    char* const As = smem;
    char* const Bs = smem + NS * Cfg::A_BYTES;

    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
            glds16(a_src[i] + k0, As + buf * Cfg::A_BYTES + (i * 4 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < B_INST; ++i)
            glds16(b_src[i] + k0, Bs + buf * Cfg::B_BYTES + (i * 4 + wave) * 1024);
    };

    // ---- fragment read offsets (bytes inside a tile) ----------------------------------------------------
    int a_off[FM][2], b_off[FN][2];
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const int row = wm * TM + mi * 16 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[mi][kk] = row * ROWB + (((kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) {
        const int row = wn * TN + ni * 16 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[ni][kk] = row * ROWB + (((kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7)) << 4);
    }

    const int nk = p.K / BK;
    // K-tile order of conv-shaped problems (sc_gemm_args.tap_c, see gemm256_bf16.hip): the same order in every kernel, so the
    // result does not depend on which tile family the dispatcher picks for a given row count
    const int tap_c = p.tap_c;
    auto koff = [&](int kt) -> int {
        if (tap_c == 0) return kt * BK;
        const int c = kt / 3, j = kt - 3 * c;
        return (j == 0 ? 0 : (3 - j) * tap_c) + c * BK;
    };
    // the first NS - 1 K-tiles go into flight here, ahead of the optional statistics prologue
    if constexpr (NS == 2) {
        stage(0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < NS - 1; ++j)
            if (j < nk) stage(j, koff(j));
    }

    // ---- LayerNorm in the PROLOGUE (round 6; the text tower's LN -> QKV and LN -> fc1 pairs: K = the model width, so a tile holds whole
    // rows of A).  ln_colsum given WITHOUT ln_stats: A holds RAW rows, W = W0 diag(gamma) folded by the caller, and
    //     C = rstd_m (A W^T - mean_m ln_colsum) + bias,   ln_colsum[n] = sum_k W[n, k],  bias[n] = sum_k beta[k] W0[n, k] + bias0[n]
    // (the LN = 1 consumer of gemm256_bf16.hip), with mean_m / rstd_m computed HERE: every workgroup reads its BM rows once more (two
    // passes in registers, half a wave per row as layernorm16_kernel; they come out of the L2 - the DMA of K-tile 0 is already in
    // flight) instead of a LayerNorm launch writing and this kernel re-reading a normalised copy.
    const bool ln_self = p.ln_colsum != nullptr && p.ln_stats == nullptr;
    float* const row_stat = (float*)(smem + Cfg::RING_BYTES);          // [BM][2]
    if (ln_self) {
        const int l = tid & 31, hw = tid >> 5;
        const int nch = p.K >> 3;                                        // 16-byte chunks per row (K <= 1024: at most 4 per lane)
        for (int r = hw; r < BM; r += 8) {
            const uint16_t* xr = A + (int64_t)min(m0 + r, p.M - 1) * p.lda;
            float v[4][8];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = l + i * 32;
                if (ch < nch) {
                    const uint4 u = *(const uint4*)(xr + ch * 8);
                    v[i][0] = bflo(u.x); v[i][1] = bfhi(u.x); v[i][2] = bflo(u.y); v[i][3] = bfhi(u.y);
                    v[i][4] = bflo(u.z); v[i][5] = bfhi(u.z); v[i][6] = bflo(u.w); v[i][7] = bfhi(u.w);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += v[i][j];
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)p.K;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (l + i * 32 < nch) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = v[i][j] - mean;
                        sq = fmaf(d, d, sq);
                    }
                }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if (l == 0) {
                row_stat[2 * r] = mean;
                row_stat[2 * r + 1] = rsqrtf(sq / (float)p.K + p.ln_eps);
            }
        }
        // (visible to the epilogue: the K loop below has at least one workgroup barrier)
    }
    // accumulators start from the bias (same summation order as gemm256_bf16.hip: the tile variants agree bitwise)
    f32x4 acc[FM][FN];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) {
        const int n = n0 + wn * TN + ni * 16 + (lane & 15);
        const float bv = (bias && n < p.N && !ln_self) ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) acc[mi][ni] = f32x4{bv, bv, bv, bv};
    }

    auto mma_tile = [&](int buf) {
        const char* as = As + buf * Cfg::A_BYTES;
        const char* bs = Bs + buf * Cfg::B_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int mi = 0; mi < FM; ++mi) af[mi] = *(const bf16x8*)(as + a_off[mi][kk]);
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) bfr[ni] = *(const bf16x8*)(bs + b_off[ni][kk]);
#pragma unroll
            for (int mi = 0; mi < FM; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    };
    if constexpr (NS == 2) {
        __syncthreads();   // hipcc drains the DMA (vmcnt(0)) in front of the barrier
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(buf ^ 1, koff(kt + 1));
            mma_tile(buf);
            __syncthreads();
        }
    } else {
        // K-tiles kt .. kt + NS - 2 in flight.  Per K-tile: wait until tile kt has landed (all but the NS - 2 newest stages of THIS
        // wave), barrier (every wave's share has landed, and every wave is done with tile kt - 1), re-stage tile kt - 1's buffer
        // with tile kt + NS - 1, multiply.
        constexpr int IPW = A_INST + B_INST;                  // LDS-DMA instructions per wave and stage
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < nk) stage((kt + NS - 1) % NS, koff(kt + NS - 1));
            mma_tile(kt % NS);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // the epilogue reuses the ring
        asm volatile("" ::: "memory");
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    float* Cs = (float*)smem;
    const bool transposed = (p.n_split >= 0) && (n0 >= p.n_split);
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) {
        const int nl = wn * TN + ni * 16 + (lane & 15);
        const int n = n0 + nl;
        const float bv = (bias && n < p.N) ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const int ml = wm * TM + mi * 16 + 4 * (lane >> 4);
            f32x4 v = acc[mi][ni];
            if (ln_self) {                                  // y = rstd_m (acc - mean_m s_n) + c_n
                const float sn = n < p.N ? p.ln_colsum[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x2 st = *(const f32x2*)(row_stat + 2 * (ml + r));
                    v[r] = fmaf(st.y, fmaf(-st.x, sn, v[r]), bv);
                }
            }
            if (p.act == 1 && p.aux_mode == 0) {
                const f32x2 g0 = gelu_bf2(f32x2{v[0], v[1]}), g1 = gelu_bf2(f32x2{v[2], v[3]});
                v = f32x4{g0.x, g0.y, g1.x, g1.y};
            }
            if (transposed) {
                *(f32x4*)(Cs + nl * BM + ml) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(ml + r) * BN + nl] = v[r];
            }
        }
    }
    __syncthreads();
    if (!transposed) {
        constexpr int CPR = BN / 8;   // 8-element chunks per row
#pragma unroll
        for (int it = 0; it < BM * CPR / 256; ++it) {
            const int q = it * 256 + tid;
            const int row = q / CPR, cc = q % CPR;
            const int m = m0 + row, n = n0 + cc * 8;
            if (m < p.M && n + 8 <= p.N) {
                const f32x4 lo = *(const f32x4*)(Cs + row * BN + cc * 8);
                const f32x4 hi = *(const f32x4*)(Cs + row * BN + cc * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (p.drop_p > 0.f) {                    // train-mode dropout before the residual add (same mask as gemm256_bf16.hip)
                    const uint32_t thr = (uint32_t)(p.drop_p * 65536.f + 0.5f);
                    const float sc = 1.f / (1.f - p.drop_p);
                    const uint32_t keep = sc_keep8((uint32_t)m * (uint32_t)p.N + (uint32_t)n, p.drop_seed, thr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (keep >> e) & 1u ? v[e] * sc : 0.f;
                }
                if (Rs) {
                    const uint4 rv = *(const uint4*)(Rs + (int64_t)m * p.ldr + n);
                    v[0] += bflo(rv.x); v[1] += bfhi(rv.x); v[2] += bflo(rv.y); v[3] += bfhi(rv.y);
                    v[4] += bflo(rv.z); v[5] += bfhi(rv.z); v[6] += bflo(rv.w); v[7] += bfhi(rv.w);
                }
                if (p.aux_mode) {        // activation fused with the aux operand Ct (see sc_gemm_args.aux_mode): bit-identical to GEMM + sc_act_bf16
                    uint16_t* X = p.Ct + coff + (int64_t)m * p.ldc + n;
                    uint4 r;             // the values the plain GEMM would have stored
                    r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]); r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
                    const float rr[8] = {bflo(r.x), bfhi(r.x), bflo(r.y), bfhi(r.y), bflo(r.z), bfhi(r.z), bflo(r.w), bfhi(r.w)};
                    if (p.aux_mode == 1) {
                        *(uint4*)X = r;                                              // u
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = act_fwd(rr[e], p.act);   // act(u)
                    } else {
                        const uint4 uu = *(const uint4*)X;                           // saved u
                        const float u8[8] = {bflo(uu.x), bfhi(uu.x), bflo(uu.y), bfhi(uu.y), bflo(uu.z), bfhi(uu.z), bflo(uu.w), bfhi(uu.w)};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = rr[e] * act_grad(u8[e], p.act);
                    }
                }
                if (p.out_f32) {
                    float* C = (float*)p.C + coff + (int64_t)m * p.ldc + n;
                    *(f32x4*)C = f32x4{v[0], v[1], v[2], v[3]};
                    *(f32x4*)(C + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    uint16_t* C = (uint16_t*)p.C + coff + (int64_t)m * p.ldc + n;
                    uint4 o;
                    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
                    o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
                    *(uint4*)C = o;
                }
            }
        }
    } else {
        constexpr int CPR = BM / 8;
        const int H = (p.N - p.n_split) / p.dh;
#pragma unroll
        for (int it = 0; it < BN * CPR / 256; ++it) {
            const int q = it * 256 + tid;
            const int nrow = q / CPR, mc = q % CPR;
            const int m = m0 + mc * 8, n = n0 + nrow;
            if (m < p.M && n < p.N) {
                const f32x4 lo = *(const f32x4*)(Cs + nrow * BM + mc * 8);
                const f32x4 hi = *(const f32x4*)(Cs + nrow * BM + mc * 8 + 4);
                const int nn = n - p.n_split, hh = nn / p.dh, d = nn % p.dh;
                uint16_t* dst;
                if (p.seg_chunk) {       // ragged rows: (first row, pitch) of the utterance that owns this 8-row chunk
                    const int2 sg = *(const int2*)(p.seg_chunk + 4 * (m >> 3));
                    dst = p.Ct + (int64_t)(p.N - p.n_split) * sg.x + (int64_t)nn * sg.y + (m - sg.x);
                } else {
                    const int b = m / p.R, t = m % p.R;
                    dst = p.Ct + (((int64_t)b * H + hh) * p.dh + d) * p.R + t;
                }
                uint4 o;
                o.x = pack2bf(lo[0], lo[1]); o.y = pack2bf(lo[2], lo[3]);
                o.z = pack2bf(hi[0], hi[1]); o.w = pack2bf(hi[2], hi[3]);
                *(uint4*)dst = o;
            }
        }
    }
}

template <int BM, int BN, int NS = 2>
int launch(const sc_gemm_args& a, hipStream_t s) {
    using Cfg = GemmCfg<BM, BN, NS>;
    static sc_lds_attr_once attr;
    if (hipError_t e = sc_set_max_lds_once(attr, gemm_bf16_kernel<BM, BN, NS>, Cfg::LDS_BYTES); e != hipSuccess) {
        sc_set_error("hipFuncSetAttribute(gemm %dx%d): %s", BM, BN, hipGetErrorString(e));
        return -3;
    }
    const int nM = (a.M + BM - 1) / BM, nN = (a.N + BN - 1) / BN;
    dim3 grid(nM * nN, 1, a.nb1 * a.nb2);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, NS>), grid, dim3(256), Cfg::LDS_BYTES, s, a);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int sc_gemm256_launch(const sc_gemm_args& a, hipStream_t s);    // gemm256_bf16.hip
int sc_gemm256_bn(const sc_gemm_args& a);

extern "C" int32_t sc_gemm_stats_strips(const sc_gemm_args* a) {
    if (!a || a->N <= 0) return 0;
    const int bn = sc_gemm256_bn(*a);
    return (a->N + bn - 1) / bn;
}

extern "C" int sc_gemm_bf16(const sc_gemm_args* args, void* stream) {
    SC_CHECK(args != nullptr, "sc_gemm_bf16: null args");
    sc_gemm_args a = *args;
    SC_CHECK(a.A && a.W && a.C, "sc_gemm_bf16: null operand");
    SC_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "sc_gemm_bf16: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SC_CHECK(a.K % BK == 0, "sc_gemm_bf16: K=%d must be a multiple of %d", a.K, BK);
    SC_CHECK(a.N % 8 == 0, "sc_gemm_bf16: N=%d must be a multiple of 8", a.N);
    SC_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0, "sc_gemm_bf16: leading dims must be multiples of 8");
    SC_CHECK(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0 && ((uintptr_t)a.C % 16) == 0,
             "sc_gemm_bf16: operands must be 16-byte aligned");
    SC_CHECK(a.act == 0 || a.act == 1 || (a.act == 2 && a.aux_mode != 0), "sc_gemm_bf16: act=%d", a.act);
    SC_CHECK(a.drop_p >= 0.f && a.drop_p < 1.f, "sc_gemm_bf16: drop_p=%f", (double)a.drop_p);
    SC_CHECK(a.tap_c == 0 || (a.tap_c > 0 && a.tap_c % 64 == 0 && a.K == 3 * a.tap_c),
             "sc_gemm_bf16: tap_c=%d needs tap_c %% 64 == 0 and K == 3 * tap_c (K=%d)", a.tap_c, a.K);
    SC_CHECK(a.drop_p == 0.f || ((int64_t)a.M * a.N < (int64_t)1 << 32 && a.nb1 * a.nb2 == 1),
             "sc_gemm_bf16: dropout needs M*N < 2^32 and no batch");
    if (a.nb1 < 1) a.nb1 = 1;
    if (a.nb2 < 1) a.nb2 = 1;
    if (a.residual) SC_CHECK(a.ldr % 8 == 0 && ((uintptr_t)a.residual % 16) == 0, "sc_gemm_bf16: residual alignment");
    // the epilogues read the bias as 16-byte f32x4 groups (gemm_epilogue.inc): base and batch strides must keep that alignment
    if (a.bias) SC_CHECK(((uintptr_t)a.bias % 16) == 0 && a.sBias1 % 4 == 0 && a.sBias2 % 4 == 0,
                         "sc_gemm_bf16: bias must be 16-byte aligned with batch strides that are multiples of 4 floats");
    if (a.n_split >= 0) {
        SC_CHECK(a.Ct != nullptr && a.dh > 0 && (a.R > 0 || a.seg_chunk), "sc_gemm_bf16: transposed store needs Ct, dh, R (or seg_chunk)");
        if (a.seg_chunk)
            SC_CHECK(a.M % SC_SEG_ROWS == 0 && ((uintptr_t)a.seg_chunk % 16) == 0 && a.nb1 * a.nb2 == 1,
                     "sc_gemm_bf16: a segment table needs M %% 8 == 0, a 16-byte aligned table and no batch");
        else
            SC_CHECK(a.R % 8 == 0 && a.M % a.R == 0, "sc_gemm_bf16: transposed store needs R %% 8 == 0, M %% R == 0");
        SC_CHECK(a.n_split % 128 == 0 && (a.N - a.n_split) % a.dh == 0, "sc_gemm_bf16: transposed store needs n_split %% 128 == 0, (N - n_split) %% dh == 0");
        SC_CHECK(a.out_f32 == 0, "sc_gemm_bf16: transposed store is bf16 only");
    } else {
        a.n_split = -1;
    }
    if (a.tn) {
        SC_CHECK(a.M % 256 == 0 && a.N % 256 == 0 && !a.bias && !a.residual && a.act == 0 && a.drop_p == 0.f && a.n_split < 0 &&
                 !a.ln_stats && !a.stats_out && !a.res_stats && a.tap_c == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0,
                 "sc_gemm_bf16: the TN form needs M, N %% 256 == 0 and a plain epilogue (M=%d N=%d)", a.M, a.N);
        SC_CHECK(a.tile == 0 || a.tile == 2 || a.tile == 8, "sc_gemm_bf16: the TN form is built on the 256 x 256 tile only");
        SC_CHECK(a.k_total == 0 || (a.k_total % 64 == 0 && a.nb2 == 1 && (int64_t)(a.nb1 - 1) * a.K < a.k_total && (int64_t)a.nb1 * a.K >= a.k_total),
                 "sc_gemm_bf16: ragged TN slices need k_total %% 64 == 0 and (nb1 - 1) K < k_total <= nb1 K");
        return sc_gemm256_launch(a, (hipStream_t)stream);
    }
    if (a.aux_mode) {
        SC_CHECK((a.aux_mode == 1 || a.aux_mode == 2) && (a.act == 1 || a.act == 2) && a.Ct && a.n_split < 0 && !a.out_f32 && !a.tn &&
                 a.drop_p == 0.f && !a.ln_stats && !a.stats_out && !a.res_stats && ((uintptr_t)a.Ct % 16) == 0,
                 "sc_gemm_bf16: aux_mode needs act 1 / 2, the aux pointer in Ct (16-byte aligned), bf16 output and a plain epilogue");
        // 128-row tiles: both activations; 256-row tiles (round 4): erf-GELU, plain epilogue (no residual)
        SC_CHECK(a.tile == 0 || a.tile == 1 || a.tile == 3 || a.tile == 13 || a.tile == 14 || a.tile == 15 || ((a.tile == 2 || a.tile == 7 || a.tile == 8) && a.act == 1 && !a.residual),
                 "sc_gemm_bf16: aux_mode on the 256-row tiles needs act = 1 (erf-GELU) and no residual");
    } else {
        SC_CHECK(a.act == 0 || a.act == 1, "sc_gemm_bf16: act=%d (2 = QuickGELU needs aux_mode)", a.act);
    }
    const bool ln = a.ln_stats || a.stats_out || a.res_stats;
    const bool ln_self = a.ln_colsum && !a.ln_stats;            // LayerNorm in the prologue of the 128- / 64-row tiles (gemm_bf16_kernel)
    if (ln_self) {
        SC_CHECK(!a.stats_out && !a.res_stats && a.K <= 1024 && a.ln_eps > 0.f && a.tap_c == 0 && !a.tn && a.n_split < 0 && a.lda >= a.K &&
                     ((uintptr_t)a.ln_colsum % 4) == 0,
                 "sc_gemm_bf16: LayerNorm prologue (ln_colsum without ln_stats) needs K <= 1024, ln_eps, plain rows and no transposed store (K=%d)", a.K);
        SC_CHECK(a.tile == 0 || a.tile == 1 || a.tile == 3 || a.tile == 13 || a.tile == 14 || a.tile == 15,
                 "sc_gemm_bf16: the LayerNorm prologue is built into the 128- and 64-row tiles (tile=%d)", a.tile);
    }
    if (ln) {
        // LayerNorm folded into the GEMMs: built into the 256-row tile family only (csrc/gemm256_bf16.hip)
        SC_CHECK(a.tile == 0 || a.tile == 2 || a.tile == 7 || a.tile == 8, "sc_gemm_bf16: LayerNorm folding needs the 256-row tile family");
        SC_CHECK(a.nb1 * a.nb2 == 1 && a.out_f32 == 0, "sc_gemm_bf16: LayerNorm folding: no batch, bf16 output");
        SC_CHECK(a.ln_eps > 0.f, "sc_gemm_bf16: LayerNorm folding needs ln_eps");
        if (a.ln_stats) {
            SC_CHECK(a.ln_colsum && a.ln_ns >= 1 && a.ln_ns <= 4 && !a.residual && a.drop_p == 0.f && !a.stats_out && !a.res_stats,
                     "sc_gemm_bf16: a LayerNorm-folded consumer takes ln_stats + ln_colsum (1..4 strips), no residual, no dropout");
            SC_CHECK(((uintptr_t)a.ln_colsum % 16) == 0 && ((uintptr_t)a.ln_stats % 16) == 0, "sc_gemm_bf16: ln_colsum / ln_stats alignment");
        } else {
            SC_CHECK(a.stats_out && a.residual && a.act == 0, "sc_gemm_bf16: a statistics producer needs stats_out, a residual and act = 0");
            SC_CHECK((a.N + sc_gemm256_bn(a) - 1) / sc_gemm256_bn(a) <= 4, "sc_gemm_bf16: more than 4 statistics strips (N = %d > 1024)", a.N);
            if (a.res_stats)
                SC_CHECK(a.res_gamma && a.res_beta && a.res_ns >= 1 && a.res_ns <= 4 && ((uintptr_t)a.res_gamma % 16) == 0 &&
                         ((uintptr_t)a.res_beta % 16) == 0, "sc_gemm_bf16: a raw residual needs res_gamma / res_beta (16-byte aligned) and 1..4 strips");
        }
    }
    hipStream_t s = (hipStream_t)stream;
    int tile = a.tile;
    if (tile == 0 && ln) tile = 2;
    if (tile == 0) {
        const int64_t tiles256 = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) * a.nb1 * a.nb2;
        if (a.N <= 64 && a.n_split < 0) tile = 3;            // narrow outputs (grouped pos_conv, N = 48)
        else if (!ln_self && (!a.aux_mode || (a.act == 1 && !a.residual)) && a.M >= 512 && a.N >= 192 && tiles256 >= 192 && (a.n_split < 0 || a.n_split % 64 == 0)) tile = 2;
        else {
            // small problems (the text tower's 2048 packed rows): 128 x 64 tiles give twice the workgroups, 10-20 % faster up to two
            // waves of 128 x 128 tiles per CU (tools/bench_small_gemm.py)
            const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.nb1 * a.nb2;
            const int64_t tiles64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64) * a.nb1 * a.nb2;
            // long-K products with a narrow output (text tower: fc2, the input gradients of fc1 / QKV): 64 x 64 tiles put them on
            // 2-3 times the CUs, each streaming half the A rows: 18.4 -> 13.1 us (2048 x 512 x 2048), 20.2 -> 16.3 (2048 x 768 x 2304)
            // (the K loop runs at the rate the CU issues its LDS-DMA pieces, ~0.35 us per 16 KiB K-tile whatever the ring depth - an
            // 8-stage ring changed nothing -, so what helps is more CUs; round 4: also the K = 512 / 768 products with a narrow
            // output, 8.0 -> 6.6 us at 2048 x 512 x 512, 10.4 -> 9.0 at 2048 x 768 x 768)
            if (a.n_split < 0 && a.K >= 512 && tiles64 <= 2 * (int64_t)sc_num_cus()) tile = 15;
            else tile = (tiles128 <= 2 * (int64_t)sc_num_cus() && a.n_split < 0) ? 3 : 1;
        }
    }
    switch (tile) {
        case 1: return launch<128, 128>(a, s);
        case 2: case 32: case 34: case 7: case 8:   // 2 = 256-row tile, width (256 / 192) chosen by wave quantisation; 7 / 8 force it
            SC_CHECK(a.n_split < 0 || a.n_split % 192 == 0 || a.n_split % 256 == 0,
                     "sc_gemm_bf16: 256-row tiles need n_split %% 192 == 0 or %% 256 == 0");
            return sc_gemm256_launch(a, s);
        case 3: case 13: {
            SC_CHECK(a.n_split < 0, "sc_gemm_bf16: 128x64 tile has no transposed store");
            // at most one workgroup per CU anyway: the 4-stage ring hides the operand round trips the second workgroup would have
            const int64_t wgs = (int64_t)((a.M + 127) / 128) * ((a.N + 63) / 64) * a.nb1 * a.nb2;
            if (tile == 13 || (a.tile == 0 && wgs <= (int64_t)sc_num_cus() && a.K >= 1024)) return launch<128, 64, 4>(a, s);
            return launch<128, 64>(a, s);
        }
        case 14: case 15:                                      // 64 x 64 (2 / 4 stages): few-row problems with narrow outputs
            SC_CHECK(a.n_split < 0, "sc_gemm_bf16: 64x64 tile has no transposed store");
            return tile == 15 ? launch<64, 64, 4>(a, s) : launch<64, 64>(a, s);
        default: sc_set_error("sc_gemm_bf16: tile=%d", a.tile); return -1;
    }
}
