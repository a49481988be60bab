// 128 x {256,192} x 64 bf16 MFMA GEMM for gfx950, TWO workgroups per CU: the same wave block as gemm256_bf16.hip (one wave =
// 128 x 64 or 128 x 48 outputs = 8 x 4 (8 x 3) fragments of v_mfma_f32_16x16x32_bf16, same K order, same epilogue text, so results
// are bit-identical), but a workgroup is ONE 4-wave group with its own LDS ring (64 KiB) instead of two groups sharing the weight
// tile.  Two such workgroups are resident per CU (256 VGPRs each, one wave per SIMD each) and are NOT synchronised with each other:
// they drift into complementary phases, so one workgroup's epilogue (GELU on the VALU, the 64 / 128 KiB store burst) and its
// operand waits run beside the other workgroup's MFMA segments on the same SIMD, and the store bursts of the 512 resident
// workgroups are spread over time instead of arriving from every CU at once (gemm256: all 256 persistent workgroups reach
// their epilogues together).  Price: the weight tile is staged once per 128 rows instead of once per 256 (LDS-DMA bytes per MFMA
// x 1.5).
//
// Not persistent: one workgroup per tile; the hardware dispatches the next tile's workgroup when one retires, its prologue
// (first-load latency) hides under the co-resident workgroup's K loop.
//
// LDS: A ring of 2 x 16 KiB (128 rows x 64 k), ONE weight buffer of BN rows x 64 k (32 / 24 KiB): a K-tile's weight fragments are
// all read into registers first (8 x ds_read_b128), after which the next K-tile's weights are DMA'd over them.
//
// K-tile kt (A buffer par = kt & 1):
//   L0  read B fragments (all) + A fragments of rows 0-63 ; lgkmcnt(0) ; barrier 1   (the weight buffer is free)
//       DMA B(kt+1) -> weight buffer ; DMA A(kt+1) -> A[par ^ 1]
//   C0  32 MFMA (rows 0-63)
//   L1  read A fragments of rows 64-127
//   C1  32 MFMA (rows 64-127)
//       vmcnt(0) ; barrier 2          (K-tile kt+1 landed for every wave; every wave is done reading A[par])
// Hazards: A[par ^ 1] was last read in K-tile kt-1, before its barrier 2; B(kt) reads retire (lgkmcnt) before barrier 1.
#include <algorithm>

#include "sc_common.h"

namespace {

constexpr int BK = 64, ROWB = 128;
constexpr int A_BYTES = 128 * ROWB;              // 16 KiB: one A K-tile
constexpr int B_OFF = 2 * A_BYTES;               // weight buffer behind the A ring
// names the shared epilogue text expects: the wave's private 8 KiB staging region = smem + wave * 8 KiB (the A ring, free after
// the K loop)
constexpr int BUF_BYTES = 0, EPI_BYTES = 8192;

#define SC_BAR()                               \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DIAG 3 = no epilogue (timing only, results wrong).
template <int DIAG, int BN, int ACT, int DROP, int RES>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const sc_gemm_args p) {
    constexpr int TN = BN / 4, FN = TN / 16;
    constexpr int NBP = BN / 32;                             // weight DMA pieces (1 KiB = 8 rows) per wave and K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int wm = 0;
    const int wn = wave;

    const int nM = (p.M + 127) >> 7, nN = (p.N + BN - 1) / BN;
    const int n_tiles = nM * nN;
    // ---- block -> tile (XCD-aware, banded; see gemm_bf16.hip): 16 M-tiles x all N-tiles per band ---------------
    int cm0, cn0;
    {
        const int vb = blockIdx.x;
        const int xcd = vb & 7, q = n_tiles >> 3, r = n_tiles & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        constexpr int GM = 16;
        const int band = L / (GM * nN), first_m = band * GM;
        const int gm = min(GM, nM - first_m);
        const int within = L - band * GM * nN;
        cm0 = (first_m + within % gm) << 7;
        cn0 = (within / gm) * BN;
    }

    const int z = blockIdx.z, z1 = z / p.nb2, z2 = z % p.nb2;
    const uint16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const uint16_t* W = p.W + z1 * p.sW1 + z2 * p.sW2;
    const float* bias = p.bias ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    const uint16_t* Rs = (RES && p.residual) ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    const int64_t coff = z1 * p.sC1 + z2 * p.sC2;

    // ---- DMA sources as 32-bit element offsets from the (uniform) operand bases: piece i of a wave = rows 8 (4 i + wave) .. + 7 -----
    uint32_t a_off[4], b_off[NBP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        a_off[i] = (uint32_t)min(cm0 + row, p.M - 1) * (uint32_t)p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        b_off[i] = (uint32_t)min(cn0 + row, p.N - 1) * (uint32_t)p.ldw + c * 8;
    }
    auto dma_A = [&](int par, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(A + (a_off[i] + k0), smem + par * A_BYTES + (i * 4 + wave) * 1024);
    };
    auto dma_B = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NBP; ++i) glds16(W + (b_off[i] + k0), smem + B_OFF + (i * 4 + wave) * 1024);
    };

    // ---- fragment read offsets: row = 16 f + (lane & 15)  =>  the swizzle term depends on the lane only -----------
    const int sw = (lane >> 1) & 7;
    const int frag_off0 = (lane & 15) * ROWB + (((lane >> 4)) ^ sw) * 16;          // kk = 0
    const int frag_off1 = (lane & 15) * ROWB + ((4 + (lane >> 4)) ^ sw) * 16;      // kk = 1
    const int b_base = B_OFF + wn * TN * ROWB;
    const int nk = p.K / BK;
    const int tap_c = p.tap_c;                   // conv-shaped A: same K-tile visiting order as gemm256 (bit-identical sums)
    auto koff = [&](int kt) -> int {
        if (tap_c == 0) return kt * BK;
        const int c = kt / 3, j = kt - 3 * c;
        return (j == 0 ? 0 : (3 - j) * tap_c) + c * BK;
    };
    const uint32_t drop_thr = DROP ? (uint32_t)(p.drop_p * 65536.f + 0.5f) : 0u;
    const float drop_scale = DROP ? 1.f / (1.f - p.drop_p) : 1.f;

#define SC_MFMA_HALF(MS)                                                                                  \
    do {                                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
                _Pragma("unroll") for (int ni = 0; ni < FN; ++ni)                                         \
                    acc[(MS) * 4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                     \
                        bf[ni][kk], af[mi][kk], acc[(MS) * 4 + mi][ni], 0, 0, 0);                         \
    } while (0)

    dma_B(0);
    dma_A(0, 0);

    // accumulators start from the bias; transposed fragments (weight operand first): a lane owns one output row and four
    // consecutive columns (see gemm256_bf16.hip)
    f32x4 acc[8][FN];
    {
        const int bl = lane >> 4;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = cn0 + wn * TN + ni * 16 + 4 * bl;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias && n + 4 <= p.N) bv = *(const f32x4*)(bias + n);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = bv;
        }
    }
    bf16x8 af[4][2], bf[FN][2];

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SC_BAR();

    int kt = 0;
    do {
        const int par = kt & 1;
        const char* as = smem + par * A_BYTES;
        const char* bs = smem + b_base;
        // ---------------- L0
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            bf[ni][0] = *(const bf16x8*)(bs + ni * 16 * ROWB + frag_off0);
            bf[ni][1] = *(const bf16x8*)(bs + ni * 16 * ROWB + frag_off1);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(as + mi * 16 * ROWB + frag_off0);
            af[mi][1] = *(const bf16x8*)(as + mi * 16 * ROWB + frag_off1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SC_BAR();
        if (kt + 1 < nk) {
            const int k1 = koff(kt + 1);
            dma_B(k1);
            dma_A(par ^ 1, k1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- C0
        SC_MFMA_HALF(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- L1
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(as + (4 + mi) * 16 * ROWB + frag_off0);
            af[mi][1] = *(const bf16x8*)(as + (4 + mi) * 16 * ROWB + frag_off1);
        }
        // ---------------- C1
        SC_MFMA_HALF(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SC_BAR();
    } while (++kt < nk);

    if (DIAG == 3) {
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) t += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (t == 123.456f) ((float*)p.C)[tid] = t;
    } else {
        constexpr int last_par = 0;
#include "gemm_epilogue.inc"
    }
}

}  // namespace

template <int DIAG, int BN, int ACT, int DROP, int RES>
static int launch128__(const sc_gemm_args& a, hipStream_t s) {
    constexpr int LDS = B_OFF + BN * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm128_kernel<DIAG, BN, ACT, DROP, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            sc_set_error("hipFuncSetAttribute(gemm128): %s", hipGetErrorString(e));
            return -3;
        }
        attr_set = true;
    }
    const int nM = (a.M + 127) / 128, nN = (a.N + BN - 1) / BN;
    dim3 grid(nM * nN, 1, a.nb1 * a.nb2);
    hipLaunchKernelGGL((gemm128_kernel<DIAG, BN, ACT, DROP, RES>), grid, dim3(256), LDS, s, a);
    SC_LAUNCH_CHECK();
    return 0;
}

template <int DIAG, int BN, int ACT, int DROP>
static int launch128_(const sc_gemm_args& a, hipStream_t s) {
    if constexpr (DIAG == 0) {
        if (a.residual) return launch128__<DIAG, BN, ACT, DROP, 1>(a, s);
    }
    return launch128__<DIAG, BN, ACT, DROP, 0>(a, s);
}

template <int DIAG, int BN>
static int launch128(const sc_gemm_args& a, hipStream_t s) {
    if constexpr (DIAG == 0) {
        if (a.drop_p > 0.f) return a.act == 1 ? launch128_<DIAG, BN, 1, 1>(a, s) : launch128_<DIAG, BN, 0, 1>(a, s);
    }
    return a.act == 1 ? launch128_<DIAG, BN, 1, 0>(a, s) : launch128_<DIAG, BN, 0, 0>(a, s);
}

// operand extents the 32-bit DMA offsets can address
bool sc_gemm128_fits(const sc_gemm_args& a) {
    return (int64_t)a.M * a.lda + a.K < ((int64_t)1 << 31) && (int64_t)a.N * a.ldw + a.K < ((int64_t)1 << 31);
}

int sc_gemm128_launch(const sc_gemm_args& a_in, hipStream_t s) {
    sc_gemm_args a = a_in;
    a.reserved = (a_in.reserved == 1 || (a_in.reserved == 0 && a_in.residual != nullptr)) ? 1 : 0;
    if (sc_option(1)) a.reserved = 0;
    if (a.tile == 33) return launch128<3, 256>(a, s);        // diagnostics only
    if (a.tile == 11) return launch128<0, 192>(a, s);
    if (a.tile == 10) return launch128<0, 256>(a, s);
    const bool ok192 = (a.n_split < 0 || a.n_split % 192 == 0);
    const bool ok256 = (a.n_split < 0 || a.n_split % 256 == 0);
    auto cost = [&](int BN) {
        const double tiles = (double)((a.M + 127) / 128) * ((a.N + BN - 1) / BN) * a.nb1 * a.nb2;
        const int slots = 2 * sc_num_cus();
        return std::max(tiles / slots, 1.0) * BN * (BN == 192 ? 1.12 : 1.0);
    };
    if (ok192 && (!ok256 || cost(192) < cost(256))) return launch128<0, 192>(a, s);
    return launch128<0, 256>(a, s);
}
