// 256 x {256,192} x 64 bf16 MFMA GEMM for gfx950: persistent, deep-pipelined variant of gemm_bf16.hip for the large GEMMs.
//
// 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns a 128 x 64 (or 128 x 48) output block = 8 x 4 (8 x 3) fragments
// of v_mfma_f32_16x16x32_bf16.  One workgroup per CU (128 KiB LDS), so two waves share each SIMD: waves 0-3 (wm = 0)
// and 4-7 (wm = 1) run STAGGERED by one barrier interval - while one group is in its MFMA segment the other issues
// LDS reads / LDS-DMA - which keeps the matrix pipe fed by one wave at a time (MI355X_MICROARCH "Two waves per SIMD").
//
// LDS: 2 K-tile buffers x {A, B} x 2 half-tiles of 128 rows x 64 k (16 KiB each), same XOR-swizzled 128-B row
// image as gemm_bf16.hip.  Staging is LDS-DMA (global_load_lds_dwordx4) that stays in flight across barriers:
// raw s_barrier (no implicit vmcnt(0)), one counted `s_waitcnt vmcnt(NBI)` per K-tile.
//
// Round 5: K-tile kt = TWO phases, each [L segment | barrier | C segment: 32 MFMA | barrier]  (in-kernel stamps had the four-phase loop
// at 2390 cycles per K-tile against the 2048 its MFMAs take: ~43 cycles around each of its 8 barriers):
//   X  L: read B(n-sub 0) 4 + A(m-sub 0) 8 + B(n-sub 1) 4 x b128 ; DMA A(kt+1) -> other buffer      C: (m0, n0), (m0, n1)
//   Y  L: read A(m-sub 1) 8 ; DMA B(kt+2) -> this buffer ; s_waitcnt vmcnt(NBI)                     C: (m1, n1), (m1, n0)
// Phase p spans barrier intervals 2p, 2p+1 for wm = 0 and 2p+1, 2p+2 for wm = 1 (X = 0, Y = 1 of K-tile kt, X = 2 of kt + 1 ...):
//   WAR  A slot of the other buffer (A(kt-1)): last read in L(Y(kt-1)) - by this group 2 intervals, by the staggered group ONE interval
//        (and one barrier) before this group's DMA issue in L(X(kt)); B slot (B(kt)): last read in L(X(kt)), re-staged in L(Y(kt)).
//        Round 6: the barrier that closes an L segment is entered behind s_waitcnt lgkmcnt(0) (SC_BAR_L), so those reads have RETURNED
//        before any wave passes it - the order is a counter's, not LDS arrival order plus a DMA's memory round trip;
//   RAW  A(kt+1) (issued X(kt)) and B(kt+1) (issued Y(kt-1)) are retired by every wave's vmcnt(NBI) in L(Y(kt)) and first read in
//        L(X(kt+1)), one barrier later for either group.
// The four-phase form (SC_GEMM_4PHASE, A/B builds) as it was:
// K-tile kt (buffer par = (kt + base) & 1) = 4 phases, each [L segment | barrier | C segment: 16 MFMA | barrier]:
//   P0  L: read A(m-sub 0) 8 x b128 + B(n-sub 0) 4 x b128            C: quadrant (m0, n0)
//   P1  L: read B(n-sub 1) 4 ; DMA A(kt+1) -> other buffer            C: (m0, n1)
//   P2  L: read A(m-sub 1) 8                                          C: (m1, n1)
//   P3  L: DMA B(kt+2) -> this buffer ; s_waitcnt vmcnt(NBI)          C: (m1, n0)   (B(n0) kept in registers)
// Hazards (phase p spans barrier intervals 2p, 2p+1 for wm = 0 and 2p+1, 2p+2 for wm = 1):
//   WAR  a slot is re-staged >= 2 phases after its last ds_read (A: read P0/P2, staged P1 of the next tile;
//        B: read P0/P1, staged P3), so the staggered group's reads have retired (lgkmcnt) a barrier earlier;
//   RAW  tile kt+1 (A issued P1(kt), B issued P3(kt-1)) is retired by every wave's vmcnt(NBI) in L(P3(kt)) and first
//        read in L(P0(kt+1)), one barrier later for either group.
//
// PERSISTENT: min(tiles, CUs) workgroups walk the tile list (virtual block id = i * gridDim.x + blockIdx.x through the
// same XCD-aware banded map as a plain launch).  When a tile's K loop ends, K-tile 0 of the NEXT tile is DMA'd into the
// ring buffer that the last K-tile did not use, and only then the epilogue runs - in the other buffer (8 KiB per wave,
// four 32-row passes) - so the next tile's first-load latency (3-4 us, measured with in-kernel stamps:
// tools/epi_stamps.py) hides under the GELU / store work instead of following it.  B(1) of the next tile is issued
// after the epilogue's closing barrier (it lands in the epilogue's buffer).  `base` carries the buffer parity across tiles.
#include <algorithm>

#include "sc_common.h"

namespace {

constexpr int BK = 64, ROWB = 128;
constexpr int HALF_BYTES = 128 * ROWB;           // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;        // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF_BYTES;         // 128 KiB
constexpr int EPI_BYTES = BUF_BYTES / 8;         // per-wave epilogue staging inside ONE ring buffer: 8 KiB
// LN = 2 (row-statistics producer, see below): per wave [32 rows][8 chunks] (sum, sum of squares) of one epilogue pass, then
// [128 rows] per wave for the whole tile, behind the ring
constexpr int STATL_BYTES = 8 * 32 * 4 * 8;      // 8 KiB (chunk pairs: adjacent lanes are added by DPP first)
constexpr int STATW_BYTES = 8 * 128 * 8;         // 8 KiB
// LN != 0: the tile's 256 statistics rows (first 4 strips = 32 bytes each) and two vectors of column constants, prefetched by
// LDS-DMA during K-tile 0 so that the epilogue never waits for global memory
constexpr int ROWST_BYTES = 256 * 32;            // 8 KiB
constexpr int COLC_BYTES = 2 * 256 * 4;          // 2 KiB
constexpr int SC_STAT_STRIDE = 8;                // strips per row in a statistics buffer: [M][8][2] fp32 (one 64-byte line per row)

#define SC_BAR()                               \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// The barrier that closes an L segment of the two-phase K loop (round 6): every ds_read of the segment has RETURNED
// (s_waitcnt lgkmcnt(0)) before the wave enters the barrier, so the other wave group's LDS-DMA into the slot those reads came from -
// issued one barrier interval later - is ordered behind them by a counter, not by "a DMA takes a memory round trip to arrive".  The
// MFMAs behind the barrier need the fragments anyway.  Measured on the step's eight GEMM shapes, same box, alternating rounds
// (tools/ab_gemm_libs.py): 3016 us with the wait vs 3026 us without, bit-identical outputs.  SC_GEMM_NO_LGKM_BAR: the round-5 form (A/B).
#ifndef SC_GEMM_NO_LGKM_BAR
#define SC_BAR_L()                                              \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        SC_BAR();                                               \
    } while (0)
#else
#define SC_BAR_L() SC_BAR()
#endif

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DIAG (diagnostic builds): 3 = no epilogue (timing only, results wrong), 4 = wall-clock stamps of each tile's sections
// written to p.Ct (results right; tools/epi_stamps.py).
// DIAG = 5 is not a diagnostic but the TN form (p.tn): both operands are stored with the REDUCTION index as their row index,
//   C[m, n] = sum_r A[r, m] W[r, n]        (a weight gradient dW = dY^T X read straight from the row-major activations),
// so a K-tile is staged as [64 r][256 m] / [64 r][256 n] and the MFMA fragments (8 consecutive r per lane) are read with
// ds_read_b64_tr_b16, the LDS transpose read of gfx950.  LDS image of a half-tile ([64 r][128 columns], 16 KiB, same ring
// slots as the NT form): 8-row x 32-column subtiles of 512 B,
//   off(r, ch) = 2048 (r >> 3) + 512 (ch >> 2) + 64 (r & 7) + 16 ((ch & 3) ^ ((r >> 2) & 3))      (ch = 16-byte chunk of the row),
// conflict-free for the transposed reads of the 16x16x32 operand (cdna_hip_programming T10, image (a)); the LDS-DMA writes it
// linearly (one wave instruction = two subtiles), the swizzle sits in the per-lane global source address.  Four base registers
// serve all fragment reads, the rest are immediates.  ACT = 1: erf-GELU in the epilogue (compile-time so the 128
// evaluations per lane form straight-line code).
// BN = 256 or 192 output columns per tile (wave block 128 x 64 or 128 x 48).  The narrower tile exists for wave
// quantisation: with M = 32768, N = 768 / 2304 give 384 / 1152 tiles of 256 x 256 (1.5 / 4.5 rounds over 256 CUs) but
// 512 / 1536 tiles of 256 x 192 (exactly 2 / 6 rounds).
//
// LN: LayerNorm without a LayerNorm kernel (round 3; the encoder layer's two LayerNorms used to be 25 launches and 2.4 GB of
// HBM traffic per forward).  A LayerNorm sits between a residual GEMM (out_proj / fc2: "producer") and the next GEMM (fc1 / QKV:
// "consumer"), and its output is also the next residual.  Here the residual stream stays RAW (pre-LayerNorm rows, bf16) and
//   LN = 2  the producer's epilogue also emits, per output row and column strip (= N-tile), the sum and the sum of squares of
//           the bf16 values it stored (p.stats_out [M][8][2]); if its residual operand is itself a raw row (p.res_stats), it is
//           normalised on the fly: r = (raw - mean) rstd gamma[n] + beta[n];
//   LN = 1  the consumer multiplies the RAW rows with W' = W diag(gamma) (folded on the host) and finishes in its epilogue:
//           y = rstd_m (acc - mean_m s_n) + c_n,  s_n = sum_k W'[n,k],  c_n = sum_k beta[k] W[n,k] + bias[n]  (p.ln_colsum, p.bias),
//           mean_m / rstd_m from the producer's strips (p.ln_stats, p.ln_ns).
template <int DIAG, int BN, int ACT, int DROP, int RES, int LN>
__global__ __launch_bounds__(512) void gemm256_kernel(const sc_gemm_args p) {
    constexpr int TN = BN / 4, FN = TN / 16, NB1 = FN - 2;   // per-wave columns, fragments, fragments of n-sub 1
    constexpr int NBI = BN / 64;                             // B-tile DMA instructions per wave and K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LN_OFF_ROWST = LDS_BYTES + (LN == 2 ? STATL_BYTES + STATW_BYTES : 0);
    constexpr int LN_OFF_COLC = LN_OFF_ROWST + ROWST_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nM = (p.M + 255) >> 8, nN = (p.N + BN - 1) / BN;
    const int n_tiles = nM * nN;
    // ---- virtual block -> tile.  Blocks b, b + 8, ... share an XCD and its L2.  Round 4: the M-tiles are split among the XCDs
    // FIRST (XCD x owns M-tiles [nM x / 8, nM (x + 1) / 8)) and each XCD walks its own rows in bands of 8 M-tiles x all N-tiles, so an A
    // row-tile is only ever filled into ONE L2, whatever M is.  The round-1..3 map cut the band-major tile list into 8 equal runs:
    // aligned with the bands only when nM was a multiple of 64 (B x 512 rows), and with the ragged layout's arbitrary row counts a
    // band straddled two XCDs most of the time (PMC: 487 MB per launch at M = 64 x 504 against 434 MB at M = 64 x 512).  An XCD's list
    // may be one row of tiles shorter than its neighbour's; a virtual block past its XCD's list has no tile (returns false).
    const bool xcd_rows = (gridDim.x & 7) == 0 && nM >= 8 && !p.reserved3;      // reserved3: A/B switch (sc_set_option(3, 1): round-3 map)
    auto tile_of = [&](int vb, int& m0, int& n0) -> bool {
        constexpr int GM = 8;
        if (xcd_rows) {
            const int xcd = vb & 7, idx = vb >> 3;
            const int m_lo = (nM * xcd) >> 3, m_hi = (nM * (xcd + 1)) >> 3;
            if (idx >= (m_hi - m_lo) * nN) return false;
            const int band = idx / (GM * nN), first_m = m_lo + band * GM;
            const int gm = min(GM, m_hi - first_m);
            const int within = idx - band * GM * nN;
            m0 = (first_m + within % gm) << 8;
            n0 = (within / gm) * BN;
            return true;
        }
        if (vb >= n_tiles) return false;
        const int xcd = vb & 7, q = n_tiles >> 3, r = n_tiles & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
        const int band = L / (GM * nN), first_m = band * GM;
        const int gm = min(GM, nM - first_m);
        const int within = L - band * GM * nN;
        m0 = (first_m + within % gm) << 8;
        n0 = (within / gm) * BN;
        return true;
    };

    const int z = blockIdx.z, z1 = z / p.nb2, z2 = z % p.nb2;
    const uint16_t* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const uint16_t* W = p.W + z1 * p.sW1 + z2 * p.sW2;
    const float* bias = p.bias ? p.bias + z1 * p.sBias1 + z2 * p.sBias2 : nullptr;
    const uint16_t* Rs = (RES && p.residual) ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;   // RES: compile-time variant (registers)
    const int64_t coff = z1 * p.sC1 + z2 * p.sC2;

    // ---- DMA sources: one wave-instruction = 8 rows; A: 2 halves x 16 instr (wave w issues {w, w + 8} of each half),
    //      B: BN / 8 instr over the BN-row B region (wave w issues {w, w + 8, ...}) --------------------------------
    const uint16_t* a_src[2][2];
    const uint16_t* b_src[NBI];
    constexpr bool TNM = (DIAG == 5);
    auto set_sources = [&](int m0, int n0) {
        int sl = lane;                           // opaque copy: keeps the lane-only terms from being hoisted out of the
        asm volatile("" : "+v"(sl));             // tile loop and living across the K loop
        if constexpr (TNM) {
            // wave instruction j of a half (16 per half): subtile row group j >> 1, column groups 2 (j & 1) + {0, 1}
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int j = i * 8 + wave;
                    const int row = 8 * (j >> 1) + ((sl >> 2) & 7);
                    const int ch = 4 * (2 * (j & 1) + (sl >> 5)) + ((sl & 3) ^ ((row >> 2) & 3));
                    a_src[h][i] = A + (int64_t)row * p.lda + m0 + h * 128 + ch * 8;
                }
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const int j = (i * 8 + wave) & 15, hb = (i * 8 + wave) >> 4;
                const int row = 8 * (j >> 1) + ((sl >> 2) & 7);
                const int ch = 4 * (2 * (j & 1) + (sl >> 5)) + ((sl & 3) ^ ((row >> 2) & 3));
                b_src[i] = W + (int64_t)row * p.ldw + n0 + hb * 128 + ch * 8;
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (i * 8 + wave) * 8 + (sl >> 3);            // row inside the half-tile
                const int c = (sl & 7) ^ ((row >> 1) & 7);
                a_src[h][i] = A + (int64_t)min(m0 + h * 128 + row, p.M - 1) * p.lda + c * 8;
            }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int row = (i * 8 + wave) * 8 + (sl >> 3);                // row inside the B region
            const int c = (sl & 7) ^ ((row >> 1) & 7);
            b_src[i] = W + (int64_t)min(n0 + row, p.N - 1) * p.ldw + c * 8;
        }
    };
    auto dma_A = [&](int par, int k0) {
        const int64_t ko = TNM ? (int64_t)k0 * p.lda : (int64_t)k0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                glds16(a_src[h][i] + ko, smem + par * BUF_BYTES + h * HALF_BYTES + (i * 8 + wave) * 1024);
    };
    auto dma_B = [&](int par, int k0) {
        const int64_t ko = TNM ? (int64_t)k0 * p.ldw : (int64_t)k0;
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            glds16(b_src[i] + ko, smem + par * BUF_BYTES + 2 * HALF_BYTES + (i * 8 + wave) * 1024);
    };
    // LN: statistics rows of the tile (consumer: of A's rows; producer: of the raw residual's rows) and the column constants
    // (consumer: s_n, c_n; producer: gamma, beta of the residual's LayerNorm) -> LDS.  One 16-byte piece per lane: wave w brings
    // rows 32 w .. 32 w + 31 (two pieces each), waves 0 / 1 one vector of 256 columns each.
    auto dma_ln = [&](int m0, int n0) {
        const float* st = LN == 1 ? p.ln_stats : p.res_stats;
        if (LN != 0 && st) {
            int sl = lane;
            asm volatile("" : "+v"(sl));
            const int row = wave * 32 + (sl >> 1);
            glds16(st + (int64_t)min(m0 + row, p.M - 1) * (2 * SC_STAT_STRIDE) + (sl & 1) * 4, smem + LN_OFF_ROWST + wave * 1024);
            if (wave < 2) {
                const float* v = LN == 1 ? (wave == 0 || !bias ? p.ln_colsum : bias) : (wave == 0 ? p.res_gamma : p.res_beta);
                glds16(v + min(n0 + sl * 4, p.N - 4), smem + LN_OFF_COLC + wave * 1024);
            }
        }
    };
#define SC_WAIT_NBI()                                                      \
    do {                                                                   \
        if (NBI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     \
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");              \
    } while (0)

    // ---- fragment read offsets: row = 16*f + (lane & 15)  =>  swizzle term depends on the lane only ----------
    const int sw = (lane >> 1) & 7;
    const int frag_off0 = (lane & 15) * ROWB + (((lane >> 4)) ^ sw) * 16;          // kk = 0
    const int frag_off1 = (lane & 15) * ROWB + ((4 + (lane >> 4)) ^ sw) * 16;      // kk = 1
    const int a_base = wm * HALF_BYTES;                                             // wave's A half
    // NT: wave's TN rows of the B region; TN form: half wn >> 1 of the B region, column groups 2 (wn & 1) + {0, 1} of it
    const int b_base = TNM ? 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + (wn & 1) * 1024 : 2 * HALF_BYTES + wn * TN * ROWB;
    // TN form: a 16 x 32 operand fragment (16 columns f of the staged tile, 32 reduction rows kk) = two transposed reads, rows
    // 32 kk + 8 g + 4 s + (0..3) for s = 0, 1: lane 4 q + pp of a 16-lane group supplies row q, chunk 2 f + (pp >> 1), byte 8 (pp & 1)
    int tr_base[2][2];                                                              // [s][f & 1]
    {
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                tr_base[s2][e] = 2048 * g + 64 * (4 * s2 + q) + 16 * ((2 * e + (pp >> 1)) ^ ((2 * g + s2) & 3)) + 8 * (pp & 1);
    }
    auto ld_frag = [&](const char* base, int f, int kk) -> bf16x8 {                 // fragment f (16 columns) of a half-tile image
        if constexpr (TNM) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            const char* b0p = base + 512 * (f >> 1) + 8192 * kk;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b0p + tr_base[0][f & 1]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(b0p + tr_base[1][f & 1]));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *(const bf16x8*)(base + f * 16 * ROWB + (kk ? frag_off1 : frag_off0));
        }
    };
    // TN form with ragged slices: slice z1 covers reduction rows [z1 K, min((z1 + 1) K, k_total))
    const int nk = (TNM && p.k_total > 0 ? min(p.K, p.k_total - (int)(blockIdx.z / p.nb2) * p.K) : p.K) / BK;
    // K-tile kt -> first k of the tile.  Conv-shaped A (tap_c = C_in, K = 3 C_in, lda = 2 C_in): per 64-channel block visit tap 0,
    // tap 2, tap 1 - tap 2 of row r is tap 0 of row r + 1, so that tile is fetched again while the L2 still holds it.
    const int tap_c = p.tap_c;
    auto koff = [&](int kt) -> int {
        if (tap_c == 0) return kt * BK;
        const int c = kt / 3, j = kt - 3 * c;
        return (j == 0 ? 0 : (3 - j) * tap_c) + c * BK;
    };
    // train-mode dropout is a compile-time variant: in the runtime-switched form its mask / scale registers and the branch in
    // the store loop cost the plain instantiations 4 % (rocprofv3, r01 v6 -> v7)
    const uint32_t drop_thr = DROP ? (uint32_t)(p.drop_p * 65536.f + 0.5f) : 0u;
    const float drop_scale = DROP ? 1.f / (1.f - p.drop_p) : 1.f;

    int iter_ = 0;
#define SC_STAMP(K)                                                                                          \
    do {                                                                                                     \
        if (DIAG == 4 && (blockIdx.x & 63) == 0 && lane == 0 && iter_ < 8)                                   \
            ((long long*)p.Ct)[(((blockIdx.x >> 6) * 8 + iter_) * 8 + wave) * 8 + (K)] = wall_clock64();     \
    } while (0)

#define SC_STAMPC(K)                                                                                         \
    do {                                                                                                     \
        if (DIAG == 4 && (blockIdx.x & 63) == 0 && lane == 0 && iter_ < 8)                                   \
            ((long long*)p.Ct)[(((blockIdx.x >> 6) * 8 + iter_) * 8 + wave) * 8 + (K)] = clock64();          \
    } while (0)

#define SC_MFMA_QUAD(MS, NS, BF)                                                                         \
    do {                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                 \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                             \
                _Pragma("unroll") for (int ni = 0; ni < ((NS) == 0 ? 2 : NB1); ++ni)                     \
                    acc[(MS) * 4 + mi][(NS) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(         \
                        BF[ni][kk], af[mi][kk], acc[(MS) * 4 + mi][(NS) * 2 + ni], 0, 0, 0);             \
        __builtin_amdgcn_s_setprio(0);                                                                   \
    } while (0)

    // ---- first tile: K-tile 0 into buffer 0 -------------------------------------------------------------------
    int vb = blockIdx.x, m0, n0, base = 0;
    if (!tile_of(vb, m0, n0)) return;             // (a slot past its XCD's list: uniform for the workgroup, before any barrier / DMA)
    set_sources(m0, n0);
    dma_B(0, 0);
    dma_A(0, 0);

    while (true) {
        SC_STAMP(0);
        // accumulators start from the bias (a lane's fragment column is fixed: n = wave's first column + 16 ni + lane % 16),
        // so the epilogue neither waits on a bias load nor spends an add per element
        // The MFMA is issued with the WEIGHT fragment as its first operand: an accumulator fragment is then the transposed
        // 16 x 16 block, i.e. a lane owns ONE output row (m = 16 mi + lane % 16) and FOUR CONSECUTIVE columns
        // (n = 16 ni + 4 (lane / 16) + r) - the epilogue assembles 16-byte row chunks in registers (one lane-row exchange per
        // fragment pair) and needs no LDS staging.
        f32x4 acc[8][FN];
        int bl = lane >> 4;
        asm volatile("" : "+v"(bl));
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = n0 + wn * TN + ni * 16 + 4 * bl;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (LN != 1 && bias && n + 4 <= p.N) bv = *(const f32x4*)(bias + n);      // LN = 1: the bias is part of c_n (epilogue)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = bv;
        }
        bf16x8 af[4][2], b0[2][2], b1[NB1][2];

        // ---- prologue: A(0), B(0) are in flight (issued before the previous epilogue); B(1) goes to the other buffer ----
        if (nk > 1) {
            dma_B(base ^ 1, koff(1));
            SC_WAIT_NBI();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SC_STAMP(1);
        SC_STAMPC(6);                                // shader-clock ticks around the K loop (tools/epi_stamps.py: clock under load)
        SC_BAR();
        if (wm == 1) SC_BAR();                       // stagger the second wave group by one barrier interval

        int kt = 0;
        do {                                         // nk >= 1 (checked by the launcher): no zero-trip path to carry acc through
            const int par = (kt + base) & 1;
            const char* as = smem + par * BUF_BYTES + a_base;
            const char* bs = smem + par * BUF_BYTES + b_base;
#ifndef SC_GEMM_4PHASE
            // ---------------- X  (round 5: TWO phases of 32 MFMAs per K-tile instead of four of 16 - half the barriers; same operand
            // registers, same accumulation order => same bits.  X: L = every fragment of B and A(m-sub 0) + DMA A(kt+1) -> other buffer;
            // Y: L = A(m-sub 1) + DMA B(kt+2) -> this buffer + the counted wait.  Hazards: header.)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                b0[ni][0] = ld_frag(bs, ni, 0);
                b0[ni][1] = ld_frag(bs, ni, 1);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = ld_frag(as, mi, 0);
                af[mi][1] = ld_frag(as, mi, 1);
            }
#pragma unroll
            for (int ni = 0; ni < NB1; ++ni) {
                b1[ni][0] = ld_frag(bs, 2 + ni, 0);
                b1[ni][1] = ld_frag(bs, 2 + ni, 1);
            }
            if (LN != 0 && kt == 0) dma_ln(m0, n0);      // older than every later wait of this K loop: in LDS long before the epilogue
            if (kt + 1 < nk) dma_A(par ^ 1, koff(kt + 1));
            SC_BAR_L();
            SC_MFMA_QUAD(0, 0, b0);
            SC_MFMA_QUAD(0, 1, b1);
            SC_BAR();
            // ---------------- Y
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = ld_frag(as, 4 + mi, 0);
                af[mi][1] = ld_frag(as, 4 + mi, 1);
            }
            if (kt + 2 < nk) {
                dma_B(par, koff(kt + 2));
                SC_WAIT_NBI();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SC_BAR_L();
            SC_MFMA_QUAD(1, 1, b1);
            SC_MFMA_QUAD(1, 0, b0);
            SC_BAR();
#else
            // ---------------- P0
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                b0[ni][0] = ld_frag(bs, ni, 0);
                b0[ni][1] = ld_frag(bs, ni, 1);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = ld_frag(as, mi, 0);
                af[mi][1] = ld_frag(as, mi, 1);
            }
            SC_BAR();
            SC_MFMA_QUAD(0, 0, b0);
            SC_BAR();
            // ---------------- P1
#pragma unroll
            for (int ni = 0; ni < NB1; ++ni) {
                b1[ni][0] = ld_frag(bs, 2 + ni, 0);
                b1[ni][1] = ld_frag(bs, 2 + ni, 1);
            }
            if (LN != 0 && kt == 0) dma_ln(m0, n0);      // older than every later wait of this K loop: in LDS long before the epilogue
            if (kt + 1 < nk) dma_A(par ^ 1, koff(kt + 1));
            SC_BAR();
            SC_MFMA_QUAD(0, 1, b1);
            SC_BAR();
            // ---------------- P2
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = ld_frag(as, 4 + mi, 0);
                af[mi][1] = ld_frag(as, 4 + mi, 1);
            }
            SC_BAR();
            SC_MFMA_QUAD(1, 1, b1);
            SC_BAR();
            // ---------------- P3
            if (kt + 2 < nk) {
                dma_B(par, koff(kt + 2));
                SC_WAIT_NBI();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SC_BAR();
            SC_MFMA_QUAD(1, 0, b0);
            SC_BAR();
#endif
        } while (++kt < nk);
        if (wm == 0) SC_BAR();                       // pairs the staggered group's last barrier
        SC_STAMP(2);
        SC_STAMPC(7);

        // ---- next tile's K-tile 0 into the buffer the last K-tile did not use (its last reads are a K-tile old) ---------
        const int last_par = (nk - 1 + base) & 1;
        const int cm0 = m0, cn0 = n0;
        vb += gridDim.x;
        const bool has_next = tile_of(vb, m0, n0);
        if (has_next) {
            set_sources(m0, n0);
            base = last_par ^ 1;
            dma_B(base, 0);
            dma_A(base, 0);
        }

        if (DIAG == 3) {                             // timing-only: consume the accumulators without an epilogue
            float t = 0.f;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni) t += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
            if (t == 123.456f) ((float*)p.C)[tid] = t;
        } else {
            // ---- epilogue: each wave streams its 128 x TN block through a private 8 KiB region of buffer last_par,
            //      32 rows a pass: fragments -> LDS (fp32) -> full 16-B row chunks -> +residual -> bf16 stores ----------
            // lane id re-materialised through an opaque asm: everything below that depends only on the lane would otherwise be
            // hoisted out of the tile loop as loop-invariant and stay live across the K loop (VGPR budget: 256)
#include "gemm_epilogue.inc"
        }
        SC_STAMP(5);
        ++iter_;
        if (LN == 2) {
            // the tile's row statistics: the four column strips of the waves wn = 0..3 are added in order and leave as ONE strip
            // per (row, N-tile).  The barrier doubles as the one below.
            SC_BAR();
            if (tid < 256) {
                const int rw = tid & 127, g = tid >> 7;
                const f32x2* sw = (const f32x2*)(smem + LDS_BYTES + STATL_BYTES);
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const f32x2 t = sw[(g * 4 + w2) * 128 + rw];
                    a += t.x;
                    b += t.y;
                }
                const int m = cm0 + g * 128 + rw;
                if (m < p.M) *(f32x2*)(p.stats_out + ((int64_t)m * SC_STAT_STRIDE + cn0 / BN) * 2) = f32x2{a, b};
            }
            if (!has_next) break;
        } else {
            if (!has_next) break;
            SC_BAR();                                // every wave is done with its staging region before B(1) lands in it
        }
    }
}

}  // namespace

template <int DIAG, int BN, int ACT, int DROP, int RES, int LN>
static int launch256__(const sc_gemm_args& a, hipStream_t s) {
    constexpr int LDS = LDS_BYTES + (LN == 2 ? STATL_BYTES + STATW_BYTES : 0) + (LN != 0 ? ROWST_BYTES + COLC_BYTES : 0);
    static sc_lds_attr_once attr;
    if (hipError_t e = sc_set_max_lds_once(attr, gemm256_kernel<DIAG, BN, ACT, DROP, RES, LN>, LDS); e != hipSuccess) {
        sc_set_error("hipFuncSetAttribute(gemm256): %s", hipGetErrorString(e));
        return -3;
    }
    const int nM = (a.M + 255) / 256, nN = (a.N + BN - 1) / BN;
    dim3 grid(std::min(nM * nN, sc_num_cus()), 1, a.nb1 * a.nb2);
    hipLaunchKernelGGL((gemm256_kernel<DIAG, BN, ACT, DROP, RES, LN>), grid, dim3(512), LDS, s, a);
    SC_LAUNCH_CHECK();
    return 0;
}

// SC_DIAG_BUILD (libspeechclip_hip_diag.so, speechclip_plus_amd/build.py): the build that ALSO holds the timing-only / stamped kernels
// (DIAG 3 / 4: tile ids 32 / 34) and the opt-in LayerNorm-folded instantiations (LN 1 / 2; measured no faster).
// The product library carries neither - 8 instead of 24 instantiations of this kernel - and sc_gemm_bf16 refuses the tile ids / LN
// arguments with a message instead of running a kernel whose results are wrong by design (VERDICT r03 "weak" 9).
template <int DIAG, int BN, int ACT, int DROP>
static int launch256_(const sc_gemm_args& a, hipStream_t s) {
    if constexpr (DIAG == 0) {          // the residual epilogue is its own instantiation (its prefetch registers)
        if (a.residual) {
#ifdef SC_DIAG_BUILD
            if constexpr (ACT == 0) {   // statistics producer: residual GEMMs without an activation (out_proj, fc2)
                if (a.stats_out) return launch256__<DIAG, BN, ACT, DROP, 1, 2>(a, s);
            }
#endif
            return launch256__<DIAG, BN, ACT, DROP, 1, 0>(a, s);
        }
#ifdef SC_DIAG_BUILD
        if constexpr (DROP == 0) {      // LayerNorm-folded consumer (QKV, fc1): no dropout site there
            if (a.ln_stats) return launch256__<DIAG, BN, ACT, DROP, 0, 1>(a, s);
        }
#endif
    }
    return launch256__<DIAG, BN, ACT, DROP, 0, 0>(a, s);
}

template <int DIAG, int BN>
static int launch256(const sc_gemm_args& a, hipStream_t s) {
    if constexpr (DIAG == 0) {          // the diagnostic builds have no dropout variant
        if (a.aux_mode == 1) return launch256__<DIAG, BN, 3, 0, 0, 0>(a, s);      // erf-GELU fused with the aux operand (checked by sc_gemm_bf16:
        if (a.aux_mode == 2) return launch256__<DIAG, BN, 4, 0, 0, 0>(a, s);      // act = 1, no residual / dropout / transposed store / LN)
        if (a.drop_p > 0.f) return a.act == 1 ? launch256_<DIAG, BN, 1, 1>(a, s) : launch256_<DIAG, BN, 0, 1>(a, s);
    }
    return a.act == 1 ? launch256_<DIAG, BN, 1, 0>(a, s) : launch256_<DIAG, BN, 0, 0>(a, s);
}

// rounds of workgroups the grid needs with BN-wide tiles, times a per-tile efficiency factor
static double tile_cost(const sc_gemm_args& a, int BN) {
    const double tiles = (double)((a.M + 255) / 256) * ((a.N + BN - 1) / BN) * a.nb1 * a.nb2;
    const int cus = sc_num_cus();
    const double rounds = (double)(((long)tiles + cus - 1) / cus);
    return rounds * BN * (BN == 192 ? 1.12 : 1.0);       // measured: a 256 x 192 tile runs ~12 % below the 256 x 256 rate
}

// output tile width the dispatcher picks for this problem (the statistics producer writes one strip per N-tile: its consumers
// need the count)
int sc_gemm256_bn(const sc_gemm_args& a) {
    if (a.tile == 7) return 192;
    if (a.tile == 8 || a.tile == 32 || a.tile == 34) return 256;
    const bool ok192 = (a.n_split < 0 || a.n_split % 192 == 0);
    const bool ok256 = (a.n_split < 0 || a.n_split % 256 == 0);
    return (ok192 && (!ok256 || tile_cost(a, 192) < tile_cost(a, 256))) ? 192 : 256;
}

int sc_gemm256_launch(const sc_gemm_args& a_in, hipStream_t s) {
    sc_gemm_args a = a_in;
    // bit 0: non-temporal C stores.  auto = only when a residual is given (the output is the next residual stream and is read
    // next by a LayerNorm pass; those tiles leave through the LDS-staged epilogue as full 128-byte rows).  Extending it to every large
    // output was measured and reverted in round 1 (FC2 then reads its A operand from HBM instead of the MALL); on the
    // register-direct epilogue it must stay off: half-line non-temporal writes cost 25 % (75.7 vs 55.9 us on the out_proj shape).
    a.reserved = (a_in.reserved == 1 || (a_in.reserved == 0 && a_in.residual != nullptr)) ? 1 : 0;
    if (sc_option(1)) a.reserved = 0;      // A/B switch (tools/): plain stores everywhere
    a.reserved3 = sc_option(3);            // A/B switch (tools/): 1 = the round-3 block -> tile map (band-major list cut into 8 runs)
    if (a.tn) return launch256_<5, 256, 0, 0>(a, s);     // TN operands (weight gradients): checked by sc_gemm_bf16
#ifdef SC_DIAG_BUILD
    if (a.tile == 32) return launch256<3, 256>(a, s);   // diagnostics only (tools/epi_probe.py, tools/epi_stamps.py)
    if (a.tile == 34) return launch256<4, 256>(a, s);
#else
    if (a.tile == 32 || a.tile == 34 || a.ln_stats || a.stats_out || a.res_stats) {
        sc_set_error("sc_gemm_bf16: tile ids 32 / 34 (timing-only / stamped kernels) and the LayerNorm-folded GEMMs are built into "
                     "libspeechclip_hip_diag.so only (speechclip_plus_amd/build.py; _lib.diag_lib())");
        return -1;
    }
#endif
    if (a.tile == 7) return launch256<0, 192>(a, s);
    if (a.tile == 8) return launch256<0, 256>(a, s);
    if (sc_gemm256_bn(a) == 192) return launch256<0, 192>(a, s);
    return launch256<0, 256>(a, s);
}
