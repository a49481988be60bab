// HuBERT positional convolution (grouped Conv1d, kernel 128, 16 groups, padding 64, last frame dropped) + bias + GELU + residual
// for gfx950:   out[b, t, g*DG + n] = gelu(bias + sum_{j < 128} sum_{ci < DG} W[g][n][j*DG + ci] . x[b, t + j - 64, g*DG + ci]) + x[b, t, g*DG + n]
// (fairseq TransformerEncoder.pos_conv + SamePad + GELU, /root/reference/avssl/module/speech_encoder_plus.py:32-37).
//
// As a GEMM per (group, utterance) this is [R x 48] = [R x 6144] . [6144 x 48] with a TOEPLITZ left operand: row t of it is the 128-frame
// window starting at padded frame t, so consecutive rows share 127 / 128 of their data.  The generic GEMM (sc_gemm_bf16, lda = 48) stages
// that window matrix K-tile by K-tile: 1.5 MiB of LDS-DMA per 128-row tile for 24 KiB of distinct input - the launch was bound by the
// L2 -> LDS path (8.8 GB per launch, 0.55 ms, 0.23 of the MFMA peak).  Here the input slab of one (group, utterance, frame block) -
// (64 NW + 127) frames x DG channels - is loaded into LDS ONCE and every tap reads it at a row offset; only the weights stream
// (one 9 KiB tile per PAIR of taps = 96 k = 3 MFMA k-steps, three-stage LDS-DMA ring, two tiles ahead, one barrier per pair).
//
// One workgroup = NW waves = 64 NW output frames x DG channels; wave w owns frames 64 w .. 64 w + 63 = 4 x (DG / 16) fragments of
// v_mfma_f32_16x16x32_bf16 (weight fragment first: a lane owns one frame and 4 consecutive channels).  Same k order, bias-initialised
// accumulators, GELU and rounding as the GEMM path: results are bit-identical to it (tests/test_gpu_kernels.py).
//
// LDS: slab rows (frames) at 128-byte pitch, 16-byte atoms XOR-swizzled by (row >> 1) & 7 (reads at arbitrary row offsets: at most
// two-way conflicts); weight tile = per k-step a [DG rows x 64 B] image, atoms swizzled by {0, 2, 3, 1}[(row >> 2) & 3] (conflict-free).
#include "sc_common.h"

namespace {

template <int DG, int NW>                         // channels per group; waves per workgroup (8: one workgroup per CU, 4: two)
__global__ __launch_bounds__(NW * 64) void posconv_kernel(const uint16_t* __restrict__ xg, const uint16_t* __restrict__ w,
                                                      const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                      uint16_t* __restrict__ out, int B, int R, int D, int G, int Kp, int Rp,
                                                      const int32_t* __restrict__ row0, int seg_rows) {
    constexpr int FN = DG / 16;                   // channel fragments per wave
    constexpr int KS = 2 * DG / 32;               // MFMA k-steps per tap pair
    constexpr int APR = DG / 8;                   // 16-byte atoms per frame
    constexpr int BT_BYTES = KS * DG * 64;        // one weight tile (a tap pair)
    constexpr int PC_ROWS = NW * 64;              // output frames per workgroup
    constexpr int SLAB_ROWS = PC_ROWS + 128;      // + kernel - 1 (127), rounded to the DMA piece (8 rows)
    constexpr int SLAB_BYTES = SLAB_ROWS * 128;
    constexpr int NPW = (KS * FN + NW - 1) / NW;  // weight DMA pieces per wave and tap pair
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;

    // ---- block -> (group, utterance, frame block): the workgroups of one group share its 0.6 MiB of weights, so a group stays on
    //      one XCD (blocks b, b + 8, ... share an XCD and its L2)
    const int mb = (R + PC_ROWS - 1) / PC_ROWS;
    int g, bi, m0;
    {
        const int per_g = B * mb;
        int L;
        if ((G & 7) == 0) {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;        // idx in [0, (G / 8) * per_g)
            g = xcd * (G >> 3) + idx / per_g;
            L = idx % per_g;
        } else {
            g = blockIdx.x / per_g;
            L = blockIdx.x % per_g;
        }
        bi = L / mb;
        m0 = (L % mb) * PC_ROWS;
    }
    const uint16_t* xs = xg + ((int64_t)g * B + bi) * Rp * DG;            // padded frames of this (group, utterance): [Rp, DG]
    int64_t orow = (int64_t)bi * R;                                       // first output row of the utterance
    if (row0) {
        // ragged rows (sc_segments): R = the longest pitch (grid sizing); the utterance's own pitch decides what this workgroup does
        const int r0 = row0[bi];
        R = row0[bi + 1] - r0;
        if (m0 >= R) return;                                              // uniform for the workgroup, before any barrier
        Rp = R + Kp;
        xs = xg + ((int64_t)g * (seg_rows + (int64_t)Kp * B) + r0 + (int64_t)Kp * bi) * DG;
        orow = r0;
    }
    const uint16_t* wg = w + (int64_t)g * DG * Kp * DG;                   // [DG out, Kp * DG] tap-major
    const int ldw = Kp * DG;
    const int ntp = Kp / 2;                                               // tap pairs

    // bias first (the accumulators start from it): its loads retire before the LDS-DMA below in the in-order vmcnt
    f32x4 bv[FN];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) bv[ni] = bias ? *(const f32x4*)(bias + g * DG + ni * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");

    // ---- slab: pieces of 8 rows; wave w issues pieces w, w + NW, ...
#pragma unroll
    for (int i = 0; i < SLAB_ROWS / (8 * NW); ++i) {
        const int p = i * NW + wave;
        const int row = p * 8 + (lane >> 3);
        const int at = (lane & 7) ^ ((row >> 1) & 7);
        const int rr = min(m0 + row, Rp - 1);
        const uint16_t* src = xs + (int64_t)rr * DG + (at < APR ? at * 8 : 0);      // atoms >= APR of a row are never read
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, 0, 0);
    }
    // ---- weight tiles: KS * FN pieces of 16 rows x 64 B per tap pair, NPW per wave (9 pieces at DG = 48: a wave whose last
    //      piece would be past the end repeats its first - identical bytes to the same place - so every wave counts the same)
    const uint16_t* bsrc[NPW];
    int bdst[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int pc = wave + NW * i;
        if (pc >= KS * FN) pc = wave;
        const int s = pc / FN, nb = pc % FN;
        const int n = 16 * nb + (lane >> 2);
        const int gt = (n >> 2) & 3;
        const int f = ((gt << 1) & 3) ^ ((gt >> 1) * 3);                  // {0, 2, 3, 1}
        const int at = (lane & 3) ^ f;
        bsrc[i] = wg + (int64_t)n * ldw + s * 32 + at * 8;
        bdst[i] = SLAB_BYTES + (s * DG + 16 * nb) * 64;
    }
    auto dma_w = [&](int tp, int st) {
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + tp * 2 * DG),
                                             (__attribute__((address_space(3))) void*)(smem + bdst[i] + st * BT_BYTES), 16, 0, 0);
    };
    dma_w(0, 0);
    dma_w(1, 1);                                  // Kp >= 4 (checked by the launcher)

    // ---- accumulators start from the bias
    f32x4 acc[4][FN];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = bv[ni];

    // ---- per-lane constants of the fragment reads.  k-step s, lane group q -> atom 4 s + q of the tap pair: tap 2 tp + jq, channel atom ca
    int jq[KS], ca[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int a = 4 * s + q;
        jq[s] = a / APR;
        ca[s] = a % APR;
    }
    const int rb = wave * 64 + l15;                                       // + 16 mi: the lane's frame inside the block
    const int gtb = (l15 >> 2) & 3;
    const int boff = SLAB_BYTES + l15 * 64 + ((q ^ (((gtb << 1) & 3) ^ ((gtb >> 1) * 3))) << 4);

    int st = 0;
    for (int tp = 0; tp < ntp; ++tp) {
        // tile tp landed for this wave (tile tp + 1's two pieces may stay in flight); after the barrier it landed for every wave and
        // every wave is done reading stage (tp + 2) % 3
        if (tp + 1 < ntp) {
            if (NPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (NPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tp + 2 < ntp) dma_w(tp + 2, st == 0 ? 2 : st - 1);
        const char* bs = smem + boff + st * BT_BYTES;
        st = (st == 2) ? 0 : st + 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 bf[FN], af[4];
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) bf[ni] = *(const bf16x8*)(bs + (s * DG + 16 * ni) * 64);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = rb + 16 * mi + 2 * tp + jq[s];
                af[mi] = *(const bf16x8*)(smem + row * 128 + ((ca[s] ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < FN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
    }

    // ---- epilogue: GELU, + residual, bf16; a lane stores 4 consecutive channels of its frame per fragment
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int t = m0 + wave * 64 + 16 * mi + l15;
        if (t >= R) continue;
        const int64_t o = (orow + t) * D + g * DG + 4 * q;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const f32x4 v = acc[mi][ni];
            const f32x2 g0 = gelu_bf2(f32x2{v[0], v[1]}), g1 = gelu_bf2(f32x2{v[2], v[3]});
            float r0 = g0.x, r1 = g0.y, r2 = g1.x, r3 = g1.y;
            if (res) {
                const uint2 rv = *(const uint2*)(res + o + ni * 16);
                r0 += bflo(rv.x); r1 += bfhi(rv.x); r2 += bflo(rv.y); r3 += bfhi(rv.y);
            }
            uint2 u;
            u.x = pack2bf(r0, r1);
            u.y = pack2bf(r2, r3);
            *(uint2*)(out + o + ni * 16) = u;
        }
    }
}

template <int DG, int NW>
static int launch_posconv(const uint16_t* xg, const uint16_t* w, const float* bias, const uint16_t* res, uint16_t* out, int B, int R,
                          int D, int G, int Kp, int Rp, hipStream_t s, const int32_t* row0 = nullptr, int seg_rows = 0) {
    constexpr int PC_ROWS = NW * 64;
    constexpr int LDS = (PC_ROWS + 128) * 128 + 3 * (2 * DG / 32) * DG * 64;
    static sc_lds_attr_once attr;
    if (hipError_t e = sc_set_max_lds_once(attr, posconv_kernel<DG, NW>, LDS); e != hipSuccess) {
        sc_set_error("hipFuncSetAttribute(posconv): %s", hipGetErrorString(e));
        return -3;
    }
    const int mb = (R + PC_ROWS - 1) / PC_ROWS;
    hipLaunchKernelGGL((posconv_kernel<DG, NW>), dim3(G * B * mb), dim3(NW * 64), LDS, s, xg, w, bias, res, out, B, R, D, G, Kp, Rp, row0, seg_rows);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int sc_posconv_bf16(const sc_bf16* xg, const sc_bf16* w, const float* bias, const sc_bf16* residual, sc_bf16* out,
                               int32_t B, int32_t R, int32_t D, int32_t G, int32_t Kp, int32_t Rp, void* stream) {
    SC_CHECK(xg && w && out, "sc_posconv_bf16: null pointer");
    SC_CHECK(B > 0 && R > 0 && G > 0 && D % G == 0, "sc_posconv_bf16: bad shape B=%d R=%d D=%d G=%d", B, R, D, G);
    const int DG = D / G;
    SC_CHECK(DG == 48 || DG == 64, "sc_posconv_bf16: channels per group must be 48 or 64 (D=%d, G=%d)", D, G);
    SC_CHECK(Kp == 128 && Rp >= R + Kp - 1, "sc_posconv_bf16: kernel must be 128 taps and Rp >= R + 127 (Kp=%d, Rp=%d)", Kp, Rp);
    SC_CHECK(((uintptr_t)xg % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
                 (!residual || ((uintptr_t)residual % 8) == 0) && (!bias || ((uintptr_t)bias % 16) == 0),
             "sc_posconv_bf16: alignment");
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *xp = (const uint16_t*)xg, *wp = (const uint16_t*)w, *rp = (const uint16_t*)residual;
    uint16_t* op = (uint16_t*)out;
    // Dg = 48: 256-frame workgroups of 4 waves, TWO per CU (76 KiB of LDS each; they drift apart, so one's operand waits and barrier
    // fall under the other's MFMAs: 283 vs 306 us at the step's shape); Dg = 64: its 96 KiB allow one workgroup per CU, 8 waves x 512 frames
    // (396 vs 514 us)
    return DG == 48 ? launch_posconv<48, 4>(xp, wp, bias, rp, op, B, R, D, G, Kp, Rp, s) : launch_posconv<64, 8>(xp, wp, bias, rp, op, B, R, D, G, Kp, Rp, s);
}

extern "C" int sc_posconv_seg_bf16(const sc_bf16* xg, const sc_bf16* w, const float* bias, const sc_bf16* residual, sc_bf16* out,
                                   const sc_segments* seg, int32_t D, int32_t G, int32_t Kp, void* stream) {
    SC_CHECK(xg && w && out && seg && seg->row0, "sc_posconv_seg_bf16: null pointer");
    SC_CHECK(seg->B > 0 && seg->max_pitch > 0 && seg->rows > 0 && G > 0 && D % G == 0, "sc_posconv_seg_bf16: bad shape B=%d D=%d G=%d", seg->B, D, G);
    const int DG = D / G;
    SC_CHECK(DG == 48 || DG == 64, "sc_posconv_seg_bf16: channels per group must be 48 or 64 (D=%d, G=%d)", D, G);
    SC_CHECK(Kp == 128, "sc_posconv_seg_bf16: kernel must be 128 taps (Kp=%d)", Kp);
    SC_CHECK(((uintptr_t)xg % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 8) == 0 &&
                 (!residual || ((uintptr_t)residual % 8) == 0) && (!bias || ((uintptr_t)bias % 16) == 0),
             "sc_posconv_seg_bf16: alignment");
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *xp = (const uint16_t*)xg, *wp = (const uint16_t*)w, *rp = (const uint16_t*)residual;
    uint16_t* op = (uint16_t*)out;
    const int R = seg->max_pitch;
    return DG == 48 ? launch_posconv<48, 4>(xp, wp, bias, rp, op, seg->B, R, D, G, Kp, R + Kp, s, seg->row0, seg->rows)
                    : launch_posconv<64, 8>(xp, wp, bias, rp, op, seg->B, R, D, G, Kp, R + Kp, s, seg->row0, seg->rows);
}
