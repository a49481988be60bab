// Weight gradient of the grouped positional convolution (fully trainable HuBERT, avssl/module/speech_encoder_plus.py:29-40 under
// trainable: true):
//     gw[g][co][tap Dg + ci] = sum_m du[g][m][co] * xg[g][m + tap][ci]          m = b Rp + t over the halo-padded slab rows
// i.e. per (group, tap) a Dg x Dg correlation of two [rows, Dg] matrices, the second one shifted by `tap` rows.  As a GEMM
// (dy^T . Toeplitz view, 48 x 6144 outputs per group) it is a bad shape for every tile family - 48 output rows, 40 960
// reduction rows - and ran at 0.12 PFLOP/s behind two transposes; here a workgroup owns (group, block of taps, slice of rows),
// stages 128-row chunks of du and of xg (plus TB - 1 rows for the shifts) linearly in LDS by LDS-DMA, and every wave reads its
// operands with ds_read_b64_tr_b16 (the reduction index is the ROW index of both LDS images): the du fragments once per k-step,
// the xg fragments at a row offset = its tap.  Accumulators: TPW taps x (Dg / 16)^2 fragments per wave.
// Row slices (Z) are separate fp32 partials, reduced by sc_colsum_f32 in slice order (deterministic).
#include "sc_common.h"

namespace {

__device__ __forceinline__ void glds16_(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef short s16x4_ __attribute__((ext_vector_type(4)));
typedef short s16x8_ __attribute__((ext_vector_type(8)));

template <int DG, int TPW>      // channels per group (48 / 64), taps per wave; 4 waves -> TB = 4 TPW taps per workgroup
__global__ __launch_bounds__(256) void posconv_wgrad_kernel(const uint16_t* __restrict__ du, const uint16_t* __restrict__ xg,
                                                            float* __restrict__ part, int64_t rows, int Kp, int64_t total_elems) {
    constexpr int NB = DG / 16, TB = 4 * TPW, CH = 128;
    constexpr int PITCH = DG * 2;                            // bytes per LDS row (linear image)
    constexpr int DU_BYTES = CH * PITCH;                     // 12 / 16 KiB
    constexpr int X_ROWS = CH + TB;                          // rows the shifted reads touch (TB - 1 needed, one spare)
    constexpr int DU_INSTR = DU_BYTES / 1024;                // one wave instruction of LDS-DMA = 1 KiB
    // x chunk rounded up so that the four waves issue the same number of instructions (one counted wait serves all)
    constexpr int X_INSTR = ((DU_INSTR + (X_ROWS * PITCH + 1023) / 1024 + 3) / 4) * 4 - DU_INSTR;
    constexpr int X_BYTES = X_INSTR * 1024;
    constexpr int NI = (DU_INSTR + X_INSTR) / 4;             // instructions per wave and chunk
    constexpr int BUF = DU_BYTES + X_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.y, tb = blockIdx.x, z = blockIdx.z, Z = gridDim.z;
    const int tap0 = tb * TB;
    const int64_t rows_z = rows / Z, row0 = z * rows_z;
    const int nchunk = (int)(rows_z / CH);
    const uint16_t* dug = du + (int64_t)g * rows * DG;
    const uint16_t* xgg = xg + (int64_t)g * rows * DG;
    const int64_t x_limit = total_elems - ((int64_t)g * rows * DG) - 8;      // last 16-byte piece inside the whole slab buffer

    auto stage = [&](int buf, int c) {
        char* base = smem + buf * BUF;
        const int64_t r = row0 + (int64_t)c * CH;
        for (int i = wave; i < DU_INSTR + X_INSTR; i += 4) {
            if (i < DU_INSTR) {
                glds16_(dug + r * DG + i * 512 + lane * 8, base + i * 1024);
            } else {
                const int j = i - DU_INSTR;
                int64_t e = (r + tap0) * DG + j * 512 + lane * 8;
                e = e < x_limit ? e : x_limit;                     // rows past the buffer meet du = 0 (trailing halo): any finite data
                glds16_(xgg + e, base + DU_BYTES + j * 1024);
            }
        }
    };

    f32x4 acc[TPW][NB][NB];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[j][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane offsets: lane 4 q + pp of a 16-lane group gq supplies row (8 gq + 4 s + q), columns 4 pp .. 4 pp + 3
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int lane_off = (8 * gq + q) * PITCH + pp * 8;
    auto tr = [&](const char* p) -> s16x4_ {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)p);
    };
    auto frag = [&](const char* img, int row, int cb) -> bf16x8 {          // 16 columns cb, reduction rows row .. row + 31
        const char* p = img + row * PITCH + cb * 32 + lane_off;
        const s16x4_ lo = tr(p), hi = tr(p + 4 * PITCH);
        const s16x8_ v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    stage(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) stage(buf ^ 1, c + 1);
        // the chunk staged one iteration ago: everything but the newest loads must have landed
        if (c + 1 < nchunk) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const char* dimg = smem + buf * BUF;
        const char* ximg = dimg + DU_BYTES;
#pragma unroll
        for (int ks = 0; ks < CH / 32; ++ks) {
            bf16x8 af[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) af[a] = frag(dimg, ks * 32, a);
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                bf16x8 bfv[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) bfv[b] = frag(ximg, ks * 32 + wave * TPW + j, b);
#pragma unroll
                for (int a = 0; a < NB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[j][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfv[b], acc[j][a][b], 0, 0, 0);
            }
        }
        __syncthreads();                                     // every wave is done with this buffer before it is staged again
    }
    // part[z][g][co][tap Dg + ci]: a lane holds co = 16 a + 4 (lane / 16) + r, ci = 16 b + lane % 16
    float* out = part + ((int64_t)z * gridDim.y + g) * DG * (int64_t)Kp * DG;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = tap0 + wave * TPW + j;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * a + 4 * (lane >> 4) + r, ci = 16 * b + (lane & 15);
                    out[(int64_t)co * Kp * DG + tap * DG + ci] = acc[j][a][b][r];
                }
    }
}

template <int DG, int TPW>
static int launch_wgrad(const uint16_t* du, const uint16_t* xg, float* part, int G, int64_t rows, int Kp, int Z, hipStream_t s) {
    constexpr int TB = 4 * TPW, CH = 128, PITCH = DG * 2;
    constexpr int DU_INSTR = CH * PITCH / 1024;
    constexpr int X_INSTR = ((DU_INSTR + ((CH + TB) * PITCH + 1023) / 1024 + 3) / 4) * 4 - DU_INSTR;
    constexpr int LDS = 2 * (CH * PITCH + X_INSTR * 1024);
    static sc_lds_attr_once attr;
    if (hipError_t e = sc_set_max_lds_once(attr, posconv_wgrad_kernel<DG, TPW>, LDS); e != hipSuccess) {
        sc_set_error("hipFuncSetAttribute(posconv_wgrad): %s", hipGetErrorString(e));
        return -3;
    }
    hipLaunchKernelGGL((posconv_wgrad_kernel<DG, TPW>), dim3(Kp / TB, G, Z), dim3(256), LDS, s, du, xg, part, rows, Kp,
                       (int64_t)G * rows * DG);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int sc_posconv_wgrad_bf16(const sc_bf16* du, const sc_bf16* xg, float* part, int32_t G, int64_t rows, int32_t Dg, int32_t Kp,
                                     int32_t Z, void* stream) {
    SC_CHECK(du && xg && part, "sc_posconv_wgrad_bf16: null pointer");
    SC_CHECK((Dg == 48 || Dg == 64) && G > 0 && Z > 0 && Kp % 16 == 0 && rows > 0 && rows % ((int64_t)128 * Z) == 0,
             "sc_posconv_wgrad_bf16: Dg in {48, 64}, Kp %% 16 == 0, rows %% (128 Z) == 0 (Dg=%d Kp=%d rows=%lld Z=%d)", Dg, Kp,
             (long long)rows, Z);
    SC_CHECK(((uintptr_t)du % 16) == 0 && ((uintptr_t)xg % 16) == 0 && ((uintptr_t)part % 16) == 0, "sc_posconv_wgrad_bf16: alignment");
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *dp = (const uint16_t*)du, *xp = (const uint16_t*)xg;
    return Dg == 48 ? launch_wgrad<48, 4>(dp, xp, part, G, rows, Kp, Z, s) : launch_wgrad<64, 2>(dp, xp, part, G, rows, Kp, Z, s);
}
