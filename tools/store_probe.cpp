// Store-path probe for the GEMM epilogue (diagnostics only; not part of the library): what does one `global_store_dwordx4`
// wave-instruction cost a CU, as a function of how the 64 lanes' 16-byte pieces fall on 128-byte lines?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/store_probe.cpp && /tmp/store_probe
// One 512-thread workgroup per CU; every wave writes its 128 x 64 bf16 block (16 KiB, 16 instructions) of a 256 x 256 tile of a
// row-major [M, N] matrix, tile after tile (persistent walk), nothing else - the store phase of gemm256's epilogue in isolation.
// Patterns (rows x bytes covered by ONE wave-instruction):
//   0: 16 rows x 64 B   (register-direct epilogue: 4 lanes per row)
//   1:  8 rows x 128 B  (full lines: 8 lanes per row)
//   2: 64 rows x 16 B   (row per lane)
//   3: 32 rows x 32 B   (2 lanes per row)
//   4:  4 rows x 256 B  (two full lines per row; needs a 128-column wave block -> wave = 64 rows x 128 cols)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT, int NT>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int M, int N, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nN = N / 256;
    const u32x4 v = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + wave};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = t * gridDim.x + blockIdx.x;
        const int m0 = (tile / nN) * 256, n0 = (tile % nN) * 256;
        if (m0 >= M) break;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int row, colb;                      // row inside the wave's block, byte offset inside the block row
            int bm = wm * 128, bn = wn * 64;     // block origin
            if (PAT == 0) { row = (i >> 1) * 16 + (lane & 15); colb = (i & 1) * 64 + (lane >> 4) * 16; }
            else if (PAT == 1) { row = i * 8 + (lane >> 3); colb = (lane & 7) * 16; }
            else if (PAT == 2) { row = (i >> 3) * 64 + lane; colb = (i & 7) * 16; }
            else if (PAT == 3) { row = (i >> 2) * 32 + (lane >> 1); colb = (i & 3) * 32 + (lane & 1) * 16; }
            else { bm = (wave >> 1) * 64; bn = (wave & 1) * 128; row = i * 4 + (lane >> 4); colb = (lane & 15) * 16; }
            unsigned short* dst = C + (size_t)(m0 + bm + row) * N + n0 + bn + colb / 2;
            if (NT) __builtin_nontemporal_store(v, (u32x4*)dst);
            else *(u32x4*)dst = v;
        }
    }
}

template <int PAT, int NT>
static void run(unsigned short* C, int M, int N, const char* name, int cus = 256) {
    const int tiles = (M / 256) * (N / 256);
    const int per = (tiles + cus - 1) / cus;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((store_kernel<PAT, NT>), dim3(cus), dim3(512), 0, 0, C, M, N, per);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = (double)M * N * 2;
    const double instr_per_cu = bytes / 1024.0 / cus;
    printf("%-34s CUs=%d M=%d N=%d  %.1f us  %.2f TB/s  %.1f B/clk/CU@2.1GHz  %.0f cyc/instr/CU@2.1GHz\n", name, cus, M, N, best * 1e3, bytes / best / 1e9,
           bytes / cus / (best * 1e-3 * 2.1e9), best * 1e-3 * 2.1e9 / instr_per_cu);
}

int main() {
    {   // a 2.1 GB output (conv layer 0's): does the write rate hold beyond the Infinity Cache / TLB reach?
        unsigned short* C;
        const int Mb = 2048000, Nb = 512;
        CK(hipMalloc(&C, (size_t)Mb * Nb * 2));
        CK(hipMemset(C, 0, (size_t)Mb * Nb * 2));
        run<1, 0>(C, Mb, Nb, "8 rows x 128 B (full lines)");
        run<1, 1>(C, Mb, Nb, "8 rows x 128 B, nontemporal");
        run<0, 0>(C, Mb, Nb, "16 rows x 64 B");
        CK(hipFree(C));
    }
    const int M = 32768;
    for (int N : {768, 2304, 3072}) {
        unsigned short* C;
        CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        run<0, 0>(C, M, N, "16 rows x 64 B");
        run<1, 0>(C, M, N, "8 rows x 128 B (full lines)");
        run<2, 0>(C, M, N, "64 rows x 16 B (row per lane)");
        run<3, 0>(C, M, N, "32 rows x 32 B");
        run<4, 0>(C, M, N, "4 rows x 256 B");
        run<0, 1>(C, M, N, "16 rows x 64 B, nontemporal");
        run<1, 1>(C, M, N, "8 rows x 128 B, nontemporal");
        if (N == 2304)
            for (int cus : {8, 32, 64, 128}) {       // fewer storing CUs (blocks 0..cus-1: spread over the 8 XCDs): is the limit per CU or chip-wide?
                run<0, 0>(C, M / (256 / cus), N, "16 rows x 64 B", cus);
                run<1, 0>(C, M / (256 / cus), N, "8 rows x 128 B (full lines)", cus);
            }
        CK(hipFree(C));
    }
    return 0;
}
