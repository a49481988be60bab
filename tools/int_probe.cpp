// Integer VALU issue-rate probe (diagnostics only): cycles per wave64 instruction for the operations a stateless dropout hash can be
// built from (v_mul_lo_u32 vs the 24-bit multiplies, shifts, xors, v_alignbit rotates), 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_ab/int_probe tools/int_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ void probe(unsigned* out, long long* cyc, int iters) {
    unsigned a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned c = 0x7feb352du, d = 15;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
        if (OP == 0) {
#define MULLO(v) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(c));
            R8(MULLO) R8(MULLO) R8(MULLO) R8(MULLO)
        } else if (OP == 1) {
#define MUL24(v) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v) : "v"(c));
            R8(MUL24) R8(MUL24) R8(MUL24) R8(MUL24)
        } else if (OP == 2) {
#define MAD24(v) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(d));
            R8(MAD24) R8(MAD24) R8(MAD24) R8(MAD24)
        } else if (OP == 3) {
#define XOR(v) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(c));
            R8(XOR) R8(XOR) R8(XOR) R8(XOR)
        } else if (OP == 4) {
#define SHR(v) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v));
            R8(SHR) R8(SHR) R8(SHR) R8(SHR)
        } else if (OP == 5) {
#define ROT(v) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(v) : "v"(d));
            R8(ROT) R8(ROT) R8(ROT) R8(ROT)
        } else if (OP == 6) {
#define MULHI(v) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v) : "v"(c));
            R8(MULHI) R8(MULHI) R8(MULHI) R8(MULHI)
        } else {
#define XSH(v) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v) : "v"(c));
            R8(XSH) R8(XSH) R8(XSH) R8(XSH)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, unsigned* out, long long* cyc) {
    const int iters = 2000, threads = 512;          // 8 waves per CU = 2 per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 2 waves x iters x 32 instructions
    printf("%-16s %.3f ns per wave instruction per SIMD (%.1f us for %d)\n", name, ms * 1e6 / (2.0 * iters * 32), ms * 1e3, 2 * iters * 32);
}

int main() {
    unsigned* out;
    long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&cyc, 256 * 8));
    run<0>("v_mul_lo_u32", out, cyc);
    run<1>("v_mul_u32_u24", out, cyc);
    run<2>("v_mad_u32_u24", out, cyc);
    run<3>("v_xor_b32", out, cyc);
    run<4>("v_lshrrev_b32", out, cyc);
    run<5>("v_alignbit_b32", out, cyc);
    run<6>("v_mul_hi_u32", out, cyc);
    run<7>("v_lshl_add_u32", out, cyc);
    return 0;
}
