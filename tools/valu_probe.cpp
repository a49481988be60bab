// VALU issue-rate probe (diagnostics only): cycles per wave64 instruction on one SIMD for v_fma_f32, v_pk_fma_f32, v_rcp_f32, v_exp_f32
// and v_cvt_pk_bf16_f32, at 1, 2 and 4 resident waves per SIMD.  Cycles from s_memtime inside the kernel (independent of the clock).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_ab/valu_probe tools/valu_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP>
__global__ void probe(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0001f, d = 1e-4f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
#define FMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(d));
            R8(FMA) R8(FMA) R8(FMA) R8(FMA)
        } else if (OP == 1) {
#define P8(X) X(p0) X(p1) X(p2) X(p3) X(p4) X(p5) X(p6) X(p7)
#define PFMA(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c2), "v"(d2));
            P8(PFMA) P8(PFMA) P8(PFMA) P8(PFMA)
        } else if (OP == 2) {
#define RCP(v) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
            R8(RCP) R8(RCP) R8(RCP) R8(RCP)
        } else if (OP == 3) {
#define EXP(v) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
            R8(EXP) R8(EXP) R8(EXP) R8(EXP)
        } else if (OP == 4) {
#define PMUL(v) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(c2));
            P8(PMUL) P8(PMUL) P8(PMUL) P8(PMUL)
        } else {
#define CVT(v) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(c));
            R8(CVT) R8(CVT) R8(CVT) R8(CVT)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU: waves spread over the 4 SIMDs
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);      // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[256];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < 256; ++i) s += (double)h[i];
        const double per_instr_wave = s / 256 / (iters * 32.0);
        printf("%-22s %d wave(s)/SIMD: %.2f s_memtime ticks per instruction per wave (%.2f per SIMD); wall %.1f us -> %.2f ns per instruction per SIMD, "
               "tick = %.3f ns\n", name, waves_per_simd, per_instr_wave, per_instr_wave / waves_per_simd, ms * 1e3,
               ms * 1e6 / (iters * 32.0 * waves_per_simd), ms * 1e6 / (s / 256));
    }
}

int main() {
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * sizeof(long long)));
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<4>("v_pk_mul_f32", out, cyc);
    run<2>("v_rcp_f32", out, cyc);
    run<3>("v_exp_f32", out, cyc);
    run<5>("v_cvt_pk_bf16_f32", out, cyc);
    return 0;
}
