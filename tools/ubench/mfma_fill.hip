// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD when NACC accumulators are used round-robin
// and NF independent VALU fillers (v_fma_f32) sit between consecutive MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC, int NF, int LDSR>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ uint4 lds[4096];
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    union { uint4 u; bf16x8_t v; } A, B;
    A.u = make_uint4(threadIdx.x, 1, 2, 3); B.u = make_uint4(5, threadIdx.x, 7, 8);
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = threadIdx.x * 0.5f + j;
    lds[threadIdx.x] = A.u; lds[threadIdx.x + 256] = B.u;
    __syncthreads();
    const uint4* lp = lds + (threadIdx.x & 255);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (LDSR && (u % LDSR) == 0) { B.u = lp[(u & 7) * 256]; }
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j % 8]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int j = 0; j < 8; ++j) s += f[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NF, int LDSR>
void run(float* d) {
    const int iters = 2000;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<NACC, NF, LDSR>), dim3(256), dim3(256), 0, 0, d, 10);
    hipEventRecord(s);
    hipLaunchKernelGGL((k<NACC, NF, LDSR>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double mf = (double)iters * 32;
    printf("NACC %d NF %2d LDSR %d : %7.1f ns per MFMA  (%.2f PF chip)\n", NACC, NF, LDSR, ms * 1e6 / mf, 256.0 * 4 * 32768 * mf / (ms * 1e-3) / 1e15);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<1, 0, 0>(d); run<2, 0, 0>(d); run<4, 0, 0>(d);
    run<1, 2, 0>(d); run<2, 2, 0>(d); run<4, 2, 0>(d);
    run<1, 4, 0>(d); run<2, 4, 0>(d); run<4, 4, 0>(d); run<8, 4, 0>(d);
    run<2, 6, 0>(d); run<4, 6, 0>(d);
    run<2, 8, 0>(d); run<4, 8, 0>(d);
    run<2, 12, 0>(d); run<4, 12, 0>(d);
    run<2, 0, 1>(d); run<4, 0, 1>(d); run<2, 4, 1>(d); run<4, 4, 1>(d); run<2, 4, 2>(d); run<4, 6, 2>(d);
    return 0;
}
