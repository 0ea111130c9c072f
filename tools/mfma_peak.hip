// Sustained dense bf16 MFMA rate of this GPU (ceiling calibration for roofline fractions).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int wpb, int bpc, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * bpc), block(64 * wpb);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * 32 * 32 * 16 * NACC * (double)iters * grid.x * wpb;
        printf("NACC %d waves/block %d blocks/CU %d iters %d: %.3f ms  %.1f TFLOP/s\n", NACC, wpb, bpc, iters, ms, fl / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<4>(4, 1, 20000);     // 1 wave / SIMD, short (~0.5 ms)
    run<4>(4, 1, 200000);    // sustained (~5 ms)
    run<4>(4, 2, 100000);    // 2 waves / SIMD
    run<8>(4, 1, 100000);
    return 0;
}
