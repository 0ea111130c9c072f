// Micro-benchmark: MFMA rate of a wgrad-like inner loop: every MFMA consumes one B fragment fetched from LDS (two
// ds_read_b64_tr_b16, or one ds_read_b128) RD units ahead; NW waves per block (4 = one per SIMD, 8 = two per SIMD), NACC accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short v4s_t __attribute__((__vector_size__(4 * sizeof(short))));

template <int MODE>   // 0: two tr reads, 1: one b128 read, 2: no reads
__device__ __forceinline__ uint4 frag(const char* a) {
    if (MODE == 0) {
        v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a));
        v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a + 256));
        union { v4s_t v; uint2 u; } ul, uh; ul.v = lo; uh.v = hi;
        return make_uint4(ul.u.x, ul.u.y, uh.u.x, uh.u.y);
    }
    if (MODE == 1) return *(const uint4*)a;
    return make_uint4(1, 2, 3, 4);
}

template <int NW, int NACC, int RD, int MODE, int PER>   // PER: MFMAs per fetched fragment (operand re-use)
__global__ __launch_bounds__(64 * NW, NW / 4) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 64 * NW) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    union { uint4 u; bf16x8_t v; } A, B;
    A.u = make_uint4(threadIdx.x, 1, 2, 3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = lds + ((lane & 15) >> 2) * 64 + (lane >> 4) * 32 + (lane & 3) * 8 + wave * 1024;   // tr16 lane pattern-ish, conflict-free rows
    const char* base128 = lds + (lane & 31) * 80 + (lane >> 5) * 16 + wave * 4096;
    constexpr int NU = 56;
    uint4 ring[RD + 1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < RD; ++u) ring[u] = frag<MODE>((MODE == 0 ? base : base128) + (u % 16) * 1536);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + RD < NU) ring[(u + RD) % (RD + 1)] = frag<MODE>((MODE == 0 ? base : base128) + ((u + RD) % 16) * 1536);
            __builtin_amdgcn_sched_barrier(0);
            B.u = ring[u % (RD + 1)];
#pragma unroll
            for (int r = 0; r < PER; ++r) acc[(u * PER + r) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, acc[(u * PER + r) % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}

template <int NW, int NACC, int RD, int MODE, int PER>
void run(float* d) {
    const int iters = 400;
    auto kk = k<NW, NACC, RD, MODE, PER>;
    hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(kk, dim3(256), dim3(64 * NW), 64 * 1024, 0, d, 4);
    hipEventRecord(s);
    hipLaunchKernelGGL(kk, dim3(256), dim3(64 * NW), 64 * 1024, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double mf = (double)iters * 56 * PER * NW;                       // MFMAs per CU
    printf("waves %d acc %d dist %d %s x%d MFMA/frag : %6.2f PF chip (%.1f ns per MFMA per SIMD)\n", NW, NACC, RD, MODE == 0 ? "2x tr16_b64" : MODE == 1 ? "1x b128   " : "no LDS    ", PER,
           256.0 * 32768 * mf / (ms * 1e-3) / 1e15, ms * 1e6 / (mf / 4));
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<4, 7, 3, 2, 1>(d); run<8, 7, 3, 2, 1>(d);
    run<4, 7, 3, 0, 1>(d); run<8, 7, 3, 0, 1>(d); run<8, 7, 6, 0, 1>(d); run<8, 7, 2, 0, 1>(d);
    run<4, 7, 3, 1, 1>(d); run<8, 7, 3, 1, 1>(d);
    run<8, 7, 3, 0, 2>(d); run<8, 7, 3, 0, 3>(d); run<4, 7, 3, 0, 2>(d); run<4, 7, 3, 0, 3>(d);
    run<8, 7, 3, 1, 2>(d); run<4, 7, 3, 1, 2>(d);
    return 0;
}
