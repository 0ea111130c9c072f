// Probe of `buffer_load_dwordx4 ... offen lds` (LDS-DMA) as the weight-gradient kernel wants to use it:
//   (1) lane l of a wave lands at LDS byte  M0 + 16 * l  (lane-linear image), (2) an M0 above 64 KB addresses the upper LDS,
//   (3) a lane whose buffer offset is out of range writes ZEROS (padding voxels), (4) data is visible after vmcnt(0) + barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t lds_byte) {
    // M0 carries the wave-uniform LDS byte address; written in the same statement that uses it
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_byte), "s"(rs) : "memory", "m0");
}

__global__ __launch_bounds__(256) void probe(const uint32_t* src, uint32_t nbytes, uint32_t* out, uint32_t lds_base) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4; i += 256) ((uint32_t*)lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // wave w copies 1 KB: lane l reads 16 B at element (w * 64 + (63 - l)) * 16 (reversed: the LDS image must follow the LANE, not the address);
    // lanes 5 and 40 of every wave use an out-of-range offset
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
    uint32_t voff = (uint32_t)(wave * 64 + (63 - lane)) * 16u;
    if (lane == 5 || lane == 40) voff = 0xFFFFFFF0u;
    dma16(rs, voff, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 4 * 1024 / 4 + 64; i += 256) out[i] = ((const uint32_t*)(lds + lds_base - 128))[i];   // 128 B before .. 128 B after the image
}

int main() {
    const int n = 4096;
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000000u + i;
    uint32_t *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, (1024 + 64) * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (uint32_t lds_base : {1024u, 60u * 1024u, 100u * 1024u, 150u * 1024u}) {
        hipMemset(out, 0, (1024 + 64) * 4);
        probe<<<1, 256, 160 * 1024>>>(src, n * 4, out, lds_base);
        std::vector<uint32_t> o(1024 + 64);
        hipError_t e = hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, zeros_ok = 0;
        for (int i = 0; i < 32; ++i) bad += o[i] != 0xDEADBEEFu;                  // nothing written before the image
        for (int i = 32 + 1024; i < 64 + 1024; ++i) bad += o[i] != 0xDEADBEEFu;   // ... nor after it
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const uint32_t got = o[32 + (w * 64 + l) * 4 + j];
                    if (l == 5 || l == 40) { if (got == 0) ++zeros_ok; else ++bad; }
                    else bad += got != 0x1000000u + (uint32_t)((w * 64 + (63 - l)) * 4 + j);
                }
        printf("lds_base %6u: %s  mismatches %d, out-of-range lanes wrote zeros in %d / 32 dwords (err %d)  sample o[32..35] = %08x %08x %08x %08x\n", lds_base, bad ? "FAIL" : "ok", bad, zeros_ok,
               (int)e, o[32], o[33], o[34], o[35]);
        bad_total += bad;
    }
    return bad_total != 0;
}
