// Transposed LDS fragment reads shared by the weight-gradient kernels (conv3d_wgrad.hip, conv3d_wgrad_s2.hip).
#pragma once
#include "common.hpp"

typedef short v4s_t __attribute__((__vector_size__(4 * sizeof(short))));

// Fragment = 16 bytes/lane for bf16 (8 k), 8 MFMAs worth of scalars for f32 are loaded on the fly.
// lane part of a fragment address (bytes from the fragment's first row); the rest (tile row, tap) is wave-uniform or static
template <int TR>
__device__ __forceinline__ int frag_lane_off(int pitch, int lane) {
    if (TR) {
        const int q = lane & 15, g = lane >> 4;
        return ((g >> 1) * 8 + (q >> 2)) * pitch + ((g & 1) * 16 + (q & 3) * 4) * 2;
    }
    return ((lane >> 5) * 8) * pitch + (lane & 31) * 2;
}
// a = fragment base INCLUDING the lane part: rows = 16 consecutive voxels (row pitch `pitch`), 32 channels (2 B each).
// result: lane l -> channel l&31, voxels (l>>5)*8 .. +7
template <int TR>
__device__ __forceinline__ uint4 frag_bf16(const char* a, int pitch) {
    uint4 r;
    if (TR) {
        v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a));
        v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a + 4 * pitch));
        union { v4s_t v; uint2 u; } ul, uh;
        ul.v = lo; uh.v = hi;
        r = make_uint4(ul.u.x, ul.u.y, uh.u.x, uh.u.y);
    } else {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = *(const bf16_t*)(a + j * pitch);
        r = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
    return r;
}

