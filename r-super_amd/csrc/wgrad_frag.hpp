// Transposed LDS fragment reads shared by the weight-gradient kernels (conv3d_wgrad.hip, conv3d_wgrad_s2.hip).
#pragma once
#include "common.hpp"

typedef short v4s_t __attribute__((__vector_size__(4 * sizeof(short))));

// Block -> (input-channel chunk bx, row group by, K split bz) and the split's tiles, XCD-aware.  The hardware hands linear workgroup id L = x + gx (y + gy z)
// to XCD L % 8 (private L2 each).  The blocks are renumbered V = (blocks of lower XCDs) + L / 8, so that consecutive V share an XCD, and V is decoded
// chunk-fastest: the gx * gy SIBLINGS of a split -- they sweep the same tiles and read the same dY / x rows -- run on ONE XCD and fetch those rows into its
// L2 once (with the plain decoding a 3-chunk layer put them on three XCDs: three fabric reads of every dY tile), and the splits of an XCD form a CLASS
// that owns a contiguous range of tiles and sweeps it together (neighbouring tiles share halo rows).  Any placement gives the same sums per slab set;
// only the assignment of tiles to slabs changes.  Fewer than 8 splits or 64 tiles: plain grid-stride order.
struct WgBlockMap { int bx, by, bz, tile0, tile_end, tstride; };
__device__ __forceinline__ WgBlockMap wg_block_map(int tiles, int S) {
    const unsigned gx = gridDim.x, gy = gridDim.y, sib = gx * gy;
    WgBlockMap m;
    m.bx = (int)blockIdx.x; m.by = (int)blockIdx.y; m.bz = (int)blockIdx.z;
    if (S >= 8 && tiles >= 64) {
        const unsigned T = sib * (unsigned)S, L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned q = T >> 3, r = T & 7;
        auto prefix = [&](unsigned x) { return x * q + (x < r ? x : r); };      // blocks on the XCDs below x
        const unsigned V = prefix(L & 7) + (L >> 3);
        m.bx = (int)(V % gx); m.by = (int)((V / gx) % gy); m.bz = (int)(V / sib);
        const unsigned V0 = (unsigned)m.bz * sib;                                  // class of a split = the XCD of its first sibling
        unsigned c = 0;
#pragma unroll
        for (unsigned x = 1; x < 8; ++x) c = V0 >= prefix(x) ? x : c;
        const unsigned zlo = (prefix(c) + sib - 1) / sib, zhi = (prefix(c + 1) + sib - 1) / sib;   // splits [zlo, zhi) form class c
        m.tile0 = (int)((long)tiles * zlo / S) + (m.bz - (int)zlo);
        m.tile_end = (int)((long)tiles * zhi / S);
        m.tstride = (int)(zhi - zlo);
    } else { m.tile0 = m.bz; m.tile_end = tiles; m.tstride = S; }
    return m;
}

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

