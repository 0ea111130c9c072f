// Byte/integer kernels of the report-supervision losses (gfx950, HBM/LDS-bound; no float conv):
//   dilate_pass      one ball-dilation pass (k <= 7) on 0/1 byte volumes, 4 voxels per lane as one u32
//                    (dilate_volume_conv, training/losses_foundation.py:50-99; ball of create_ball_kernel :1161)
//   ball_conv_argmax Gaussian-ball correlation + first-maximum argmax   (isolate_tumor :1423-1445)
//   insert_ball      binary ball pasted at a centre, clipped          (insert_ball :1336-1385)
//   radix histogram / select   exact top-k mask, ties by lower index  (torch.topk use at :1483-1492)
//   rank weights     GWRP rank weights on the pseudo mask              (GlobalWeightedRankPooling :442-535)
#include "common.hpp"
#include "misc.hpp"
#include "morph.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ dilation
// 4*(dz^2+dy^2+dx^2) <= k^2  <=>  inside the ball of diameter k (k odd).  For a (dz,dy) row the half width is
// L = floor(sqrt(k^2/4 - dz^2 - dy^2)).
__device__ __forceinline__ int row_halfwidth(int k, int dz, int dy) {
    const int rem4 = k * k - 4 * (dz * dz + dy * dy);
    if (rem4 < 0) return -1;
    int L = 0;
    while (4 * (L + 1) * (L + 1) <= rem4) ++L;
    return L;
}

__global__ __launch_bounds__(256) void dilate_pass_kernel(const uint8_t* in, uint8_t* out, const uint8_t* flags, long nvol, int D, int H, int W, int k) {
    const int W4 = W >> 2;                                       // W % 4 == 0 (checked on the host)
    const long words = (long)nvol * D * H * W4;
    const int r = k >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % W4);
        long t = i / W4;
        const int y = (int)(t % H); t /= H;
        const int z = (int)(t % D);
        const long v = t / D;
        if (flags && !flags[v]) { ((uint32_t*)out)[i] = 0u; continue; }     // all-zero volume: its dilation is zero, nothing to read
        const uint32_t* vol = (const uint32_t*)(in + v * (long)D * H * W);
        uint32_t acc = 0;
        for (int dz = -r; dz <= r; ++dz) {
            const int zz = z + dz;
            if (zz < 0 || zz >= D) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                const int L = row_halfwidth(k, dz, dy);
                if (L < 0) continue;
                const uint32_t* row = vol + ((long)zz * H + yy) * W4;
                const uint32_t cur = row[x4];
                uint32_t o = cur;
                if (L > 0) {
                    const uint32_t prev = x4 > 0 ? row[x4 - 1] : 0u, next = x4 + 1 < W4 ? row[x4 + 1] : 0u;
                    for (int s = 1; s <= L; ++s) {
                        o |= __builtin_amdgcn_alignbyte(cur, prev, 4 - s);    // bytes x-s .. x-s+3
                        o |= __builtin_amdgcn_alignbyte(next, cur, s);        // bytes x+s .. x+s+3
                    }
                }
                acc |= o;
            }
        }
        ((uint32_t*)out)[i] = acc;
    }
}

// 16 voxels per lane (W % 16 == 0): one 16-byte load per row + the two neighbouring words.
__global__ __launch_bounds__(256) void dilate_pass16_kernel(const uint8_t* in, uint8_t* out, const uint8_t* flags, long nvol, int D, int H, int W, int k) {
    const int W16 = W >> 4;
    const long items = (long)nvol * D * H * W16;
    const int r = k >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        const int x16 = (int)(i % W16);
        long t = i / W16;
        const int y = (int)(t % H); t /= H;
        const int z = (int)(t % D);
        const long v = t / D;
        uint4 acc = make_uint4(0, 0, 0, 0);
        if (flags && !flags[v]) { ((uint4*)out)[i] = acc; continue; }       // all-zero volume (most label planes carry no unknown voxels)
        const uint8_t* vol = in + v * (long)D * H * W;
        for (int dz = -r; dz <= r; ++dz) {
            const int zz = z + dz;
            if (zz < 0 || zz >= D) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                const int L = row_halfwidth(k, dz, dy);
                if (L < 0) continue;
                const uint4* row = (const uint4*)(vol + ((long)zz * H + yy) * W);
                const uint4 c = row[x16];
                uint4 o = c;
                if (L > 0) {
                    const uint32_t pv = x16 > 0 ? ((const uint32_t*)(row + x16))[-1] : 0u;
                    const uint32_t nx = x16 + 1 < W16 ? ((const uint32_t*)(row + x16 + 1))[0] : 0u;
                    for (int s = 1; s <= L; ++s) {
                        o.x |= __builtin_amdgcn_alignbyte(c.x, pv, 4 - s) | __builtin_amdgcn_alignbyte(c.y, c.x, s);
                        o.y |= __builtin_amdgcn_alignbyte(c.y, c.x, 4 - s) | __builtin_amdgcn_alignbyte(c.z, c.y, s);
                        o.z |= __builtin_amdgcn_alignbyte(c.z, c.y, 4 - s) | __builtin_amdgcn_alignbyte(c.w, c.z, s);
                        o.w |= __builtin_amdgcn_alignbyte(c.w, c.z, 4 - s) | __builtin_amdgcn_alignbyte(nx, c.w, s);
                    }
                }
                acc.x |= o.x; acc.y |= o.y; acc.z |= o.z; acc.w |= o.w;
            }
        }
        ((uint4*)out)[i] = acc;
    }
}

// generic-width fallback (one voxel per lane)
__global__ __launch_bounds__(256) void dilate_pass_scalar_kernel(const uint8_t* in, uint8_t* out, const uint8_t* flags, long nvol, int D, int H, int W, int k) {
    const long total = (long)nvol * D * H * W;
    const int r = k >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        long t = i / W;
        const int y = (int)(t % H); t /= H;
        const int z = (int)(t % D);
        if (flags && !flags[t / D]) { out[i] = 0; continue; }
        const uint8_t* vol = in + (t / D) * (long)D * H * W;
        uint8_t acc = 0;
        for (int dz = -r; dz <= r && !acc; ++dz)
            for (int dy = -r; dy <= r && !acc; ++dy) {
                const int L = row_halfwidth(k, dz, dy);
                const int zz = z + dz, yy = y + dy;
                if (L < 0 || zz < 0 || zz >= D || yy < 0 || yy >= H) continue;
                for (int dx = -L; dx <= L; ++dx) {
                    const int xx = x + dx;
                    if (xx >= 0 && xx < W && vol[((long)zz * H + yy) * W + xx]) { acc = 1; break; }
                }
            }
        out[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ ball correlation + argmax
// out(z,y,x) = sum over the ball of diameter d of exp(-|o|^2/(2 std^2)) * x(z+oz, y+oy, x+ox); the reference
// normalises the kernel to sum 1, a positive scale that does not move the argmax.  key = (value bits << 32) |
// ~index so a 64-bit atomicMax keeps the FIRST maximum (torch.argmax), values are >= 0.
__global__ __launch_bounds__(256) void ball_conv_argmax_kernel(const float* x, int D, int H, int W, int d_odd, float inv2s2,
                                                               unsigned long long* best, float* conv_out) {
    __shared__ float g1[64];
    __shared__ unsigned long long wbest[4];
    const int R = d_odd >> 1;                                    // integer offsets with |o| <= d/2
    for (int i = threadIdx.x; i <= R && i < 64; i += 256) g1[i] = expf(-(float)(i * i) * inv2s2);
    __syncthreads();
    const long V = (long)D * H * W;
    unsigned long long mine = 0ull;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H), zz = (int)(i / ((long)W * H));
        float acc = 0.f;
        for (int dz = -R; dz <= R; ++dz) {
            const int z = zz + dz;
            if (z < 0 || z >= D) continue;
            for (int dy = -R; dy <= R; ++dy) {
                const int y = yy + dy;
                if (y < 0 || y >= H) continue;
                const int L = row_halfwidth(d_odd, dz, dy);
                if (L < 0) continue;
                const float gzy = g1[dz < 0 ? -dz : dz] * g1[dy < 0 ? -dy : dy];
                const float* row = x + ((long)z * H + y) * W;
                const int lo = max(-L, -xx), hi = min(L, W - 1 - xx);
                float rs = 0.f;
                for (int dx = lo; dx <= hi; ++dx) rs += g1[dx < 0 ? -dx : dx] * row[xx + dx];
                acc += gzy * rs;
            }
        }
        if (conv_out) conv_out[i] = acc;
        const unsigned long long key = ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
        mine = key > mine ? key : mine;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mine, o, 64);
        mine = other > mine ? other : mine;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = wbest[0];
        for (int i = 1; i < 4; ++i) b = wbest[i] > b ? wbest[i] : b;
        atomicMax(best, b);
    }
}

// Two-stage form of the same correlation, O(k^2) instead of O(k^3) per voxel: the Gaussian factorises per axis and a
// (dz, dy) row of the ball is the x interval |dx| <= L(dz, dy), so
//     out(z,y,x) = sum_{dz,dy} g(dz) g(dy) * F_L(dz,dy)(z+dz, y+dy, x),   F_L(z,y,x) = sum_{|dx|<=L} g(dx) x(z,y,x+dx).
// Stage 1 builds the R+1 running row sums F_0..F_R (F_L = F_{L-1} + g(L) (x[x-L] + x[x+L]), two MACs per L) into the
// workspace ws[(R+1)][V]; stage 2 gathers k^2 of them per voxel (coalesced along x) and keeps the packed argmax key.
__global__ __launch_bounds__(256) void ball_rowsum_kernel(const float* __restrict__ x, int D, int H, int W, int R, float inv2s2, float* __restrict__ ws) {
    const long V = (long)D * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const float* row = x + (i - xx);
        float f = row[xx];
        ws[i] = f;
        for (int L = 1; L <= R; ++L) {
            const float g = expf(-(float)(L * L) * inv2s2);
            const float a = xx - L >= 0 ? row[xx - L] : 0.f, b = xx + L < W ? row[xx + L] : 0.f;
            f += g * (a + b);
            ws[(long)L * V + i] = f;
        }
    }
}

// One bit per (z, y) row of x: set when the row holds a non-zero.  x = sigmoid(logit) * segment mask is zero outside the report's organ
// segment, F_L of an all-zero row is all zero, and adding exact zeros changes no partial sum: stage 2 visits the set bits only.
// Block = one z plane; a wave ballots one row at a time.
__global__ __launch_bounds__(256) void ball_row_bits_kernel(const float* __restrict__ x, int H, int W, int HW, uint32_t* __restrict__ bits) {
    __shared__ uint32_t wb[64];
    const int z = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < HW; i += 256) wb[i] = 0u;
    __syncthreads();
    for (int y = wv; y < H; y += 4) {
        const float* row = x + ((long)z * H + y) * W;
        bool nz = false;
        for (int xx = lane; xx < W; xx += 64) nz |= row[xx] != 0.f;
        if (__ballot(nz) != 0ull && lane == 0) atomicOr(&wb[y >> 5], 1u << (y & 31));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW; i += 256) bits[(long)z * HW + i] = wb[i];
}

__global__ __launch_bounds__(256) void ball_gather_argmax_kernel(const float* __restrict__ ws, const uint32_t* __restrict__ bits, int HW, int D, int H, int W,
                                                                 int d_odd, float inv2s2, unsigned long long* best, float* conv_out) {
    __shared__ float g1[64];
    __shared__ signed char Lt[64 * 64];                          // row half width per (|dz|, |dy|), -1 outside the ball
    __shared__ unsigned long long wbest[4];
    const int R = d_odd >> 1;
    for (int i = threadIdx.x; i <= R; i += 256) g1[i] = expf(-(float)(i * i) * inv2s2);
    for (int i = threadIdx.x; i < (R + 1) * (R + 1); i += 256) Lt[(i / (R + 1)) * 64 + i % (R + 1)] = (signed char)row_halfwidth(d_odd, i / (R + 1), i % (R + 1));
    __syncthreads();
    const long V = (long)D * H * W;
    unsigned long long mine = 0ull;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H), zz = (int)(i / ((long)W * H));
        const int ylo = max(yy - R, 0), yhi = min(yy + R, H - 1);
        float acc = 0.f;
        for (int dz = -R; dz <= R; ++dz) {
            const int z = zz + dz;
            if (z < 0 || z >= D) continue;
            const int az = dz < 0 ? -dz : dz;
            float part = 0.f;
            // rows y in [ylo, yhi] of plane z in ascending order (the order of the dense loop), non-empty ones only
            for (int w = ylo >> 5; w <= (yhi >> 5); ++w) {
                uint32_t m = bits[(long)z * HW + w];
                const int base = w << 5;
                if (base < ylo) m &= 0xFFFFFFFFu << (ylo - base);
                if (base + 31 > yhi) m &= 0xFFFFFFFFu >> (base + 31 - yhi);
                while (m) {
                    const int y = base + __builtin_ctz(m);
                    m &= m - 1;
                    const int ay = y < yy ? yy - y : y - yy;
                    const int L = Lt[az * 64 + ay];
                    if (L < 0) continue;
                    part += g1[ay] * ws[(long)L * V + ((long)z * H + y) * W + xx];
                }
            }
            acc += g1[az] * part;
        }
        if (conv_out) conv_out[i] = acc;
        const unsigned long long key = ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
        mine = key > mine ? key : mine;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mine, o, 64);
        mine = other > mine ? other : mine;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = wbest[0];
        for (int i = 1; i < 4; ++i) b = wbest[i] > b ? wbest[i] : b;
        atomicMax(best, b);
    }
}

// The same gather with a wave per (z, y) row (W <= 64 * NX; lane l owns x = l, l + 64, ...): the walk over the occupancy bits, the half-width table
// and the row addresses is identical for every x of a row, so it runs once per row on the scalar unit (s_load of the bit words, s_ff1 over the set
// bits) and the lanes only issue the coalesced F_L loads of non-empty rows; a row whose whole (z, y) neighbourhood is empty costs 2R+1 scalar
// iterations and no vector memory traffic.  Per-voxel summation order is that of the kernel above (dz outer, rows ascending): same bits.
template <int NX>
__global__ __launch_bounds__(256) void ball_gather_rows_kernel(const float* __restrict__ ws, const uint32_t* __restrict__ bits, int HW, int D, int H, int W,
                                                               int d_odd, float inv2s2, unsigned long long* best, float* conv_out) {
    __shared__ float g1[64];
    __shared__ signed char Lt[64 * 64];
    __shared__ int ymax[64];                                     // per |dz|: the largest |dy| with a row inside the ball (L >= 0 is monotone in |dy|)
    __shared__ unsigned long long wbest[4];
    const int R = d_odd >> 1;
    for (int i = threadIdx.x; i < 64; i += 256) g1[i] = i <= R ? expf(-(float)(i * i) * inv2s2) : 0.f;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) Lt[i] = (i >> 6) <= R && (i & 63) <= R ? (signed char)row_halfwidth(d_odd, i >> 6, i & 63) : (signed char)-1;
    __syncthreads();
    for (int i = threadIdx.x; i < 64; i += 256) {
        int m = -1;
        for (int a = 0; a <= R && i <= R; ++a) m = Lt[i * 64 + a] >= 0 ? a : m;
        ymax[i] = m;
    }
    __syncthreads();
    const long V = (long)D * H * W;
    const int lane = threadIdx.x & 63, nrows = D * H;
    const float gv = g1[lane];                                   // lane a holds g(a): the per-row factor comes from v_readlane, not from memory
    const int ymv = ymax[lane];
    // the workspace as a buffer resource (the host checks (R + 1) * V * 4 < 2^31): a row's load is ONE instruction with the row's byte offset in an
    // SGPR -- the scalar unit is one per CU, and ~45 scalar instructions per gathered row (64-bit address products) were this kernel's whole cost
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, (int)((long)(R + 1) * V * 4), 0x00020000);
    const uint32_t W4 = (uint32_t)W * 4u, HW4 = (uint32_t)H * W4, V4 = (uint32_t)V * 4u;
    unsigned long long mine = 0ull;
    for (int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); row < nrows; row += gridDim.x * 4) {
        const int zz = row / H, yy = row - zz * H;
        float acc[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) acc[q] = 0.f;
        for (int dz = -R; dz <= R; ++dz) {
            const int z = zz + dz;
            if (z < 0 || z >= D) continue;
            const int az = dz < 0 ? -dz : dz;
            const int rng = __builtin_amdgcn_readlane(ymv, az);
            if (rng < 0) continue;
            const int ylo = max(yy - rng, 0), yhi = min(yy + rng, H - 1);
            const uint32_t Lv = (uint32_t)max((int)Lt[az * 64 + lane], 0) * V4;      // lane a holds the byte offset of plane F_L(|dz|, a)
            const uint32_t zb = (uint32_t)z * HW4;
            float part[NX];
#pragma unroll
            for (int q = 0; q < NX; ++q) part[q] = 0.f;
            bool any = false;
            for (int w = ylo >> 5; w <= (yhi >> 5); ++w) {
                uint32_t m = bits[(long)z * HW + w];
                const int base = w << 5;
                if (base < ylo) m &= 0xFFFFFFFFu << (ylo - base);
                if (base + 31 > yhi) m &= 0xFFFFFFFFu >> (base + 31 - yhi);
                any |= m != 0u;
                // four rows per round, their loads in flight together (one row at a time is a chain of dependent HBM/L2 latencies: the kernel's whole cost).
                // A round past the last set bit re-reads row yy of the plane with factor 0: part + 0 * v = part exactly (v is finite).
                while (m) {
                    float g[4], v[4][NX];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool live = m != 0u;
                        const int y = live ? base + __builtin_ctz(m) : yy;
                        m &= m - 1u;                              // 0 stays 0
                        const int ay = y < yy ? yy - y : y - yy;
                        const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)Lv, ay) + zb + (uint32_t)y * W4;
                        const float gy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), ay));
                        g[k] = live ? gy : 0.f;
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            const int xx = lane + 64 * q;          // a lane past the row end reads the start of the next row (finite, unused)
                            v[k][q] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)xx * 4u, so, 0));
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < NX; ++q) part[q] += g[k] * v[k][q];
                }
            }
            if (any) {                  // an untouched part is +0: adding g * 0 leaves acc as it is (acc is never -0)
                const float gz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), az));
#pragma unroll
                for (int q = 0; q < NX; ++q) acc[q] += gz * part[q];
            }
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int xx = lane + 64 * q;
            if (xx < W) {
                const long i = (long)row * W + xx;
                if (conv_out) conv_out[i] = acc[q];
                const unsigned long long key = ((unsigned long long)__float_as_uint(acc[q]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
                mine = key > mine ? key : mine;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mine, o, 64);
        mine = other > mine ? other : mine;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = wbest[0];
        for (int i = 1; i < 4; ++i) b = wbest[i] > b ? wbest[i] : b;
        atomicMax(best, b);
    }
}

// binary ball of (odd) diameter d_odd centred at (cz,cy,cx); count of set voxels accumulated into *count
__global__ void insert_ball_kernel(uint8_t* out, int D, int H, int W, int cz, int cy, int cx, int d_odd, int half, unsigned int* count) {
    const long V = (long)D * H * W;
    unsigned int c = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W) - cx, y = (int)((i / W) % H) - cy, z = (int)(i / ((long)W * H)) - cz;
        // inside the pasted kernel cube (edge 2*half+1) and inside the ball (4*|o|^2 <= d^2)
        const bool in = abs(x) <= half && abs(y) <= half && abs(z) <= half && 4 * (x * x + y * y + z * z) <= d_odd * d_odd;
        out[i] = in ? 1 : 0;
        c += in;
    }
    c = (unsigned int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// the same ball, its centre decoded on the device from the argmax key of ball_conv_argmax (no host round trip between the two)
__global__ void insert_ball_at_kernel(uint8_t* out, int D, int H, int W, const unsigned long long* best, int d_odd, int half, unsigned int* count) {
    const long V = (long)D * H * W;
    const unsigned int idx = 0xFFFFFFFFu - (unsigned int)(*best & 0xFFFFFFFFull);
    const int cx = (int)(idx % (unsigned)W), cy = (int)((idx / (unsigned)W) % (unsigned)H), cz = (int)(idx / ((unsigned)W * (unsigned)H));
    unsigned int c = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W) - cx, y = (int)((i / W) % H) - cy, z = (int)(i / ((long)W * H)) - cz;
        const bool in = abs(x) <= half && abs(y) <= half && abs(z) <= half && 4 * (x * x + y * y + z * z) <= d_odd * d_odd;
        out[i] = in ? 1 : 0;
        c += in;
    }
    c = (unsigned int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// ------------------------------------------------------------------------------------------------ exact top-k (radix select)
// values are non-negative floats (bit pattern order == value order), optionally masked by m.
// pass `shift` in {24,16,8,0}: histogram of byte (bits >> shift) & 255 over elements whose higher bits == prefix.
__global__ __launch_bounds__(256) void radix_hist_kernel(const float* x, const uint8_t* m, long V, uint32_t prefix, int shift, unsigned int* hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t hmask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        if ((b & hmask) == (prefix & hmask)) atomicAdd(&h[(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// mask = (bits > thr) | first `need_eq` elements (index order) with bits == thr.  Single block, ordered scan.
__global__ __launch_bounds__(1024) void topk_mark_kernel(const float* x, const uint8_t* m, long V, uint32_t thr, unsigned int need_eq, uint8_t* out) {
    __shared__ unsigned int cnt[1024];
    const long per = (V + 1023) / 1024;
    const long b0 = (long)threadIdx.x * per, b1 = min(V, b0 + per);
    unsigned int c = 0;
    for (long i = b0; i < b1; ++i) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        c += (b == thr);
    }
    cnt[threadIdx.x] = c;
    __syncthreads();
    // exclusive prefix (Hillis-Steele on 1024 entries)
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned int v = threadIdx.x >= o ? cnt[threadIdx.x - o] : 0u;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int before = cnt[threadIdx.x] - c;
    for (long i = b0; i < b1; ++i) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        uint8_t o = b > thr;
        if (b == thr) { o = before < need_eq; ++before; }
        out[i] = o;
    }
}

// all elements equal to the threshold are wanted (the usual case: distinct values) -> no ordering, fully parallel
__global__ void topk_mark_all_kernel(const float* x, const uint8_t* m, long V, uint32_t thr, uint8_t* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        out[i] = b >= thr;
    }
}

// Device-resident radix select (no host round trips): state = {prefix, remaining, ties, need} in device memory.
//   per pass:  hist (as above, prefix read from the state)  ->  scan (one block: pick the digit, update the state)
//   then:      mark (parallel: bits > thr, and every tie when all ties are wanted)  +  ordered tie pass (single block; exits
//              immediately unless only some of the elements equal to the threshold are wanted)
// blockIdx.y = which of the batched selections (k values) of one call: its state / histogram sit 260 words apart
__global__ __launch_bounds__(256) void radix_hist_dev_kernel(const float* x, const uint8_t* m, long V, const unsigned int* state, int shift, unsigned int* hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    state += blockIdx.y * 260; hist += blockIdx.y * 260;
    const uint32_t prefix = state[0];
    const uint32_t hmask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long)gridDim.x * 256) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        if ((b & hmask) == (prefix & hmask)) atomicAdd(&h[(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// one wave: digits from 255 down until the running count reaches `remaining`.  Lane l owns digits 255-4l .. 252-4l; an
// inclusive wave scan of the per-lane sums locates the lane, the lane walks its four digits.  Clears the histogram.
__global__ __launch_bounds__(64) void radix_scan_kernel(unsigned int* hist, unsigned int* state, int shift) {
    const int lane = threadIdx.x;
    hist += blockIdx.x * 260; state += blockIdx.x * 260;
    unsigned int h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * lane - j]; hist[255 - 4 * lane - j] = 0; }
    const unsigned int mine = h[0] + h[1] + h[2] + h[3];
    unsigned int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const unsigned int remaining = state[1];
    const unsigned int before = incl - mine;
    if (before < remaining && incl >= remaining) {               // exactly one lane
        unsigned int acc = before;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (acc + h[j] >= remaining) {
                const unsigned int digit = 255 - 4 * lane - j;
                state[0] |= digit << shift;
                state[1] = remaining - acc;
                state[2] = h[j];                                 // after the last pass: elements equal to the threshold
                if (shift == 0) state[3] = (remaining - acc) >= h[j] ? 0xFFFFFFFFu : (remaining - acc);
                break;
            }
            acc += h[j];
        }
    }
}

// clip: the selection is AND-ed with the mask m (voxels outside m take part in the ranking with value 0, as in the dense top-k, but are
// never marked -- the reference's "no tumor_mask value outside the ball" step folded in)
__global__ void topk_mark_dev_kernel(const float* x, const uint8_t* m, long V, const unsigned int* state, uint8_t* out, int clip) {
    state += blockIdx.y * 260; out += (size_t)blockIdx.y * V;
    const uint32_t thr = state[0];
    const bool all_ties = state[3] == 0xFFFFFFFFu;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        const bool in = !m || m[i];
        const uint32_t b = in ? __float_as_uint(x[i]) : 0u;
        out[i] = (all_ties ? (b >= thr) : (b > thr)) && (in || !clip);
    }
}

// ties in index order (lower index first): only runs when some but not all elements equal to the threshold are wanted
__global__ __launch_bounds__(1024) void topk_ties_dev_kernel(const float* x, const uint8_t* m, long V, const unsigned int* state, uint8_t* out, int clip) {
    __shared__ unsigned int cnt[1024];
    state += blockIdx.x * 260; out += (size_t)blockIdx.x * V;
    const unsigned int need_eq = state[3];
    if (need_eq == 0xFFFFFFFFu) return;
    const uint32_t thr = state[0];
    const long per = (V + 1023) / 1024;
    const long b0 = (long)threadIdx.x * per, b1 = min(V, b0 + per);
    unsigned int c = 0;
    for (long i = b0; i < b1; ++i) {
        const uint32_t b = (!m || m[i]) ? __float_as_uint(x[i]) : 0u;
        c += (b == thr);
    }
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned int v = threadIdx.x >= o ? cnt[threadIdx.x - o] : 0u;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int before = cnt[threadIdx.x] - c;
    for (long i = b0; i < b1; ++i) {
        const bool in = !m || m[i];
        const uint32_t b = in ? __float_as_uint(x[i]) : 0u;
        if (b == thr) { out[i] = before < need_eq && (in || !clip); ++before; }
    }
}

struct TopkKs { unsigned int k[4]; };
__global__ void topk_state_init_kernel(unsigned int* ws, TopkKs ks) {
    ws += blockIdx.x * 260;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ws[4 + i] = 0;      // histogram
    if (threadIdx.x == 0) { ws[0] = 0; ws[1] = ks.k[blockIdx.x]; ws[2] = 0; ws[3] = 0xFFFFFFFFu; }
}

// ------------------------------------------------------------------------------------------------ GWRP rank weights
// compact the voxels of the pseudo mask, then rank each by value (desc) / index (asc) against all others.
__global__ void compact_kernel(const float* x, const uint8_t* pm, long V, float* vals, uint32_t* idx, unsigned int* n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x)
        if (pm[i]) { const unsigned int p = atomicAdd(n, 1u); vals[p] = x[i]; idx[p] = (uint32_t)i; }
}

// Order key: larger value first, ties by lower index -> "j precedes i" is ONE unsigned 64-bit compare (the float is mapped to its
// order-preserving unsigned image, the index complemented).  Each thread ranks two candidates per LDS tile element read.
__device__ __forceinline__ unsigned long long rank_key(float v, uint32_t id) {
    const uint32_t b = __float_as_uint(v);
    const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // monotone: larger float -> larger unsigned
    return ((unsigned long long)ord << 32) | (unsigned long long)(~id);
}

// Block = 64 candidates (two per thread of a 32-lane group) x 8 partitions of every 256-key tile: n / 64 blocks instead of n / 512, so a pseudo mask of a
// few thousand voxels (the usual case) spreads over the whole chip instead of a dozen CUs (8 k voxels: 183 -> ~25 us).  Keys are unique (the index is
// part of the key) and padding keys are 0 < every real key, so the count is the exact rank whatever the partitioning.
__global__ __launch_bounds__(256) void rank_weight_kernel(const float* vals, const uint32_t* idx, unsigned int n, float dlog2, float scale, float* w) {
    __shared__ unsigned long long sk[256];
    __shared__ unsigned int sr[8][64];
    const unsigned int c = threadIdx.x & 31, part = threadIdx.x >> 5;
    const unsigned int i0 = blockIdx.x * 64 + c, i1 = i0 + 32;
    const uint32_t id0 = i0 < n ? idx[i0] : 0u, id1 = i1 < n ? idx[i1] : 0u;
    const unsigned long long k0 = i0 < n ? rank_key(vals[i0], id0) : ~0ull, k1 = i1 < n ? rank_key(vals[i1], id1) : ~0ull;
    unsigned int r0 = 0, r1 = 0;
    for (unsigned int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        sk[threadIdx.x] = j0 + threadIdx.x < n ? rank_key(vals[j0 + threadIdx.x], idx[j0 + threadIdx.x]) : 0ull;
        __syncthreads();
#pragma unroll 8
        for (unsigned int j = 0; j < 32; ++j) {
            const unsigned long long kj = sk[part * 32 + j];
            r0 += kj > k0;
            r1 += kj > k1;
        }
    }
    sr[part][c] = r0;
    sr[part][c + 32] = r1;
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned int r = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) r += sr[p][threadIdx.x];
        const unsigned int i = blockIdx.x * 64 + threadIdx.x;
        if (i < n) w[idx[i]] = exp2f((float)r * dlog2) * scale;       // d^rank * N / sum_{r<N} d^r
    }
}

// The same weights from a rank ORDER (ids[r] = voxel of rank r, from a sort): identical arithmetic, O(n) -- the pairwise count above is O(n^2) and takes
// milliseconds for the pseudo masks of large tumours (33 k voxels at 40 mm).
__global__ __launch_bounds__(256) void rank_assign_kernel(const long long* ids, unsigned int n, float dlog2, float scale, float* w) {
    const unsigned int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) w[ids[r]] = exp2f((float)r * dlog2) * scale;
}

__global__ void mask_op_kernel(uint8_t* a, const uint8_t* b, long V, int op) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
        const uint8_t x = a[i], y = b[i];
        a[i] = op == 0 ? (x & y) : op == 1 ? (x | y) : (uint8_t)(x & (y ? 0 : 1));
    }
}
__global__ void zero_where_kernel(float* x, const uint8_t* m, long V) {     // x *= (1 - m)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x)
        if (m[i]) x[i] = 0.f;
}
__global__ void count_kernel(const uint8_t* m, long V, unsigned int* count) {
    unsigned int c = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) c += m[i] != 0;
    c = (unsigned int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

}  // namespace

int rs_launch_dilate_pass(const uint8_t* in, uint8_t* out, const uint8_t* flags, long nvol, int D, int H, int W, int k, hipStream_t st) {
    if (k < 1 || k > 7 || !(k & 1)) return RS_ERR_ARG;
    if ((W & 15) == 0) {
        const long items = nvol * D * H * (W >> 4);
        hipLaunchKernelGGL(dilate_pass16_kernel, dim3(rs_elem_blocks((size_t)items)), dim3(256), 0, st, in, out, flags, nvol, D, H, W, k);
    } else if ((W & 3) == 0) {
        const long words = nvol * D * H * (W >> 2);
        hipLaunchKernelGGL(dilate_pass_kernel, dim3(rs_elem_blocks((size_t)words)), dim3(256), 0, st, in, out, flags, nvol, D, H, W, k);
    } else {
        hipLaunchKernelGGL(dilate_pass_scalar_kernel, dim3(rs_elem_blocks((size_t)(nvol * D * H * W))), dim3(256), 0, st, in, out, flags, nvol, D, H, W, k);
    }
    return rs_check_launch();
}

long rs_ball_workspace_floats(int D, int H, int W, int d_odd) {
    return (long)((d_odd >> 1) + 1) * D * H * W + (long)D * ((H + 31) / 32);
}

int rs_launch_ball_conv_argmax(const float* x, int D, int H, int W, int d_odd, float std, unsigned long long* best, float* conv_out, float* ws,
                               hipStream_t st) {
    if ((d_odd >> 1) >= 64) return RS_ERR_UNSUPPORTED;
    const long V = (long)D * H * W;
    int blocks = (int)((V + 255) / 256);
    if (ws) {
        const int HW = (H + 31) / 32;
        if (HW > 64) return RS_ERR_UNSUPPORTED;
        uint32_t* bits = (uint32_t*)(ws + (size_t)((d_odd >> 1) + 1) * V);          // row-occupancy bits behind the row sums
        hipLaunchKernelGGL(ball_row_bits_kernel, dim3(D), dim3(256), 0, st, x, H, W, HW, bits);
        hipLaunchKernelGGL(ball_rowsum_kernel, dim3(blocks), dim3(256), 0, st, x, D, H, W, d_odd >> 1, 1.f / (2.f * std * std), ws);
        const int rb = (D * H + 3) / 4;
#define RS_ROWS(NX) hipLaunchKernelGGL(ball_gather_rows_kernel<NX>, dim3(rb), dim3(256), 0, st, (const float*)ws, (const uint32_t*)bits, HW, D, H, W, d_odd, \
                                       1.f / (2.f * std * std), best, conv_out)
        const bool small = (long)((d_odd >> 1) + 1) * V * 4 < (1l << 31);
        if (small && W <= 64) RS_ROWS(1);
        else if (small && W <= 128) RS_ROWS(2);
        else if (small && W <= 192) RS_ROWS(3);
        else if (small && W <= 256) RS_ROWS(4);
        else
            hipLaunchKernelGGL(ball_gather_argmax_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, (const uint32_t*)bits, HW, D, H, W, d_odd,
                               1.f / (2.f * std * std), best, conv_out);
#undef RS_ROWS
        return rs_check_launch();
    }
    hipLaunchKernelGGL(ball_conv_argmax_kernel, dim3(blocks), dim3(256), 0, st, x, D, H, W, d_odd, 1.f / (2.f * std * std), best, conv_out);
    return rs_check_launch();
}

int rs_launch_insert_ball(uint8_t* out, int D, int H, int W, int cz, int cy, int cx, int d_odd, int half, unsigned int* count, hipStream_t st) {
    hipLaunchKernelGGL(insert_ball_kernel, dim3(rs_elem_blocks((size_t)D * H * W)), dim3(256), 0, st, out, D, H, W, cz, cy, cx, d_odd, half, count);
    return rs_check_launch();
}

int rs_launch_insert_ball_at(uint8_t* out, int D, int H, int W, const unsigned long long* best, int d_odd, int half, unsigned int* count, hipStream_t st) {
    hipLaunchKernelGGL(insert_ball_at_kernel, dim3(rs_elem_blocks((size_t)D * H * W)), dim3(256), 0, st, out, D, H, W, best, d_odd, half, count);
    return rs_check_launch();
}

int rs_launch_radix_hist(const float* x, const uint8_t* m, long V, uint32_t prefix, int shift, unsigned int* hist, hipStream_t st) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(rs_elem_blocks((size_t)V) > 512 ? 512 : rs_elem_blocks((size_t)V)), dim3(256), 0, st, x, m, V, prefix, shift, hist);
    return rs_check_launch();
}

int rs_launch_topk_mark(const float* x, const uint8_t* m, long V, uint32_t thr, unsigned int need_eq, uint8_t* out, hipStream_t st) {
    if (need_eq == 0xFFFFFFFFu) hipLaunchKernelGGL(topk_mark_all_kernel, dim3(rs_elem_blocks((size_t)V)), dim3(256), 0, st, x, m, V, thr, out);
    else hipLaunchKernelGGL(topk_mark_kernel, dim3(1), dim3(1024), 0, st, x, m, V, thr, need_eq, out);
    return rs_check_launch();
}

// workspace: nk * 260 u32 on the device = {prefix, remaining, ties, need, hist[256]} per selection; out: nk volumes
int rs_launch_topk_select(const float* x, const uint8_t* m, long V, const unsigned int* k, int nk, uint8_t* out, unsigned int* ws, int clip, hipStream_t st) {
    if (nk < 1 || nk > 4) return RS_ERR_ARG;
    const int hb = rs_elem_blocks((size_t)V) > 512 ? 512 : rs_elem_blocks((size_t)V);
    TopkKs ks = {{0, 0, 0, 0}};
    for (int i = 0; i < nk; ++i) ks.k[i] = k[i];
    hipLaunchKernelGGL(topk_state_init_kernel, dim3(nk), dim3(256), 0, st, ws, ks);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(radix_hist_dev_kernel, dim3(hb, nk), dim3(256), 0, st, x, m, V, (const unsigned int*)ws, shift, ws + 4);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(nk), dim3(64), 0, st, ws + 4, ws, shift);
    }
    hipLaunchKernelGGL(topk_mark_dev_kernel, dim3(rs_elem_blocks((size_t)V), nk), dim3(256), 0, st, x, m, V, (const unsigned int*)ws, out, clip);
    hipLaunchKernelGGL(topk_ties_dev_kernel, dim3(nk), dim3(1024), 0, st, x, m, V, (const unsigned int*)ws, out, clip);
    return rs_check_launch();
}

int rs_launch_compact(const float* x, const uint8_t* pm, long V, float* vals, uint32_t* idx, unsigned int* n, hipStream_t st) {
    hipLaunchKernelGGL(compact_kernel, dim3(rs_elem_blocks((size_t)V)), dim3(256), 0, st, x, pm, V, vals, idx, n);
    return rs_check_launch();
}

int rs_launch_rank_weights(const float* vals, const uint32_t* idx, unsigned int n, float dlog2, float scale, float* w, hipStream_t st) {
    if (!n) return RS_OK;
    hipLaunchKernelGGL(rank_weight_kernel, dim3((n + 63) / 64), dim3(256), 0, st, vals, idx, n, dlog2, scale, w);
    return rs_check_launch();
}

int rs_launch_rank_assign(const long long* ids, unsigned int n, float dlog2, float scale, float* w, hipStream_t st) {
    if (!n) return RS_OK;
    hipLaunchKernelGGL(rank_assign_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, dlog2, scale, w);
    return rs_check_launch();
}

// Bit-packed label ingestion.  The reference stores label / unk / chosen-segment volumes as
// np.packbits(bool (C, D, H, W), axis=0) (dataset_abdomenatlas_UFO.py:955,970,975) and inflates them on the host with
// np.unpackbits(...)[:C] (:1031-1034) before a 16x larger H2D copy.  Here the packed bytes travel and are inflated on the
// device: out[b][c][v] = (packed[b][c >> 3][v] >> (7 - (c & 7))) & 1   (packbits is MSB-first).
// Thread = 16 voxels of one byte plane: one 16-byte load, up to eight 16-byte stores.
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, int P, int C, long V) {
    const long nvec = (V + 15) / 16;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nvec) return;
    const int pl = blockIdx.y % P, b = blockIdx.y / P;
    const uint8_t* src = packed + ((size_t)b * P + pl) * V;
    const long v0 = t * 16;
    const bool full = v0 + 16 <= V && (V & 15) == 0;
    uint8_t in[16];
    if (full) *(uint4*)in = *(const uint4*)(src + v0);
    else {
#pragma unroll
        for (int j = 0; j < 16; ++j) in[j] = v0 + j < V ? src[v0 + j] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = pl * 8 + k;
        if (c >= C) break;
        uint8_t o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = (in[j] >> (7 - k)) & 1;
        uint8_t* dst = out + ((size_t)b * C + c) * V + v0;
        if (full) *(uint4*)dst = *(const uint4*)o;
        else {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (v0 + j < V) dst[j] = o[j];
        }
    }
}

int rs_launch_unpack_bits(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, hipStream_t st) {
    const long nvec = (V + 15) / 16;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)((nvec + 255) / 256), (unsigned)(B * P)), dim3(256), 0, st, packed, out, P, C, V);
    return rs_check_launch();
}

__global__ void plane_flags_zero_kernel(uint8_t* flags, long planes);

// The same inflation for SOME class planes only: plane (b, c) is written iff flags[b * C + c] != 0 or force[c] != 0 (either table may be null; both null = every
// plane).  A byte plane none of whose eight classes is wanted is not read.  What the report losses need of a bit-packed volume: the lesion channels (force) and,
// of the unknown map, the planes that hold a voxel at all (flags from plane_any_bits_kernel) -- the other planes of `out` are never written and never read.
__global__ __launch_bounds__(256) void unpack_bits_sel_kernel(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, int P, int C, long V,
                                                              const uint8_t* __restrict__ flags, const uint8_t* __restrict__ force) {
    const long nvec = (V + 15) / 16;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int pl = blockIdx.y % P, b = blockIdx.y / P;
    unsigned want = 0;                                           // wave-uniform (depends on the block's plane only)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = pl * 8 + k;
        if (c < C && ((flags && flags[(size_t)b * C + c]) || (force && force[c]) || (!flags && !force))) want |= 1u << k;
    }
    if (!want || t >= nvec) return;
    const uint8_t* src = packed + ((size_t)b * P + pl) * V;
    const long v0 = t * 16;
    const bool full = v0 + 16 <= V && (V & 15) == 0;
    uint8_t in[16];
    if (full) *(uint4*)in = *(const uint4*)(src + v0);
    else {
#pragma unroll
        for (int j = 0; j < 16; ++j) in[j] = v0 + j < V ? src[v0 + j] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!((want >> k) & 1u)) continue;
        const int c = pl * 8 + k;
        uint8_t o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = (in[j] >> (7 - k)) & 1;
        uint8_t* dst = out + ((size_t)b * C + c) * V + v0;
        if (full) *(uint4*)dst = *(const uint4*)o;
        else {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (v0 + j < V) dst[j] = o[j];
        }
    }
}
int rs_launch_unpack_bits_sel(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, const uint8_t* flags, const uint8_t* force, hipStream_t st) {
    const long nvec = (V + 15) / 16;
    hipLaunchKernelGGL(unpack_bits_sel_kernel, dim3((unsigned)((nvec + 255) / 256), (unsigned)(B * P)), dim3(256), 0, st, packed, out, P, C, V, flags, force);
    return rs_check_launch();
}

// flags[b * C + c] = any voxel of class c of sample b set, straight from the packed bytes: the OR over byte plane (b, p) holds the eight class flags of that plane
// (bit 7 - k = class 8 p + k).  1/8 of the bytes rsuper_plane_any reads on the inflated volume.  flags must be zeroed first (plane_flags_zero_kernel).
__global__ __launch_bounds__(256) void plane_any_bits_kernel(const uint8_t* __restrict__ packed, int P, int C, long V, uint8_t* __restrict__ flags) {
    const int pl = blockIdx.y % P, b = blockIdx.y / P;
    const uint8_t* src = packed + (size_t)blockIdx.y * V;
    const long nvec = V / 16;
    unsigned int any = 0;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const uint4 a = ((const uint4*)src)[i], b4 = ((const uint4*)src)[i + stride], c = ((const uint4*)src)[i + 2 * stride], d = ((const uint4*)src)[i + 3 * stride];
        any |= (a.x | a.y | a.z | a.w) | (b4.x | b4.y | b4.z | b4.w) | (c.x | c.y | c.z | c.w) | (d.x | d.y | d.z | d.w);
    }
    for (; i < nvec; i += stride) {
        const uint4 q = ((const uint4*)src)[i];
        any |= q.x | q.y | q.z | q.w;
    }
    if (blockIdx.x == 0) for (long j = nvec * 16 + threadIdx.x; j < V; j += 256) any |= src[j];
    any |= any >> 16; any |= any >> 8; any &= 0xFFu;             // fold the four byte lanes of the word
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
    if (any && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (((any >> (7 - k)) & 1u) && pl * 8 + k < C) flags[(size_t)b * C + pl * 8 + k] = 1;
    }
}
int rs_launch_plane_any_bits(const uint8_t* packed, int B, int P, int C, long V, uint8_t* flags, hipStream_t st) {
    hipLaunchKernelGGL(plane_flags_zero_kernel, dim3((unsigned)(((long)B * C + 255) / 256)), dim3(256), 0, st, flags, (long)B * C);
    long nb = (V / 16 + 255) / 256;
    if (nb > 32) nb = 32;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(plane_any_bits_kernel, dim3((unsigned)nb, (unsigned)(B * P)), dim3(256), 0, st, packed, P, C, V, flags);
    return rs_check_launch();
}

// any(plane) for `planes` byte volumes of V voxels each: flags[p] = 1 if the plane has a non-zero byte.  Replaces the ATen
// `.flatten(1).any(1)` reductions of the loss' host control flow (150 us each on a 46 MB uint8 tensor) at HBM rate.
__global__ __launch_bounds__(256) void plane_any_kernel(const uint8_t* __restrict__ m, long V, uint8_t* __restrict__ flags) {
    const uint8_t* pl = m + (size_t)blockIdx.y * V;
    const long nvec = V / 16;
    unsigned int any = 0;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {             // four independent 16-byte loads in flight per thread (one at a time ran at 1.4 TB/s)
        const uint4 a = ((const uint4*)pl)[i], b = ((const uint4*)pl)[i + stride], c = ((const uint4*)pl)[i + 2 * stride], d = ((const uint4*)pl)[i + 3 * stride];
        any |= (a.x | a.y | a.z | a.w) | (b.x | b.y | b.z | b.w) | (c.x | c.y | c.z | c.w) | (d.x | d.y | d.z | d.w);
    }
    for (; i < nvec; i += stride) {
        const uint4 q = ((const uint4*)pl)[i];
        any |= q.x | q.y | q.z | q.w;
    }
    if (blockIdx.x == 0) for (long j = nvec * 16 + threadIdx.x; j < V; j += 256) any |= pl[j];
    if (__any(any != 0) && (threadIdx.x & 63) == 0) flags[blockIdx.y] = 1;
}

__global__ void plane_flags_zero_kernel(uint8_t* flags, long planes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < planes) flags[i] = 0;
}

// The input range guards of train_epoch (train_ddp.py:311-313: isnan / max <= hi / min >= lo, three host synchronisations in the reference) as ONE pass
// that only sets device flags: flags[0] |= any NaN, flags[1] |= any x > hi, flags[2] |= any x < lo.  The host reads the flags one step later.
__global__ __launch_bounds__(256) void guard_range_kernel(const float* __restrict__ x, size_t n, float lo, float hi, int* __restrict__ flags) {
    int bad = 0;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)x)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= (f[j] != f[j] ? 1 : 0) | (f[j] > hi ? 2 : 0) | (f[j] < lo ? 4 : 0);
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) { const float f = x[i]; bad |= (f != f ? 1 : 0) | (f > hi ? 2 : 0) | (f < lo ? 4 : 0); }
    if (__any(bad)) {                                            // wave-uniform: nothing is written on the clean path
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
        if ((threadIdx.x & 63) == 0) {
            if (bad & 1) atomicOr(flags + 0, 1);
            if (bad & 2) atomicOr(flags + 1, 1);
            if (bad & 4) atomicOr(flags + 2, 1);
        }
    }
}
// losses_foundation.py:864-869 per sample b: a chosen segment mask that is not all zero needs a non-zero unknown-voxel map and a non-zero report volume.
// m_any / u_any: one byte per sample (rsuper_plane_any), vol: [B][T] f32.  flags[0] |= mask without unknown voxels, flags[1] |= mask without a volume.
__global__ void guard_consistency_kernel(const uint8_t* m_any, const uint8_t* u_any, const float* vol, int B, int T, int* flags) {
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (!m_any[b]) continue;
        if (!u_any[b]) atomicOr(flags + 0, 1);
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += vol[(size_t)b * T + t];
        if (s == 0.f) atomicOr(flags + 1, 1);
    }
}
int rs_launch_guard_consistency(const uint8_t* m_any, const uint8_t* u_any, const float* vol, int B, int T, int* flags, hipStream_t st) {
    hipLaunchKernelGGL(guard_consistency_kernel, dim3(1), dim3(64), 0, st, m_any, u_any, vol, B, T, flags);
    return rs_check_launch();
}
int rs_launch_guard_range(const float* x, size_t n, float lo, float hi, int* flags, hipStream_t st) {
    if (n == 0) return RS_OK;
    size_t nb = (n / 4 + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(guard_range_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, lo, hi, flags);
    return rs_check_launch();
}

int rs_launch_plane_any(const uint8_t* m, long planes, long V, uint8_t* flags, hipStream_t st) {
    hipLaunchKernelGGL(plane_flags_zero_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, st, flags, planes);   // a kernel, not a memset node (optim.hip)
    long nb = (V / 16 + 255) / 256;
    if (nb > 32) nb = 32;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(plane_any_kernel, dim3((unsigned)nb, (unsigned)planes), dim3(256), 0, st, m, V, flags);
    return rs_check_launch();
}

int rs_launch_mask_op(uint8_t* a, const uint8_t* b, long V, int op, hipStream_t st) {
    hipLaunchKernelGGL(mask_op_kernel, dim3(rs_elem_blocks((size_t)V)), dim3(256), 0, st, a, b, V, op);
    return rs_check_launch();
}
int rs_launch_zero_where(float* x, const uint8_t* m, long V, hipStream_t st) {
    hipLaunchKernelGGL(zero_where_kernel, dim3(rs_elem_blocks((size_t)V)), dim3(256), 0, st, x, m, V);
    return rs_check_launch();
}
int rs_launch_count(const uint8_t* m, long V, unsigned int* count, hipStream_t st) {
    hipLaunchKernelGGL(count_kernel, dim3(rs_elem_blocks((size_t)V)), dim3(256), 0, st, m, V, count);
    return rs_check_launch();
}
