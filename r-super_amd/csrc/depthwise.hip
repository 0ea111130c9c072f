// Depthwise 3x3x3 convolution (groups = C, stride 1, zero padding 1, no bias) on channels-last fp32 activations -- the
// `depthwise` member of DepthwiseSeparableConv and MBConv in MedFormer's attention stages (SURVEY 8f-1;
// rsuper_train/model/dim3/conv_layers.py:126-157, :198-240).  HBM / L2-bound elementwise-with-halo work, no MFMA: one
// output element costs 27 multiply-adds on data it shares with its neighbours.
//
//   forward        y[v][c]  = sum_tap w[c][tap] * x[v + off(tap)][c]
//   data gradient  dx[v][c] = sum_tap w[c][26 - tap] * dy[v + off(tap)][c]          (same kernel, flipped taps)
//   weight grad.   dw[c][tap] = sum_v dy[v][c] * x[v + off(tap)][c]                 (per-block partial rows + fixed-order reduce)
//
// Thread = (voxel, 4 consecutive channels): the 16 threads of a 64-channel group read 256 contiguous bytes per neighbour.
// A block owns one 64-channel group (blockIdx.y) so its 27 x 64 weights sit in LDS transposed to [tap][channel]
// (the state_dict layout (C, 1, 3, 3, 3) has the tap innermost).  MIOpen runs these shapes on its naive reference kernels
// (2.4-2.9 ms per call; 1.2 s per training step at 96^3).
#include "common.hpp"
#include "misc.hpp"

namespace {

constexpr int DW_CG = 64;              // channels per block
constexpr int DW_VPB = 16;             // voxels per pass: 256 threads = 16 voxels x 16 channel vectors

struct DwParams {
    const float* x;        // [N][D][H][W][C] input (forward) / dy (data gradient)
    const float* w;        // (C, 1, 3, 3, 3)
    float* y;              // output, same layout
    int N, D, H, W, C;
    int flip;              // 1: use w[c][26 - tap] (data gradient)
};

// Each thread produces DW_R consecutive outputs along W for its 4 channels: a (kd, kh) row of the halo is DW_R + 2 vectors that feed
// 3 * DW_R multiply-adds each, i.e. 9 * (DW_R + 2) loads per DW_R outputs instead of 27 per output (2x fewer at DW_R = 4) -- the kernel
// is bound by L1 / L2 transactions, not by HBM.
constexpr int DW_R = 4;

// XCD-aware work order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2): with a plain grid-stride loop the
// blocks of one XCD touch every 8th group of runs, so each L2 holds its own copy of every halo plane (measured 6x the algorithmic
// bytes on the L2 -> fabric side, rocprofv3 FETCH_SIZE).  Here XCD x owns a contiguous slab of run groups and its blocks sweep that slab
// together, so the d-1 / d+1 planes a block needs were just fetched by its neighbours on the same XCD.
struct DwSweep { long first, step, end; };
__device__ __forceinline__ DwSweep dw_sweep(long runs) {
    const long G = (runs + DW_VPB - 1) / DW_VPB;                  // groups of 16 runs (one block pass)
    const long Gx = (G + 7) / 8;                                  // groups per XCD
    const long x = blockIdx.x & 7, idx = blockIdx.x >> 3, Bx = (gridDim.x + 7 - x) / 8;
    DwSweep s;
    if (gridDim.x < 8) { s.first = blockIdx.x; s.step = gridDim.x; s.end = G; return s; }   // small launches: plain grid stride
    s.first = x * Gx + idx;
    s.step = Bx > 0 ? Bx : 1;
    s.end = min((x + 1) * Gx, G);
    return s;
}


__global__ __launch_bounds__(256, 2) void depthwise_fwd_kernel(DwParams p) {
    __shared__ float wl[27][DW_CG];
    const int c0 = blockIdx.y * DW_CG;
    const int ncl = min(DW_CG, p.C - c0);
    for (int i = threadIdx.x; i < 27 * DW_CG; i += 256) {
        const int c = i / 27, tap = i - c * 27;                       // coalesced over the (c, tap) source order
        wl[p.flip ? 26 - tap : tap][c] = c < ncl ? p.w[(size_t)(c0 + c) * 27 + tap] : 0.f;
    }
    __syncthreads();
    const int cv = threadIdx.x & 15, vl = threadIdx.x >> 4;
    const int c = c0 + cv * 4;
    if (c >= p.C) return;
    const int WR = (p.W + DW_R - 1) / DW_R;                           // runs per row
    const long runs = (long)p.N * p.D * p.H * WR;
    const DwSweep sw = dw_sweep(runs);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)((size_t)p.N * p.D * p.H * p.W * p.C * 4), 0x00020000);
    for (long gidx = sw.first; gidx < sw.end; gidx += sw.step) {
        const long r = gidx * DW_VPB + vl;
        if (r >= runs) break;
        const int xr = (int)(r % WR);
        long t2 = r / WR;
        const int yh = (int)(t2 % p.H); t2 /= p.H;
        const int zd = (int)(t2 % p.D);
        const int x0 = xr * DW_R;
        const long v0 = (t2 * p.H + yh) * p.W + x0;                   // t2 = n * D + zd
        const uint32_t boff = (uint32_t)(((size_t)v0 * p.C + c) * 4);
        float4 acc[DW_R];
#pragma unroll
        for (int j = 0; j < DW_R; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const bool okd = (unsigned)(zd + kd - 1) < (unsigned)p.D;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const bool okr = okd && (unsigned)(yh + kh - 1) < (unsigned)p.H;
                const uint32_t roff = boff + (uint32_t)(((kd - 1) * p.H + (kh - 1)) * p.W * p.C * 4);
                float4 q[DW_R + 2];
#pragma unroll
                for (int j = 0; j < DW_R + 2; ++j) {
                    // buffer load with a 32-bit byte offset: out-of-volume taps get an out-of-range offset and read as zeros -- no
                    // select, no branch, and no 64-bit address per load (54 of those were 108 VGPRs)
                    const bool ok = okr && (unsigned)(x0 + j - 1) < (unsigned)p.W;
                    const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? roff + (uint32_t)((j - 1) * p.C * 4) : 0xFFFFFFFFu, 0, 0);
                    q[j] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 wv = *(const float4*)&wl[(kd * 3 + kh) * 3 + kw][cv * 4];
#pragma unroll
                    for (int j = 0; j < DW_R; ++j) {
                        acc[j].x = fmaf(q[j + kw].x, wv.x, acc[j].x);
                        acc[j].y = fmaf(q[j + kw].y, wv.y, acc[j].y);
                        acc[j].z = fmaf(q[j + kw].z, wv.z, acc[j].z);
                        acc[j].w = fmaf(q[j + kw].w, wv.w, acc[j].w);
                    }
                }
                // Fence per halo row, with the accumulators as in/out operands so that the multiply-adds of this row cannot sink below it: otherwise
                // the scheduler issues all 54 loads of the run first and keeps them live (350 VGPRs, one wave per SIMD; a register cap alone
                // only turns that into scratch spills).
#pragma unroll
                for (int j = 0; j < DW_R; ++j)
                    asm volatile("" : "+v"(acc[j].x), "+v"(acc[j].y), "+v"(acc[j].z), "+v"(acc[j].w) :: "memory");
            }
        }
#pragma unroll
        for (int j = 0; j < DW_R; ++j)
            if (x0 + j < p.W) *(float4*)(p.y + (size_t)(v0 + j) * p.C + c) = acc[j];
    }
}

// LDS-tiled forward / data gradient for the large volumes.  The register-run kernel above re-reads every input vector 13.5 times
// through L1 / L2 and is bound by that traffic (2.2 TB/s effective on 48^3 x 256 channels; fewer loads per output ran faster in
// proportion).  Here a block owns an 8 x 8 (H x W) column of voxels x 32 channels and marches along D with a ring of three halo planes
// (10 x 10 voxels) in LDS: every input vector is fetched from L2 once per block (1.56x the unique bytes with the halo), the 27-tap
// stencil reads LDS.  Thread = (4 outputs along W, 4 channels), 128 threads per block; global loads run two planes ahead of the
// stencil in registers and land in LDS when their slot is free (two barriers per plane).  Row layout in LDS: one pad voxel after every
// four, so the two 4-voxel runs of a row that a 16-lane group reads together start 640 bytes apart and use disjoint banks.
constexpr int DL_T = 8;                                  // tile edge in H and W
constexpr int DL_C = 32;                                 // channels per block
constexpr int DL_ROW = 12;                               // LDS slots per halo row: 10 voxels + a pad slot after every 4
constexpr int DL_PLANE = (DL_T + 2) * DL_ROW;            // slots per halo plane

__global__ __launch_bounds__(128) void depthwise_lds_kernel(DwParams p, int dsegs, int tiles_w, int tiles_h) {
    __shared__ float4 ring[3][DL_PLANE][DL_C / 4];       // 46 KB
    __shared__ float wl[27][DL_C];
    const int c0 = blockIdx.y * DL_C;
    const int n = blockIdx.z;
    int bx = blockIdx.x;
    const int tw = bx % tiles_w; bx /= tiles_w;
    const int th = bx % tiles_h; bx /= tiles_h;
    const int seg = bx;
    const int dlen = (p.D + dsegs - 1) / dsegs;
    const int d0 = seg * dlen, d1 = min(p.D, d0 + dlen);
    const int h0 = th * DL_T, w0 = tw * DL_T;
    for (int i = threadIdx.x; i < 27 * DL_C; i += 128) {
        const int c = i / 27, tap = i - c * 27;
        wl[p.flip ? 26 - tap : tap][c] = c0 + c < p.C ? p.w[(size_t)(c0 + c) * 27 + tap] : 0.f;
    }
    if (d0 >= d1) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)((size_t)p.N * p.D * p.H * p.W * p.C * 4), 0x00020000);
    // staging map: element i of a halo plane = (voxel v = i / 8 in 10 x 10, channel vector i % 8); at most 7 per thread
    constexpr int NST = ((DL_T + 2) * (DL_T + 2) * (DL_C / 4) + 127) / 128;
    uint32_t goff[NST]; int lslot[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = threadIdx.x + k * 128;
        const int v = i >> 3, c4 = i & 7;
        const int hy = v / (DL_T + 2), wx = v - hy * (DL_T + 2);
        const int hh = h0 - 1 + hy, ww = w0 - 1 + wx;
        const bool ok = v < (DL_T + 2) * (DL_T + 2) && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W && c0 + c4 * 4 < p.C;
        goff[k] = ok ? (uint32_t)((((size_t)hh * p.W + ww) * p.C + c0 + c4 * 4) * 4) : 0xFFFFFFFFu;      // offset inside one depth plane
        lslot[k] = v < (DL_T + 2) * (DL_T + 2) ? (hy * DL_ROW + wx + (wx >> 2)) * (DL_C / 4) + c4 : -1;
    }
    const uint32_t plane_bytes = (uint32_t)((size_t)p.H * p.W * p.C * 4);
    auto load_plane = [&](int d, float4 (&q)[NST]) {
        const bool okd = (unsigned)d < (unsigned)p.D;
        const uint32_t base = (uint32_t)(n * p.D + (okd ? d : 0)) * plane_bytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, (okd && goff[k] != 0xFFFFFFFFu) ? base + goff[k] : 0xFFFFFFFFu, 0, 0);
            q[k] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
        }
    };
    auto store_plane = [&](int slot, const float4 (&q)[NST]) {
        float4* dst = &ring[slot][0][0];
#pragma unroll
        for (int k = 0; k < NST; ++k)
            if (lslot[k] >= 0) dst[lslot[k]] = q[k];
    };
    float4 q[NST], qn[NST];                                          // plane d + 2 (arrived, waiting for its slot) and plane d + 3 (in flight)
    // ring slot of plane d: (d - (d0 - 1)) % 3
    load_plane(d0 - 1, q); store_plane(0, q);
    load_plane(d0, q); store_plane(1, q);
    load_plane(d0 + 1, q); store_plane(2, q);
    if (d0 + 2 <= d1) load_plane(d0 + 2, q);
    __syncthreads();
    const int cv = threadIdx.x & 7, pos = threadIdx.x >> 3;          // pos 0..15: row = pos / 2, run = pos % 2
    const int row = pos >> 1, x0 = (pos & 1) * 4;
    const int c = c0 + cv * 4;
    const int xs = x0 + (x0 >> 2);                                   // first LDS slot of the run's halo (voxel x0 of the 10-wide row)
    for (int d = d0; d < d1; ++d) {
        const int rel = d - d0;                                      // planes d-1, d, d+1 sit in slots rel % 3, (rel + 1) % 3, (rel + 2) % 3
        if (d + 3 <= d1) load_plane(d + 3, qn);                      // two planes ahead: in flight during two stencils (plane d1 + 1 is never needed)
        float4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float4* pl = &ring[(rel + kd) % 3][0][0];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                float4 v[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int xv = x0 + j;                           // voxel index in the 10-wide halo row
                    v[j] = pl[((row + kh) * DL_ROW + xv + (xv >> 2)) * (DL_C / 4) + cv];
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 wv = *(const float4*)&wl[(kd * 3 + kh) * 3 + kw][cv * 4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j].x = fmaf(v[j + kw].x, wv.x, acc[j].x);
                        acc[j].y = fmaf(v[j + kw].y, wv.y, acc[j].y);
                        acc[j].z = fmaf(v[j + kw].z, wv.z, acc[j].z);
                        acc[j].w = fmaf(v[j + kw].w, wv.w, acc[j].w);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("" : "+v"(acc[j].x), "+v"(acc[j].y), "+v"(acc[j].z), "+v"(acc[j].w));
            }
        }
        const int hh = h0 + row;
        if (hh < p.H && c < p.C) {
            float* yo = p.y + ((((size_t)n * p.D + d) * p.H + hh) * p.W + w0 + x0) * p.C + c;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w0 + x0 + j < p.W) *(float4*)(yo + (size_t)j * p.C) = acc[j];
        }
        __syncthreads();                                             // every read of plane d - 1's slot is done
        if (d + 2 <= d1) store_plane(rel % 3, q);                    // plane d + 2 takes the slot of plane d - 1
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NST; ++k) q[k] = qn[k];
    }
    (void)xs;
}

// Weight gradient, stage 1: block (bx, channel group) accumulates dw over its voxels in registers (27 taps x 4 channels per
// thread), then the 16 voxel lanes of each channel vector are summed through LDS in a fixed order and the block writes one
// row part[bx][tap][c].  Stage 2 sums the rows (fixed order: deterministic) into (C, 1, 3, 3, 3).
struct DwWgParams {
    const float* x; const float* dy;
    float* part;           // [rows][27][C]
    int N, D, H, W, C;
};

__global__ __launch_bounds__(256, 2) void depthwise_wgrad_kernel(DwWgParams p) {
    __shared__ float red[4][27][DW_CG + 4];                        // one partial per wave (29 KB)
    const int c0 = blockIdx.y * DW_CG;
    const int cv = threadIdx.x & 15, vl = threadIdx.x >> 4;
    const int c = c0 + cv * 4;
    const bool cok = c < p.C;
    float4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int WR = (p.W + DW_R - 1) / DW_R;
    const long runs = (long)p.N * p.D * p.H * WR;
    const DwSweep sw = dw_sweep(runs);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)((size_t)p.N * p.D * p.H * p.W * p.C * 4), 0x00020000);
    if (cok)
        for (long gidx = sw.first; gidx < sw.end; gidx += sw.step) {
            const long r = gidx * DW_VPB + vl;
            if (r >= runs) break;
            const int xr = (int)(r % WR);
            long t2 = r / WR;
            const int yh = (int)(t2 % p.H); t2 /= p.H;
            const int zd = (int)(t2 % p.D);
            const int x0 = xr * DW_R;
            const long v0 = (t2 * p.H + yh) * p.W + x0;
            const uint32_t boff = (uint32_t)(((size_t)v0 * p.C + c) * 4);
            float4 g[DW_R];
#pragma unroll
            for (int j = 0; j < DW_R; ++j)
                g[j] = x0 + j < p.W ? *(const float4*)(p.dy + (size_t)(v0 + j) * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const bool okd = (unsigned)(zd + kd - 1) < (unsigned)p.D;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const bool okr = okd && (unsigned)(yh + kh - 1) < (unsigned)p.H;
                    const uint32_t roff = boff + (uint32_t)(((kd - 1) * p.H + (kh - 1)) * p.W * p.C * 4);
                    float4 q[DW_R + 2];
#pragma unroll
                    for (int j = 0; j < DW_R + 2; ++j) {
                        const bool ok = okr && (unsigned)(x0 + j - 1) < (unsigned)p.W;
                        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? roff + (uint32_t)((j - 1) * p.C * 4) : 0xFFFFFFFFu, 0, 0);
                        q[j] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float4& a = acc[(kd * 3 + kh) * 3 + kw];
#pragma unroll
                        for (int j = 0; j < DW_R; ++j) {
                            a.x = fmaf(q[j + kw].x, g[j].x, a.x);
                            a.y = fmaf(q[j + kw].y, g[j].y, a.y);
                            a.z = fmaf(q[j + kw].z, g[j].z, a.z);
                            a.w = fmaf(q[j + kw].w, g[j].w, a.w);
                        }
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {                 // fence: one row of loads live at a time (see the forward kernel)
                        float4& a = acc[(kd * 3 + kh) * 3 + kw];
                        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");
                    }
                }
            }
        }
    // the four voxel lanes of a wave that share a channel vector sit 16 and 32 lanes apart: two butterfly steps, fixed order
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        float4 a = acc[t];
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        }
        if ((threadIdx.x & 63) < 16) *(float4*)&red[threadIdx.x >> 6][t][cv * 4] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * DW_CG; i += 256) {
        const int t = i / DW_CG, cc = i - t * DW_CG;
        if (c0 + cc >= p.C) continue;
        p.part[((size_t)blockIdx.x * 27 + t) * p.C + c0 + cc] = (red[0][t][cc] + red[1][t][cc]) + (red[2][t][cc] + red[3][t][cc]);
    }
}

// dw[c][tap] = sum over the per-block partial rows.  Block = 32 consecutive (tap, c) elements x 8 row groups (round 3: one thread per element walked
// up to 1024 rows alone, four loads in flight -- 10-30 us of pure latency per launch, 58 launches per MedFormer step); a row of 32 threads reads 128
// contiguous bytes per partial row, every thread has four loads in flight, the 8 group sums meet in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void depthwise_wgrad_reduce_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + el;                               // i = tap * C + c
    const bool ok = i < 27 * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        int r = g;
        for (; r + 24 < rows; r += 32) {
            s0 += part[(size_t)r * 27 * C + i]; s1 += part[(size_t)(r + 8) * 27 * C + i];
            s2 += part[(size_t)(r + 16) * 27 * C + i]; s3 += part[(size_t)(r + 24) * 27 * C + i];
        }
        for (; r < rows; r += 8) s0 += part[(size_t)r * 27 * C + i];
    }
    red[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && ok) {
        float t = red[0][el];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][el];
        const int tap = i / C, c = i - tap * C;
        dw[(size_t)c * 27 + tap] = t;
    }
}

}  // namespace

int rs_depthwise_rows(long vox) {
    long b = (vox + DW_VPB * 8 - 1) / (DW_VPB * 8);                   // >= 8 passes per block
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int rs_launch_depthwise(const float* x, const float* w, float* y, int N, int D, int H, int W, int C, int flip, hipStream_t st) {
    if ((long)N * D * H * W * C >= (1L << 30)) return RS_ERR_UNSUPPORTED;       // 32-bit byte offsets of the buffer loads (< 4 GiB)
    DwParams p = {x, w, y, N, D, H, W, C, flip};
    const long vox = (long)N * D * H * W;
    static const bool no_lds = getenv("RSUPER_DW_NO_LDS") != nullptr;          // A/B switch: the register-run kernel for every shape
    // measured (B = 2): 48^3 x 256 ch 209 -> 149 us; 24^3 x 512 41 -> 40, 24^3 x 384 33 -> 37, 24^3 x 128 14 -> 20 us (the halo planes of the
    // short D segments and 6 waves per CU cost more than the L2 re-reads there): the LDS kernel takes the 32^2-and-larger planes only
    if (!no_lds && H >= 32 && W >= 32 && D >= 6 && (C % 4) == 0 && (size_t)H * W * C * 4 < 0x40000000ull) {
        const int th = (H + DL_T - 1) / DL_T, tw = (W + DL_T - 1) / DL_T, groups = (C + DL_C - 1) / DL_C;
        // split D so that >= ~1500 blocks are in flight (3 per CU), but keep >= 6 planes per segment (2 halo planes are re-read per segment)
        int dsegs = 1;
        while ((long)th * tw * groups * N * dsegs < 1536 && D / (dsegs + 1) >= 6) ++dsegs;
        hipLaunchKernelGGL(depthwise_lds_kernel, dim3((unsigned)(th * tw * dsegs), groups, N), dim3(128), 0, st, p, dsegs, tw, th);
        return rs_check_launch();
    }
    const long runs = (long)N * D * H * ((W + DW_R - 1) / DW_R);
    // the LDS weight staging is per block: aim at >= 8 passes per block, but never below ~1024 blocks in flight (small volumes)
    const long groups = (C + DW_CG - 1) / DW_CG, all = (runs + DW_VPB - 1) / DW_VPB;
    long bx = (runs + DW_VPB * 8 - 1) / (DW_VPB * 8);
    if (bx * groups < 1024) bx = (1024 + groups - 1) / groups;
    if (bx > all) bx = all;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(depthwise_fwd_kernel, dim3((unsigned)bx, (C + DW_CG - 1) / DW_CG), dim3(256), 0, st, p);
    return rs_check_launch();
}

int rs_launch_depthwise_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int D, int H, int W, int C, hipStream_t st) {
    if ((long)N * D * H * W * C >= (1L << 30)) return RS_ERR_UNSUPPORTED;
    const int rows = rs_depthwise_rows((long)N * D * H * W);
    DwWgParams p = {x, dy, part, N, D, H, W, C};
    hipLaunchKernelGGL(depthwise_wgrad_kernel, dim3(rows, (C + DW_CG - 1) / DW_CG), dim3(256), 0, st, p);
    hipLaunchKernelGGL(depthwise_wgrad_reduce_kernel, dim3((27 * C + 31) / 32), dim3(256), 0, st, (const float*)part, rows, C, dw);
    return rs_check_launch();
}
