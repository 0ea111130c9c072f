// HBM-bound kernels of the UNet path (gfx950): everything around the MFMA convolutions.
// Activations are channels-last [N][D][H][W][C] (views with channel stride ld), 16-byte vector accesses.
//
//   stats_finalize        per-block partial sums -> per-(n,c) (mean, rstd) / backward means
//                         (nn.InstanceNorm3d(eps=1e-4, affine=False), model/dim3/conv_layers.py:40-42)
//   in_bwd_finalize       InstanceNorm backward tail: dx = rstd*(g - mean(g) - x_n*mean(g*x_n)) [+ adds]
//   maxpool2 fwd/bwd      nn.MaxPool3d(2)                       (model/dim3/unet_utils.py:35-37)
//   upsample fwd/bwd      F.interpolate(trilinear, align_corners=True)   (unet_utils.py:69)
//   stem fwd/wgrad        inconv.conv1: Conv3d(1 -> C, k=3, bias=False)  (unet_utils.py:14)
//   head fwd/bwd          outc: Conv3d(C -> K, k=1) + bias               (unet.py:47)
#include <string.h>
#include "common.hpp"
#include "kernels.hpp"
#include "misc.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ stats
// part: [N][nblk][C][2] floats.  mode 0: out = (mean, rstd);  mode 1: out = (sum0/cnt, sum1/cnt).
// 1024 threads = 32 channels x 32 row groups: every row read is 32 channels x 8 B contiguous, four independent loads
// in flight per thread; f64 accumulation.
__global__ __launch_bounds__(1024) void stats_finalize_kernel(const float* part, int nblk, int C, double cnt, float eps, int mode, int split, float* out) {
    __shared__ double red[32][32][2];
    const int n = blockIdx.y;
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        const float* base = part + ((size_t)n * nblk * C + c) * 2;
        int b = g;
        for (; b + 96 < nblk; b += 128) {
            const float2 v0 = *(const float2*)(base + (size_t)b * C * 2), v1 = *(const float2*)(base + (size_t)(b + 32) * C * 2);
            const float2 v2 = *(const float2*)(base + (size_t)(b + 64) * C * 2), v3 = *(const float2*)(base + (size_t)(b + 96) * C * 2);
            s0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            s1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        }
        for (; b < nblk; b += 32) {
            const float2 v = *(const float2*)(base + (size_t)b * C * 2);
            s0 += (double)v.x; s1 += (double)v.y;
        }
    }
    red[g][cl][0] = s0; red[g][cl][1] = s1;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int k = 1; k < 32; ++k) { s0 += red[k][cl][0]; s1 += red[k][cl][1]; }
        // split > 0: two contiguous tables [N][split][2] | [N][C-split][2] (fused conv1+shortcut columns, concatenated sources)
        float* o = split <= 0 ? out + ((size_t)n * C + c) * 2
                   : c < split ? out + ((size_t)n * split + c) * 2
                               : out + (size_t)gridDim.y * split * 2 + ((size_t)n * (C - split) + (c - split)) * 2;
        if (mode == 0) {
            const double mean = s0 / cnt;
            double var = s1 / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            o[0] = (float)mean;
            o[1] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            o[0] = (float)(s0 / cnt); o[1] = (float)(s1 / cnt);
        }
    }
}

// Block-level per-channel partial sums for the elementwise kernels.
// Thread layout: CV = C/KP channel vectors, VL = 256/CV voxel lanes; thread (vl, s) owns KP channels.
template <int KP>
__device__ __forceinline__ void block_channel_sums(const float* s1, const float* s2, int C, int CV, int VL, int vl, int s,
                                                   bool active, float* part_blk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;                                  // [VL][C][2]
    if (active) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            red[((vl * C) + s * KP + j) * 2] = s1[j];
            red[((vl * C) + s * KP + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int v = 0; v < VL; ++v) { a += red[(v * C + c) * 2]; b += red[(v * C + c) * 2 + 1]; }
        part_blk[c * 2] = a; part_blk[c * 2 + 1] = b;
    }
}

// ------------------------------------------------------------------------------------------------ IN backward tail
// dx = rstd * (g - gm0 - x_n * gm1) [+ add1] [+ add2],  x_n = (x - mean) * rstd.
// The launch makes the grid stride a multiple of the channel-vector count whenever 256 % (C / KP) == 0, so a thread keeps its
// channel slot for the whole loop: the 4 x KP per-channel constants are loaded once (they were 32 loads per 16-byte vector)
// and the index math is 32-bit (the old size_t div / mod per vector cost more issue slots than the arithmetic).
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(InBwdParams p) {
    constexpr int KP = Elem<T>::KP;
    const uint32_t CV = (uint32_t)p.C / KP;
    const uint32_t total = (uint32_t)p.vox * CV;                 // per sample (< 2^32: checked by the launcher)
    const uint32_t n = blockIdx.y;
    const uint32_t stride = gridDim.x * 256u;
    const bool fixed_slot = (256u % CV) == 0;
    const T* gp = (const T*)p.g + (size_t)n * p.vox * p.ldg;
    const T* xp = (const T*)p.x + (size_t)n * p.vox * p.ldx;
    const T* a1p = p.add1 ? (const T*)p.add1 + (size_t)n * p.vox * p.lda1 : nullptr;
    const T* a2p = p.add2 ? (const T*)p.add2 + (size_t)n * p.vox * p.lda2 : nullptr;
    T* op = (T*)p.out + (size_t)n * p.vox * p.ldo;
    float mu[KP], rs[KP], m1[KP], m2[KP];
    auto constants = [&](uint32_t c) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const size_t o = ((size_t)n * p.C + c + j) * 2;
            mu[j] = p.mr[o]; rs[j] = p.mr[o + 1]; m1[j] = p.gm[o]; m2[j] = p.gm[o + 1];
        }
    };
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t v = i / CV, c = (i - v * CV) * KP;
    const uint32_t vstep = stride / CV;                          // exact when the slot is fixed
    if (fixed_slot && i < total) constants(c);
    for (; i < total; i += stride, v += vstep) {
        if (!fixed_slot) { v = i / CV; c = (i - v * CV) * KP; constants(c); }
        float g[KP], x[KP], o[KP];
        const uint4 gq = *(const uint4*)(gp + (size_t)v * p.ldg + c);
        const uint4 xq = *(const uint4*)(xp + (size_t)v * p.ldx + c);
        uint4 a1q = make_uint4(0, 0, 0, 0), a2q = make_uint4(0, 0, 0, 0);
        if (a1p) a1q = *(const uint4*)(a1p + (size_t)v * p.lda1 + c);
        if (a2p) a2q = *(const uint4*)(a2p + (size_t)v * p.lda2 + c);
        unpack16<T>(gq, g);
        unpack16<T>(xq, x);
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const float xn = (x[j] - mu[j]) * rs[j];
            o[j] = rs[j] * (g[j] - m1[j] - xn * m2[j]);
        }
        if (a1p) {
            float a[KP];
            unpack16<T>(a1q, a);
#pragma unroll
            for (int j = 0; j < KP; ++j) o[j] += a[j];
        }
        if (a2p) {
            float a[KP];
            unpack16<T>(a2q, a);
#pragma unroll
            for (int j = 0; j < KP; ++j) o[j] += a[j];
        }
        *(uint4*)(op + (size_t)v * p.ldo + c) = pack16<T>(o);
    }
}

// ------------------------------------------------------------------------------------------------ max pool 2x2x2
// grid: (blocks over output voxels of one sample, N).  Output partial stats for the next InstanceNorm.
template <typename T>
__global__ void maxpool_fwd_kernel(PoolParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP, VL = 256 / CV;
    const int vl = threadIdx.x / CV, s = threadIdx.x % CV;
    const bool active = vl < VL;
    const int n = blockIdx.y;
    const int OD = p.D / 2, OH = p.H / 2, OW = p.W / 2;
    const int ovox = OD * OH * OW;
    const int per_blk = (ovox + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * per_blk, v1 = min(ovox, v0 + per_blk);
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (active)
        for (int v = v0 + vl; v < v1; v += VL) {
            const int ow = v % OW, oh = (v / OW) % OH, od = v / (OW * OH);
            float m[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) m[j] = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int d = od * 2 + (k >> 2), h = oh * 2 + ((k >> 1) & 1), w = ow * 2 + (k & 1);
                float x[KP];
                unpack16<T>(*(const uint4*)((const T*)p.x + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * (size_t)p.ldx + s * KP), x);
#pragma unroll
                for (int j = 0; j < KP; ++j) m[j] = fmaxf(m[j], x[j]);
            }
            *(uint4*)((T*)p.y + ((size_t)n * ovox + v) * p.ldy + s * KP) = pack16<T>(m);
#pragma unroll
            for (int j = 0; j < KP; ++j) { s1[j] += m[j]; s2[j] += m[j] * m[j]; }
        }
    if (p.part) block_channel_sums<KP>(s1, s2, p.C, CV, VL, vl, s, active, p.part + ((size_t)n * gridDim.x + blockIdx.x) * p.C * 2);
}

// dx = dy routed to the FIRST maximum in (d,h,w) scan order (ATen max_pool3d keeps the first `>`), else 0.
template <typename T>
__global__ void maxpool_bwd_kernel(PoolParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP;
    const int n = blockIdx.y;
    const int OD = p.D / 2, OH = p.H / 2, OW = p.W / 2;
    const size_t total = (size_t)OD * OH * OW * CV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % CV); const int v = (int)(i / CV);
        const int ow = v % OW, oh = (v / OW) % OH, od = v / (OW * OH);
        float x[8][KP], m[KP], dy[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = od * 2 + (k >> 2), h = oh * 2 + ((k >> 1) & 1), w = ow * 2 + (k & 1);
            unpack16<T>(*(const uint4*)((const T*)p.x + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * (size_t)p.ldx + s * KP), x[k]);
#pragma unroll
            for (int j = 0; j < KP; ++j) m[j] = fmaxf(m[j], x[k][j]);
        }
        unpack16<T>(*(const uint4*)((const T*)p.y + ((size_t)n * OD * OH * OW + v) * p.ldy + s * KP), dy);
        bool done[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) done[j] = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = od * 2 + (k >> 2), h = oh * 2 + ((k >> 1) & 1), w = ow * 2 + (k & 1);
            float o[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const bool hit = !done[j] && x[k][j] == m[j];
                o[j] = hit ? dy[j] : 0.f;
                done[j] = done[j] || hit;
            }
            if (p.add) {                                         // dx = scatter(dy) + add, rounded once: what autograd's accumulation of the two gradients stores
                float a[KP];
                unpack16<T>(*(const uint4*)((const T*)p.add + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * (size_t)p.lda + s * KP), a);
#pragma unroll
                for (int j = 0; j < KP; ++j) o[j] += a[j];
            }
            *(uint4*)((T*)p.dx + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * (size_t)p.lddx + s * KP) = pack16<T>(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ stride-2 subsample
// y[od, oh, ow] = x[2 od, 2 oh, 2 ow] (output size ceil(D / 2)): the stride-(2,2,2) / pad-1 convolution of down_block(pool=False)
// (model/dim3/unet_utils.py:38-39) is the stride-1 convolution evaluated at the even voxels.  Output partial stats for the next
// InstanceNorm, same grid / partial layout as the max pool.
template <typename T>
__global__ void subsample_fwd_kernel(PoolParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP, VL = 256 / CV;
    const int vl = threadIdx.x / CV, s = threadIdx.x % CV;
    const bool active = vl < VL;
    const int n = blockIdx.y;
    const int OD = (p.D + 1) / 2, OH = (p.H + 1) / 2, OW = (p.W + 1) / 2;
    const int ovox = OD * OH * OW;
    const int per_blk = (ovox + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * per_blk, v1 = min(ovox, v0 + per_blk);
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (active)
        for (int v = v0 + vl; v < v1; v += VL) {
            const int ow = v % OW, oh = (v / OW) % OH, od = v / (OW * OH);
            const uint4 q = *(const uint4*)((const T*)p.x + ((((size_t)n * p.D + 2 * od) * p.H + 2 * oh) * p.W + 2 * ow) * (size_t)p.ldx + s * KP);
            *(uint4*)((T*)p.y + ((size_t)n * ovox + v) * p.ldy + s * KP) = q;
            float m[KP];
            unpack16<T>(q, m);
#pragma unroll
            for (int j = 0; j < KP; ++j) { s1[j] += m[j]; s2[j] += m[j] * m[j]; }
        }
    if (p.part) block_channel_sums<KP>(s1, s2, p.C, CV, VL, vl, s, active, p.part + ((size_t)n * gridDim.x + blockIdx.x) * p.C * 2);
}

// dx[d, h, w] = dy[d / 2, h / 2, w / 2] at the even voxels, 0 elsewhere (every input voxel is written)
template <typename T>
__global__ void subsample_bwd_kernel(PoolParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP;
    const int n = blockIdx.y;
    const int OH = (p.H + 1) / 2, OW = (p.W + 1) / 2, OD = (p.D + 1) / 2;
    const size_t total = (size_t)p.D * p.H * p.W * CV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % CV); const size_t v = i / CV;
        const int w = (int)(v % p.W), h = (int)((v / p.W) % p.H), d = (int)(v / ((size_t)p.W * p.H));
        uint4 q = make_uint4(0, 0, 0, 0);
        if (!((d | h | w) & 1))
            q = *(const uint4*)((const T*)p.y + (((size_t)n * OD + d / 2) * OH + h / 2) * OW * (size_t)p.ldy + (size_t)(w / 2) * p.ldy + s * KP);
        *(uint4*)((T*)p.dx + ((size_t)n * p.D * p.H * p.W + v) * (size_t)p.lddx + s * KP) = q;
    }
}

// ------------------------------------------------------------------------------------------------ trilinear, align_corners
__host__ __device__ __forceinline__ void lin_coord(int o, float scale, int I, int& i0, int& i1, float& l1) {
    // ATen area_pixel_compute_source_index(align_corners=True): src = scale * dst (float32).  No contraction of `scale * o - i0` into one fma: the
    // host sizes the backward kernel's tap tables with this same function and must see the same zero / non-zero weights as the device
#pragma clang fp contract(off)
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > I - 1) i0 = I - 1;
    i1 = i0 + (i0 < I - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <typename T>
__global__ void upsample_fwd_kernel(UpParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP, VL = 256 / CV;
    const int vl = threadIdx.x / CV, s = threadIdx.x % CV;
    const bool active = vl < VL;
    const int n = blockIdx.y;
    const int ovox = p.OD * p.OH * p.OW;
    const int per_blk = (ovox + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * per_blk, v1 = min(ovox, v0 + per_blk);
    const float sd = p.OD > 1 ? (float)(p.ID - 1) / (float)(p.OD - 1) : 0.f;
    const float sh = p.OH > 1 ? (float)(p.IH - 1) / (float)(p.OH - 1) : 0.f;
    const float sw = p.OW > 1 ? (float)(p.IW - 1) / (float)(p.OW - 1) : 0.f;
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // two output voxels per iteration: all 16 input vectors are requested before the first is used (the loop is latency-bound:
    // 8 dependent L2 round trips per voxel at 4 waves per SIMD reached only 1.5 TB/s)
    if (active)
        for (int v = v0 + vl; v < v1; v += 2 * VL) {
            uint4 q[2][8];
            float wt[2][8];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int vv = v + u * VL;
                ok[u] = vv < v1;
                const int vc = ok[u] ? vv : v;
                const int ow = vc % p.OW, oh = (vc / p.OW) % p.OH, od = vc / (p.OW * p.OH);
                int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
                lin_coord(od, sd, p.ID, d0, d1, ld);
                lin_coord(oh, sh, p.IH, h0, h1, lh);
                lin_coord(ow, sw, p.IW, w0, w1, lw);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int d = (k & 4) ? d1 : d0, h = (k & 2) ? h1 : h0, w = (k & 1) ? w1 : w0;
                    wt[u][k] = ((k & 4) ? ld : 1.f - ld) * ((k & 2) ? lh : 1.f - lh) * ((k & 1) ? lw : 1.f - lw);
                    q[u][k] = *(const uint4*)((const T*)p.x + ((((size_t)n * p.ID + d) * p.IH + h) * p.IW + w) * (size_t)p.ldx + s * KP);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u]) continue;
                float acc[KP];
#pragma unroll
                for (int j = 0; j < KP; ++j) acc[j] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {                          // same accumulation order as before: bit-identical results
                    float x[KP];
                    unpack16<T>(q[u][k], x);
#pragma unroll
                    for (int j = 0; j < KP; ++j) acc[j] += wt[u][k] * x[j];
                }
#pragma unroll
                for (int j = 0; j < KP; ++j) acc[j] = Elem<T>::rnd(acc[j]);
                *(uint4*)((T*)p.y + ((size_t)n * ovox + v + u * VL) * p.ldy + s * KP) = pack16<T>(acc);
#pragma unroll
                for (int j = 0; j < KP; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
            }
        }
    if (p.part) block_channel_sums<KP>(s1, s2, p.C, CV, VL, vl, s, active, p.part + ((size_t)n * gridDim.x + blockIdx.x) * p.C * 2);
}

// contribution range of input index i along one axis: outputs o whose i0(o)==i or i1(o)==i
__host__ __device__ __forceinline__ void up_range(int i, float scale, int O, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = O - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > O - 1) hi = O - 1;
}
__host__ __device__ __forceinline__ float up_weight(int o, int i, float scale, int I) {
    int i0, i1; float l1;
    lin_coord(o, scale, I, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    return w;
}

// per-axis candidate outputs of one input index and their interpolation weights (at most 6 for any scale >= 1/4)
struct UpAxis { int lo, n; float w[6]; };
__device__ __forceinline__ UpAxis up_axis(int i, float scale, int I, int O) {
    UpAxis a;
    int hi;
    up_range(i, scale, O, a.lo, hi);
    a.n = hi - a.lo + 1;
    if (a.n > 6) a.n = 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.w[k] = k < a.n ? up_weight(a.lo + k, i, scale, I) : 0.f;
    return a;
}

template <typename T>
__global__ void upsample_bwd_kernel(UpParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP;
    const int n = blockIdx.y;
    const size_t total = (size_t)p.ID * p.IH * p.IW * CV;
    const int ovox = p.OD * p.OH * p.OW;
    const float sd = p.OD > 1 ? (float)(p.ID - 1) / (float)(p.OD - 1) : 0.f;
    const float sh = p.OH > 1 ? (float)(p.IH - 1) / (float)(p.OH - 1) : 0.f;
    const float sw = p.OW > 1 ? (float)(p.IW - 1) / (float)(p.OW - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % CV); const int v = (int)(i / CV);
        const int iw = v % p.IW, ih = (v / p.IW) % p.IH, id = v / (p.IW * p.IH);
        const UpAxis ad = up_axis(id, sd, p.ID, p.OD), ah = up_axis(ih, sh, p.IH, p.OH), aw = up_axis(iw, sw, p.IW, p.OW);
        float acc[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) acc[j] = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (ad.w[a] == 0.f) continue;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if (ah.w[b] == 0.f) continue;
                const float wdh = ad.w[a] * ah.w[b];
                const size_t rowo = (size_t)n * ovox + ((size_t)(ad.lo + a) * p.OH + (ah.lo + b)) * p.OW;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (aw.w[c] == 0.f) continue;
                    float g[KP];
                    unpack16<T>(*(const uint4*)((const T*)p.y + (rowo + aw.lo + c) * p.ldy + s * KP), g);
                    const float wt = wdh * aw.w[c];
#pragma unroll
                    for (int j = 0; j < KP; ++j) acc[j] += wt * g[j];
                }
            }
        }
        *(uint4*)((T*)p.dx + ((size_t)n * p.ID * p.IH * p.IW + v) * p.lddx + s * KP) = pack16<T>(acc);
    }
}

// ---- second-generation trilinear kernels (round 2).  The first versions were VALU / latency bound at 18-24 % of the HBM
// rate: three runtime integer divisions per output vector in forward, and in backward 18 float weight evaluations per
// input voxel plus a load inside a data-dependent branch per tap (each waited for on its own).
//
// Forward: a thread keeps its channel slot and walks the block's output range with incrementally updated (od, oh, ow);
// corner offsets are 32-bit element offsets from one base pointer; the 8-corner accumulation keeps the original order
// (k = 0..7, fused multiply-add), as packed f32 pairs (v_pk_fma_f32): results are bit-identical to upsample_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd2_kernel(UpParams p) {
    constexpr int KP = Elem<T>::KP;
    const int CV = p.C / KP, VL = 256 / CV;
    const int vl = threadIdx.x / CV, s = threadIdx.x % CV;
    const bool active = vl < VL;
    const int n = blockIdx.y;
    const int ovox = p.OD * p.OH * p.OW;
    const int per_blk = (ovox + gridDim.x - 1) / gridDim.x;
    const int v0 = blockIdx.x * per_blk, v1 = min(ovox, v0 + per_blk);
    const float sd = p.OD > 1 ? (float)(p.ID - 1) / (float)(p.OD - 1) : 0.f;
    const float sh = p.OH > 1 ? (float)(p.IH - 1) / (float)(p.OH - 1) : 0.f;
    const float sw = p.OW > 1 ? (float)(p.IW - 1) / (float)(p.OW - 1) : 0.f;
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const T* xin = (const T*)p.x + (size_t)n * p.ID * p.IH * p.IW * p.ldx + s * KP;
    T* yout = (T*)p.y + (size_t)n * ovox * p.ldy + s * KP;
    if (active && v0 + vl < v1) {
        int v = v0 + vl;
        int ow = v % p.OW, oh = (v / p.OW) % p.OH, od = v / (p.OW * p.OH);
        for (; v < v1; v += VL) {
            int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
            lin_coord(od, sd, p.ID, d0, d1, ld);
            lin_coord(oh, sh, p.IH, h0, h1, lh);
            lin_coord(ow, sw, p.IW, w0, w1, lw);
            const uint32_t r00 = (uint32_t)(d0 * p.IH + h0) * p.IW, r01 = (uint32_t)(d0 * p.IH + h1) * p.IW;
            const uint32_t r10 = (uint32_t)(d1 * p.IH + h0) * p.IW, r11 = (uint32_t)(d1 * p.IH + h1) * p.IW;
            const uint32_t rows[4] = {r00, r01, r10, r11};
            uint4 q[8];
            float wt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                q[k] = *(const uint4*)(xin + (size_t)((rows[k >> 1] + (uint32_t)((k & 1) ? w1 : w0)) * (uint32_t)p.ldx));
                wt[k] = ((k & 4) ? ld : 1.f - ld) * ((k & 2) ? lh : 1.f - lh) * ((k & 1) ? lw : 1.f - lw);
            }
            float acc[KP];
            if (sizeof(T) == 2) {
                f32x2_t a2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a2[j] = f32x2_t{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t wq[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                    const f32x2_t w2 = {wt[k], wt[k]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2_t x2 = {__uint_as_float(wq[j] << 16), __uint_as_float(wq[j] & 0xffff0000u)};
                        a2[j] = __builtin_elementwise_fma(w2, x2, a2[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[2 * j] = a2[j][0]; acc[2 * j + 1] = a2[j][1]; }
            } else {
#pragma unroll
                for (int j = 0; j < KP; ++j) acc[j] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float x[KP];
                    unpack16<T>(q[k], x);
#pragma unroll
                    for (int j = 0; j < KP; ++j) acc[j] = fmaf(wt[k], x[j], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < KP; ++j) acc[j] = Elem<T>::rnd(acc[j]);
            *(uint4*)(yout + (size_t)v * p.ldy) = pack16<T>(acc);
#pragma unroll
            for (int j = 0; j < KP; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
            ow += VL;
            while (ow >= p.OW) { ow -= p.OW; if (++oh == p.OH) { oh = 0; ++od; } }
        }
    }
    if (p.part) block_channel_sums<KP>(s1, s2, p.C, CV, VL, vl, s, active, p.part + ((size_t)n * gridDim.x + blockIdx.x) * p.C * 2);
}

// Backward: per-axis tables in LDS (for every input index: first contributing output, count <= 6, weights -- exactly
// up_axis()'s values with the leading / trailing zero candidates trimmed), built once per block.  A thread then visits its
// taps in the same ascending (a, b, c) order as upsample_bwd_kernel (zero-weight candidates it skipped contributed
// nothing), so results are bit-identical; the innermost loop is unrolled to the block's maximal W count and its loads go
// through a buffer descriptor (out-of-range -> zeros, no branch), so NW loads are in flight per (a, b).
constexpr int UP_MAXW = 12;      // contributing outputs per input index: <= 2 / scale + 2, i.e. up to 4x up-sampling (the aux head)
struct UpEnt { int lo, n; float w[UP_MAXW]; };
template <typename T, int NW>
__device__ __forceinline__ void up_bwd_rows(const UpParams& p, const UpEnt* td, const UpEnt* th, const UpEnt* tw, int n, uint32_t i0, uint32_t total,
                                            uint32_t stride) {
    constexpr int KP = Elem<T>::KP;
    const uint32_t CV = (uint32_t)p.C / KP;
    const uint32_t ovox = (uint32_t)(p.OD * p.OH * p.OW);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.y + (size_t)n * ovox * p.ldy), 0,
                                                                        ovox * (uint32_t)p.ldy * (uint32_t)sizeof(T), 0x00020000);
    for (uint32_t i = i0; i < total; i += stride) {
        const uint32_t v = i / CV, s = i - v * CV;
        const uint32_t iw = v % (uint32_t)p.IW, t2 = v / (uint32_t)p.IW, ih = t2 % (uint32_t)p.IH, id = t2 / (uint32_t)p.IH;
        const UpEnt* ed = td + id; const UpEnt* eh = th + ih; const UpEnt* ew = tw + iw;
        const int na = ed->n, nb = eh->n, nc = ew->n, wlo = ew->lo;
        float wc[NW];
#pragma unroll
        for (int c = 0; c < NW; ++c) wc[c] = c < nc ? ew->w[c] : 0.f;
        float acc[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) acc[j] = 0.f;
        for (int a = 0; a < na; ++a) {
            const float wa = ed->w[a];
            for (int b = 0; b < nb; ++b) {
                const float wdh = wa * eh->w[b];
                const uint32_t rowo = ((uint32_t)(ed->lo + a) * (uint32_t)p.OH + (uint32_t)(eh->lo + b)) * (uint32_t)p.OW + (uint32_t)wlo;
                uint4 q[NW];
#pragma unroll
                for (int c = 0; c < NW; ++c) {
                    const uint32_t off = c < nc ? ((rowo + c) * (uint32_t)p.ldy + s * KP) * (uint32_t)sizeof(T) : 0xFFFFFFFFu;
                    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    q[c] = make_uint4(r[0], r[1], r[2], r[3]);
                }
#pragma unroll
                for (int c = 0; c < NW; ++c) {
                    float g[KP];
                    unpack16<T>(q[c], g);
                    const float wt = wdh * wc[c];
#pragma unroll
                    for (int j = 0; j < KP; ++j) acc[j] = fmaf(wt, g[j], acc[j]);
                }
            }
        }
        *(uint4*)((T*)p.dx + ((size_t)n * p.ID * p.IH * p.IW + v) * p.lddx + s * KP) = pack16<T>(acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd2_kernel(UpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    UpEnt* tab = (UpEnt*)smem;                                   // [ID] [IH] [IW]
    __shared__ int nwmax;
    if (threadIdx.x == 0) nwmax = 0;
    __syncthreads();
    const float sd = p.OD > 1 ? (float)(p.ID - 1) / (float)(p.OD - 1) : 0.f;
    const float sh = p.OH > 1 ? (float)(p.IH - 1) / (float)(p.OH - 1) : 0.f;
    const float sw = p.OW > 1 ? (float)(p.IW - 1) / (float)(p.OW - 1) : 0.f;
    for (int t = threadIdx.x; t < p.ID + p.IH + p.IW; t += 256) {
        const int axis = t < p.ID ? 0 : (t < p.ID + p.IH ? 1 : 2);
        const int i = axis == 0 ? t : (axis == 1 ? t - p.ID : t - p.ID - p.IH);
        const float sc = axis == 0 ? sd : (axis == 1 ? sh : sw);
        const int I = axis == 0 ? p.ID : (axis == 1 ? p.IH : p.IW), O = axis == 0 ? p.OD : (axis == 1 ? p.OH : p.OW);
        int lo, hi;
        up_range(i, sc, O, lo, hi);
        constexpr int CAND = 16;                                  // candidate window; the launcher rejects scales that need more
        float cw[CAND];
        int first = CAND, last = -1;
#pragma unroll
        for (int k = 0; k < CAND; ++k) {
            cw[k] = lo + k <= hi ? up_weight(lo + k, i, sc, I) : 0.f;
            if (cw[k] != 0.f) { if (first == CAND) first = k; last = k; }
        }
        UpEnt e;
        e.lo = lo + (first == CAND ? 0 : first);
        e.n = last < 0 ? 0 : min(last - first + 1, UP_MAXW);
#pragma unroll
        for (int k = 0; k < UP_MAXW; ++k) {
            float wv = 0.f;
#pragma unroll
            for (int m = 0; m < CAND; ++m) if (m == first + k && m <= last) wv = cw[m];
            e.w[k] = wv;
        }
        tab[t] = e;
        if (axis == 2) atomicMax(&nwmax, e.n);
    }
    __syncthreads();
    const UpEnt* td = tab; const UpEnt* th = tab + p.ID; const UpEnt* tw = tab + p.ID + p.IH;
    const uint32_t CV = (uint32_t)p.C / Elem<T>::KP;
    const uint32_t total = (uint32_t)(p.ID * p.IH * p.IW) * CV;
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    if (nwmax <= 4) up_bwd_rows<T, 4>(p, td, th, tw, blockIdx.y, i0, total, stride);
    else if (nwmax <= 6) up_bwd_rows<T, 6>(p, td, th, tw, blockIdx.y, i0, total, stride);
    else up_bwd_rows<T, UP_MAXW>(p, td, th, tw, blockIdx.y, i0, total, stride);
}

// ---- third-generation backward (round 4): the row reduction along w is done ONCE per output row and shared by a 2 x 2 (d, h) tile of input voxels.
// upsample_bwd2 gathers 4 x 4 x 4 output vectors per input vector (906 MB of L1 requests for 226 MB of data at 96^3 x 64 channels: 124-143 us).
// Here a thread owns (iw, 8 channels) of a 2 x 2 tile of (id, ih): it walks the output rows (od, oh) the tile receives from (about 5.5 x 5.5),
// reduces each row along w (t = sum_c wc * dy[od, oh, wlo + c]: NT coalesced 16-byte buffer loads whose row offset is a wave-uniform SGPR -- all 8
// channel groups of a voxel are read by adjacent lanes) and adds (wd * wh) * t to the four tile voxels through a per-row weight table: ~30 loads per
// output vector instead of 64, no integer division or 64-bit address arithmetic in the loop.  Association differs from upsample_bwd2
// ((wd * wh) * sum_c wc g  vs  sum of (wd * wh * wc) g): results agree to f32 rounding, not bit for bit.  The weights come from up_weight -- the same
// values as everywhere.  96^3 x 64 channels 124 -> 86 us, 48^3 x 128 43 -> 29 us (tools/bench_up.py).
// Measured and not kept: (i) the output-gradient tile staged through LDS per 8-channel chunk (16-byte accesses to 128-byte lines: request-rate bound,
// 465 us); (ii) the InstanceNorm backward tail of the producing block fused into this kernel (the tail is linear in its two tensors, so the row
// reduction ran on g and x separately): 268-274 us against 153 + 93 for the two launches -- twice the loads and registers (202 VGPRs, two waves per
// SIMD) cost more than the 226 MB write + read the fusion saves; deeper load pipelines (3-5 rows in flight) 92-236 us: the loop is issue-bound.
struct UpInParams {
    const void* g; int ldg;        // output gradient, channels-last on the (OD, OH, OW) grid
    void* dx; int lddx;
    int N, ID, IH, IW, OD, OH, OW, C;
};
#ifndef UP4_SL
#define UP4_SL 2
#endif
constexpr int UP4_ROWS = 16;       // candidate output rows of a pair of input indices (up_range of the first .. of the second): <= 2 * 2 / scale + 6
template <int NT>
__global__ __launch_bounds__(256, 3) void upsample_bwd4_kernel(UpInParams p, int nbx) {
    typedef bf16_t T;
    constexpr int CPT = 8, NWD = CPT / 2;                        // channels per thread (one 16-byte vector), 32-bit words of it
    constexpr int SL = UP4_SL;                                   // row slots: SL - 1 rows of loads in flight behind the one being reduced
    __shared__ float Wd[UP4_ROWS][2], Wh[UP4_ROWS][2];
    __shared__ float4 W4[UP4_ROWS * UP4_ROWS];                   // row (a, b) -> (wd0 wh0, wd0 wh1, wd1 wh0, wd1 wh1)
    __shared__ int span[4];                                      // od0, rd, oh0, rh
    const int tid = threadIdx.x, n = blockIdx.y;
    const int CV = p.C / CPT, tiles_h = (p.IH + 1) / 2;
    // XCD-aware order: linear block b runs on XCD b % 8; every XCD gets a contiguous run of (tile, lane block) items so that the output rows shared
    // by neighbouring tiles are fetched into one L2
    int L;
    {
        const int nt = gridDim.x, b = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = b & 7, k = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int bx = L % nbx, tile = L / nbx;
    const int id0 = (tile / tiles_h) * 2, ih0 = (tile % tiles_h) * 2;
    const float sd = p.OD > 1 ? (float)(p.ID - 1) / (float)(p.OD - 1) : 0.f;
    const float sh = p.OH > 1 ? (float)(p.IH - 1) / (float)(p.OH - 1) : 0.f;
    const float sw = p.OW > 1 ? (float)(p.IW - 1) / (float)(p.OW - 1) : 0.f;
    // dense weight tables of the tile: row r of axis d = output od_lo + r, column i = input id0 + i (zero where the output does not reach the input);
    // wave 0 builds d, wave 1 builds h; the span is trimmed to the rows with a non-zero weight
    if (tid < 128) {
        const int axis = tid >> 6, r = tid & 63;
        const int i0 = axis ? ih0 : id0, I = axis ? p.IH : p.ID, O = axis ? p.OH : p.OD;
        const float sc = axis ? sh : sd;
        int lo, hi, lo2, hi2;
        up_range(i0, sc, O, lo, hi);
        if (i0 + 1 < I) { up_range(i0 + 1, sc, O, lo2, hi2); if (hi2 > hi) hi = hi2; }
        const int o = lo + r;
        float w0 = 0.f, w1 = 0.f;
        if (r < UP4_ROWS && o <= hi) { w0 = up_weight(o, i0, sc, I); if (i0 + 1 < I) w1 = up_weight(o, i0 + 1, sc, I); }
        const unsigned long long nz = __ballot(w0 != 0.f || w1 != 0.f);
        const int first = nz ? __builtin_ctzll(nz) : 0, last = nz ? 63 - __builtin_clzll(nz) : -1;
        if (hi - lo + 1 > UP4_ROWS) __builtin_trap();            // the launcher admits scales whose candidate window fits
        if (r >= first && r <= last) { float (*Wt)[2] = axis ? Wh : Wd; Wt[r - first][0] = w0; Wt[r - first][1] = w1; }
        if (r == 0) { span[axis * 2] = lo + first; span[axis * 2 + 1] = last - first + 1; }
    }
    __syncthreads();
    const int od0 = span[0], rd = span[1], oh0 = span[2], rh = span[3];
    const int nrow = rd * rh;
    if (tid < nrow) { const int a = tid / rh, b = tid - a * rh; W4[tid] = make_float4(Wd[a][0] * Wh[b][0], Wd[a][0] * Wh[b][1], Wd[a][1] * Wh[b][0], Wd[a][1] * Wh[b][1]); }
    __syncthreads();
    const int flat = bx * 256 + tid;
    if (flat >= p.IW * CV) return;
    const int iw = flat / CV, s = flat - iw * CV;
    // this thread's w entry: first contributing output and NT weights
    int wlo;
    float wc[NT];
    {
        int lo, hi;
        up_range(iw, sw, p.OW, lo, hi);
        int first = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (first < 0 && lo + k <= hi && up_weight(lo + k, iw, sw, p.IW) != 0.f) first = k;
        wlo = lo + (first < 0 ? 0 : first);
#pragma unroll
        for (int c = 0; c < NT; ++c) wc[c] = (first >= 0 && wlo + c <= hi) ? up_weight(wlo + c, iw, sw, p.IW) : 0.f;
        if (first >= 0 && wlo + NT <= hi && up_weight(wlo + NT, iw, sw, p.IW) != 0.f) __builtin_trap();      // the launcher sized NT (up_max_count)
    }
    // operand loads: buffer resources on the sample's tensors, per-lane byte offset of tap c (constant for the whole loop: a tap past the entry re-reads
    // tap 0 with weight 0), wave-uniform row offset in an SGPR -> no address arithmetic per load
    const uint32_t ovox = (uint32_t)(p.OD * p.OH * p.OW);
    const uint32_t grow = (uint32_t)p.ldg * 2u;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.g + (size_t)n * ovox * p.ldg), 0, ovox * grow, 0x00020000);
    uint32_t goff[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const uint32_t v = (uint32_t)(wlo + (wc[c] != 0.f ? c : 0));
        goff[c] = v * grow + (uint32_t)s * 16u;
    }
    float acc[4][CPT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < CPT; ++k) acc[i][k] = 0.f;
    uint4 gq[SL][NT];
    int ia = 0, ib = 0;                                          // (a, b) of the next row to issue
    auto issue = [&](int slot) {
        const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane(((od0 + ia) * p.OH + (oh0 + ib)) * p.OW);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(grs, goff[c], row * grow, 0);
            gq[slot][c] = make_uint4(q[0], q[1], q[2], q[3]);
        }
        if (++ib == rh) { ib = 0; ++ia; }
    };
#pragma unroll
    for (int h = 0; h < SL - 1; ++h) if (h < nrow) issue(h);
    for (int r = 0; r < nrow; r += SL) {
#pragma unroll
        for (int h = 0; h < SL; ++h) {                           // static register slots: row r + h lives in slot h
            const int rr = r + h;
            if (rr >= nrow) break;
            if (rr + SL - 1 < nrow) issue((h + SL - 1) % SL);
            f32x2_t tg[NWD];
#pragma unroll
            for (int j = 0; j < NWD; ++j) tg[j] = f32x2_t{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const f32x2_t w2 = {wc[c], wc[c]};
                const uint32_t wq[4] = {gq[h][c].x, gq[h][c].y, gq[h][c].z, gq[h][c].w};
#pragma unroll
                for (int j = 0; j < NWD; ++j) {
                    const f32x2_t v2 = {__uint_as_float(wq[j] << 16), __uint_as_float(wq[j] & 0xffff0000u)};
                    tg[j] = __builtin_elementwise_fma(w2, v2, tg[j]);
                }
            }
            float t[CPT];
#pragma unroll
            for (int j = 0; j < NWD; ++j) { t[2 * j] = tg[j][0]; t[2 * j + 1] = tg[j][1]; }
            const float4 w4 = W4[rr];                            // a zero weight adds exactly nothing (fma(0, t, acc) == acc for finite t)
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < CPT; ++k) acc[i][k] = fmaf(wv[i], t[k], acc[i][k]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = id0 + (i >> 1), ih = ih0 + (i & 1);
        if (id < p.ID && ih < p.IH)
            *(uint4*)((T*)p.dx + ((size_t)((n * p.ID + id) * p.IH + ih) * p.IW + iw) * p.lddx + s * CPT) = pack16<T>(acc[i]);
    }
}
// largest number of contributing outputs of any input index of one axis (the arithmetic of the kernel: lin_coord is not contracted on either side)
static int up_max_count(int I, int O) {
    const float sc = O > 1 ? (float)(I - 1) / (float)(O - 1) : 0.f;
    int m = 1;
    for (int i = 0; i < I; ++i) {
        int lo, hi;
        up_range(i, sc, O, lo, hi);
        int first = -1, last = -2;
        for (int o = lo; o <= hi; ++o)
            if (up_weight(o, i, sc, I) != 0.f) { if (first < 0) first = o; last = o; }
        if (last - first + 1 > m) m = last - first + 1;
    }
    return m;
}
static int up_pair_window(int I, int O) {                        // widest candidate window (up_range) of an aligned pair of input indices
    const float sc = O > 1 ? (float)(I - 1) / (float)(O - 1) : 0.f;
    int m = 1;
    for (int i = 0; i < I; i += 2) {
        int lo, hi, lo2, hi2;
        up_range(i, sc, O, lo, hi);
        if (i + 1 < I) { up_range(i + 1, sc, O, lo2, hi2); if (hi2 > hi) hi = hi2; }
        if (hi - lo + 1 > m) m = hi - lo + 1;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------ stem: Conv3d(1 -> C, 3x3x3)
// thread = one voxel: its 27 neighbours live in registers, the (C,1,27) f32 weights are read through the scalar
// cache (uniform addresses -> s_load), so the inner loop is pure v_fmac with SGPR operands.
template <typename T, int C>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = (float*)smem;                                 // [256][C+1] for the stats transpose
    const float* __restrict__ w = p.w;
    const int n = blockIdx.y;
    const int vox = p.D * p.H * p.W;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const bool ok = v < vox;
    float xv[27];
    {
        const int x0 = v % p.W, y0 = (v / p.W) % p.H, z0 = v / (p.W * p.H);
        const float* xin = p.x + (size_t)n * vox;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int z = z0 + tap / 9 - 1, y = y0 + (tap % 9) / 3 - 1, x = x0 + tap % 3 - 1;
            xv[tap] = (ok && z >= 0 && z < p.D && y >= 0 && y < p.H && x >= 0 && x < p.W) ? xin[((size_t)z * p.H + y) * p.W + x] : 0.f;
        }
    }
    constexpr int KP = Elem<T>::KP;
    T* o = (T*)p.y + ((size_t)n * vox + (ok ? v : 0)) * p.ldy;
#pragma unroll 1
    for (int c8 = 0; c8 < C; c8 += KP) {
        float acc[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            float a = 0.f;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) a += xv[tap] * w[(c8 + j) * 27 + tap];
            acc[j] = Elem<T>::rnd(a);
            tile[threadIdx.x * (C + 1) + c8 + j] = ok ? acc[j] : 0.f;
        }
        if (ok) *(uint4*)(o + c8) = pack16<T>(acc);
    }
    __syncthreads();
    // channel-major reduction: thread (g, c) sums 256/G voxels
    constexpr int G = 256 / C;
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    float a = 0.f, b = 0.f;
    for (int r = g; r < 256; r += G) { const float t = tile[r * (C + 1) + c]; a += t; b += t * t; }
    __syncthreads();
    float* red = tile;                                          // [G][C][2]
    red[(g * C + c) * 2] = a; red[(g * C + c) * 2 + 1] = b;
    __syncthreads();
    if (threadIdx.x < C) {
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < G; ++k) { sa += red[(k * C + threadIdx.x) * 2]; sb += red[(k * C + threadIdx.x) * 2 + 1]; }
        float* pp = p.part + (((size_t)n * gridDim.x + blockIdx.x) * C + threadIdx.x) * 2;
        pp[0] = sa; pp[1] = sb;
    }
}

// MFMA variant for C <= 32 (every reference config: base_chan 32): out[c][v] = sum_tap w[c][tap] * x[v + off(tap)] as an
// exact-f32 v_mfma_f32_32x32x2_f32 chain (14 steps cover the 27 taps + one zero pad).  A = weights (row = channel, held
// in 14 registers per lane for the whole block), B = gathered input (column = voxel).  A lane ends up with 16 channels
// of one voxel = four 4-channel groups -> 8-byte (bf16) / 16-byte (f32) stores.  Block = 256 consecutive voxels.
#ifndef STEM_DIRECT_STORE
#define STEM_DIRECT_STORE 0
#endif
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(StemParams p) {
    __shared__ float tile[256 * 33];
    const int C = p.C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = lane & 31, kk = lane >> 5;
    const int n = blockIdx.y;
    const int vox = p.D * p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * vox), 0, (uint32_t)vox * 4u, 0x00020000);
    float wreg[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = 2 * s + kk;
        wreg[s] = (idx < C && tap < 27) ? p.w[idx * 27 + tap] : 0.f;
    }
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int vl = (wave * 2 + gi) * 32 + idx;               // voxel slot inside the block
        const int v = blockIdx.x * 256 + vl;
        const bool ok = v < vox;
        const int x0 = v % p.W, y0 = (v / p.W) % p.H, z0 = v / (p.W * p.H);
        // the 27 neighbours are v + a constant: one validity bit per (axis, offset) and a buffer load whose out-of-range offsets return zero replace a
        // coordinate triple + six compares + a 64-bit address per tap (the kernel was bound by that arithmetic, not by its MFMAs)
        const uint32_t okz = (z0 >= 1 ? 1u : 0u) | 2u | (z0 + 1 < p.D ? 4u : 0u), oky = (y0 >= 1 ? 1u : 0u) | 2u | (y0 + 1 < p.H ? 4u : 0u),
                       okx = (x0 >= 1 ? 1u : 0u) | 2u | (x0 + 1 < p.W ? 4u : 0u);
        const uint32_t tapok = ok ? (okz | (oky << 3) | (okx << 6)) : 0u;
        float xa[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int tap = 2 * s + kk;                           // kk selects one of two compile-time taps per step
            const int t0 = 2 * s, t1 = 2 * s + 1 < 27 ? 2 * s + 1 : 26;
            const int off0 = ((t0 / 9 - 1) * p.H + ((t0 % 9) / 3 - 1)) * p.W + (t0 % 3 - 1), off1 = ((t1 / 9 - 1) * p.H + ((t1 % 9) / 3 - 1)) * p.W + (t1 % 3 - 1);
            const uint32_t need0 = (1u << (t0 / 9)) | (8u << ((t0 % 9) / 3)) | (64u << (t0 % 3)), need1 = (1u << (t1 / 9)) | (8u << ((t1 % 9) / 3)) | (64u << (t1 % 3));
            const int off = kk ? off1 : off0;
            const uint32_t need = kk ? need1 : need0;
            const bool tok = tap < 27 && (tapok & need) == need;
            xa[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, tok ? (uint32_t)(v + off) * 4u : 0xFFFFFFFFu, 0, 0));
        }
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s], xa[s], acc, 0, 0, 0);
        T* o = (T*)p.y + ((size_t)n * vox + (ok ? v : 0)) * p.ldy;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int c0 = 8 * r4 + 4 * kk;                      // channels c0 .. c0+3 of voxel idx (cd_row32 layout)
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[j] = Elem<T>::rnd(acc[r4 * 4 + j]);
                tile[vl * 33 + c0 + j] = ok ? q[j] : 0.f;
            }
            if (STEM_DIRECT_STORE && ok && c0 < C) {
                if (sizeof(T) == 2) {
                    *(uint2*)(o + c0) = make_uint2(f2bf2(q[0], q[1]), f2bf2(q[2], q[3]));
                } else {
                    *(float4*)((float*)o + c0) = make_float4(q[0], q[1], q[2], q[3]);
                }
            }
        }
    }
    __syncthreads();
    if (!STEM_DIRECT_STORE) {
        // the block's 256 voxels x C channels leave through the statistics tile as 16-byte vectors, consecutive lanes -> consecutive addresses (the
        // accumulator layout gives a lane 4 channels of one voxel: 8-byte stores 64 bytes apart, a quarter of every 32-byte sector per instruction)
        constexpr int KP = Elem<T>::KP;
        const int cv = C / KP;                                   // vectors per voxel
        for (int i = threadIdx.x; i < 256 * cv; i += 256) {
            const int vl = i / cv, c0 = (i - vl * cv) * KP;
            const int v = blockIdx.x * 256 + vl;
            if (v < vox) *(uint4*)((T*)p.y + ((size_t)n * vox + v) * p.ldy + c0) = pack16<T>(tile + vl * 33 + c0);
        }
    }
    // channel-major reduction: thread (g, c) sums 32 voxels; then 8 partials per channel
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    float a = 0.f, b = 0.f;
    for (int r = g; r < 256; r += 8) { const float t = tile[r * 33 + c]; a += t; b += t * t; }
    __syncthreads();
    float* red = tile;                                          // [8][32][2]
    red[(g * 32 + c) * 2] = a; red[(g * 32 + c) * 2 + 1] = b;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < 8; ++k) { sa += red[(k * 32 + threadIdx.x) * 2]; sb += red[(k * 32 + threadIdx.x) * 2 + 1]; }
        float* pp = p.part + (((size_t)n * gridDim.x + blockIdx.x) * C + threadIdx.x) * 2;
        pp[0] = sa; pp[1] = sb;
    }
}

// Small weight gradients (stem 1->C conv, 1x1x1 head) as an exact-f32 MFMA reduction over voxels:
//     out[i][j] += sum_v A[v][i] * B[v][j],  i, j < 32,  v_mfma_f32_32x32x2_f32 (lane l: row/col l&31, voxel parity l>>5).
// One MFMA consumes two voxels (lane half kk = voxel parity); any voxel order works as long as A and B agree.  MODE 0 = stem: A = dY[v][c], B = x[v + off(tap)];
// MODE 1 = head: A = dlogits[k][v] (also summed for the bias gradient), B = features[v][c].
constexpr int SMALL_WS_ROW = 1024 + 32;
constexpr int SMALL_WS_BLOCKS = 512;

// BF (head, bf16 activations; round 6): the two products of the head's backward on v_mfma_f32_32x32x16_bf16 -- 16 voxels (weight gradient) / 16 classes (data
// gradient) per instruction of 32 cycles instead of 2 per 64-cycle exact-f32 instruction; the staged dlogits are rounded to bf16 as they enter the operand vector
// (round-to-nearest-even, the rounding every convolution's dY already carries in this mode), the features and the bias gradient are exact as before.
template <typename T, int MODE, bool BF = false>
__global__ __launch_bounds__(256) void small_wgrad_mfma_kernel(StemParams sp, HeadParams hp, int k0, int c0, float* __restrict__ ws) {
    // Chunks of 256 consecutive voxels are staged block-cooperatively with wide coalesced loads (per-lane 2/4-byte
    // gathers saturated the address unit): tb = channels-last operand [256][33] as f32; ta = head: dlogits planes
    // [32][257], stem: im2col of the image [256][29] built from loads that are contiguous across lanes.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tb = (float*)smem;                                    // [256][33]
    float* ta = tb + 256 * 33;                                   // [32][257] (head only)
    float* wl = ta + 32 * 257;                                   // [32][33] head weights W[k][c] (fused data gradient only)
    float (*red)[16][64] = (float (*)[16][64])smem;              // [4][16][64], aliases tb after the last chunk
    // MODE 1 with hp.dx: the data gradient of the head, dx[v][c] = sum_k dlogits[k][v] W[k][c], from the SAME staged dlogits tile (one more exact-f32 MFMA
    // chain per 32 voxels) -- head_bwd_data_kernel re-read the 184 MB of dlogits and spent 26 x 32 LDS-fed multiply-adds per voxel on the VALU (102 us)
    const bool with_dx = MODE == 1 && hp.dx != nullptr;
    if (with_dx) {
        for (int i = threadIdx.x; i < 32 * 33; i += 256) {
            const int k = i / 33, c = i - k * 33;
            wl[i] = (k < hp.K && c < hp.C) ? hp.w[k * hp.C + c] : 0.f;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = lane & 31, kk = lane >> 5;
    const int N = MODE == 0 ? sp.N : hp.N;
    const int V = MODE == 0 ? sp.D * sp.H * sp.W : hp.vox;
    const int per = (V + 255) / 256;                             // chunks per sample
    const long items = (long)N * per;
    constexpr int KP = Elem<T>::KP;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bias_acc = 0.f;
    // register-staged, software-pipelined: the next chunk's global loads are in flight during this chunk's MFMAs
    uint4 rx[32 / KP];                                           // this thread's voxel: 32 channels
    float4 rl[8];                                                // head: 8 x 4 dlogits of class plane tid>>3
    float ri[27];                                                // stem: the 27 image taps of this thread's voxel (coalesced over lanes)
    auto load_chunk = [&](long it) {
        const int n = (int)(it / per);
        const int v0 = (int)(it % per) * 256;
        const int v = v0 + tid;
        const T* src = MODE == 0 ? (const T*)sp.y + ((size_t)n * V + (v < V ? v : 0)) * sp.ldy : (const T*)hp.x + ((size_t)n * V + (v < V ? v : 0)) * hp.ldx;
        const int C = MODE == 0 ? sp.C : hp.C;                     // this launch covers channels [c0, c0 + 32)
#pragma unroll
        for (int c = 0; c < 32; c += KP) rx[c / KP] = (v < V && c0 + c < C) ? *(const uint4*)(src + c0 + c) : make_uint4(0, 0, 0, 0);
        if (MODE == 1) {
            const int k = tid >> 3, seg = tid & 7;
            const bool kok = k0 + k < hp.K;
            const float* row = hp.logits + ((size_t)n * hp.K + (kok ? k0 + k : 0)) * V + v0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int o = (i * 8 + seg) * 4;                 // 8 lanes cover 128 contiguous bytes of one plane
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kok) {
                    if ((V & 3) == 0) { if (v0 + o < V) q = *(const float4*)(row + o); }
                    else {
                        if (v0 + o < V) q.x = row[o];
                        if (v0 + o + 1 < V) q.y = row[o + 1];
                        if (v0 + o + 2 < V) q.z = row[o + 2];
                        if (v0 + o + 3 < V) q.w = row[o + 3];
                    }
                }
                rl[i] = q;
            }
        } else {
            const int x0 = v % sp.W, y0 = (v / sp.W) % sp.H, z0 = v / (sp.W * sp.H);
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(sp.x + (size_t)n * V), 0, (uint32_t)V * 4u, 0x00020000);
            const uint32_t okz = (z0 >= 1 ? 1u : 0u) | 2u | (z0 + 1 < sp.D ? 4u : 0u), oky = (y0 >= 1 ? 1u : 0u) | 2u | (y0 + 1 < sp.H ? 4u : 0u),
                           okx = (x0 >= 1 ? 1u : 0u) | 2u | (x0 + 1 < sp.W ? 4u : 0u);
            const uint32_t tapok = v < V ? (okz | (oky << 3) | (okx << 6)) : 0u;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {                  // neighbour = v + a constant; validity bits instead of a coordinate triple + six compares per tap
                const int off = ((tap / 9 - 1) * sp.H + ((tap % 9) / 3 - 1)) * sp.W + (tap % 3 - 1);
                const uint32_t need = (1u << (tap / 9)) | (8u << ((tap % 9) / 3)) | (64u << (tap % 3));
                ri[tap] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (tapok & need) == need ? (uint32_t)(v + off) * 4u : 0xFFFFFFFFu, 0, 0));
            }
        }
    };
    uint4 wfrag[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};      // BF: W[k][c] as the B operand of the data gradient, k = 16 s + 8 kk + j, c = idx
    if constexpr (MODE == 1 && BF) {
        if (with_dx) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s2 + 8 * kk + j;
                    wv[j] = (k < hp.K && idx < hp.C) ? hp.w[k * hp.C + idx] : 0.f;
                }
                wfrag[s2] = make_uint4(f2bf2(wv[0], wv[1]), f2bf2(wv[2], wv[3]), f2bf2(wv[4], wv[5]), f2bf2(wv[6], wv[7]));
            }
        }
    }
    if ((long)blockIdx.x < items) load_chunk(blockIdx.x);
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        // ---- registers -> LDS: tb [256][33] channels-last operand as f32; ta = head [32][257] planes / stem [256][29] taps
#pragma unroll
        for (int c = 0; c < 32; c += KP) {
            float f[KP];
            unpack16<T>(rx[c / KP], f);
#pragma unroll
            for (int j = 0; j < KP; ++j) tb[tid * 33 + c + j] = f[j];
        }
        if (MODE == 1) {
            const int k = tid >> 3, seg = tid & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float* d = ta + k * 257 + (i * 8 + seg) * 4;
                d[0] = rl[i].x; d[1] = rl[i].y; d[2] = rl[i].z; d[3] = rl[i].w;
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) ta[tid * 29 + tap] = ri[tap];
        }
        __syncthreads();
        if (it + gridDim.x < items) load_chunk(it + gridDim.x);
        if constexpr (MODE == 1 && BF) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {                        // this wave's 64 voxels: four k-steps of 16 (lane half kk: voxels 8 kk .. 8 kk + 7 of the step)
                const int vb = wave * 64 + 16 * s + 8 * kk;
                float af[8], xf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { af[j] = ta[idx * 257 + vb + j]; xf[j] = tb[(vb + j) * 33 + idx]; bias_acc += af[j]; }
                const uint4 a4 = make_uint4(f2bf2(af[0], af[1]), f2bf2(af[2], af[3]), f2bf2(af[4], af[5]), f2bf2(af[6], af[7]));
                const uint4 b4 = make_uint4(f2bf2(xf[0], xf[1]), f2bf2(xf[2], xf[3]), f2bf2(xf[4], xf[5]), f2bf2(xf[6], xf[7]));
                mma32<bf16_t>(acc, a4, b4);
            }
        } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int vl = wave * 64 + 2 * s + kk;
            float av, bv;
            if (MODE == 0) { av = tb[vl * 33 + idx]; bv = idx < 27 ? ta[vl * 29 + idx] : 0.f; }
            else { av = ta[idx * 257 + vl]; bv = tb[vl * 33 + idx]; bias_acc += av; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        }
        if (with_dx) {
            const int n = (int)(it / per), v0 = (int)(it % per) * 256;
#pragma unroll
            for (int g = 0; g < 2; ++g) {                        // this wave's 64 voxels as two 32-row tiles: D[v][c] = sum_k A[v][k] B[k][c]
                f32x16_t d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                if constexpr (MODE == 1 && BF) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        float af[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) af[j] = ta[(16 * s2 + 8 * kk + j) * 257 + wave * 64 + g * 32 + idx];
                        mma32<bf16_t>(d, make_uint4(f2bf2(af[0], af[1]), f2bf2(af[2], af[3]), f2bf2(af[4], af[5]), f2bf2(af[6], af[7])), wfrag[s2]);
                    }
                } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = 2 * s + kk;
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[k * 257 + wave * 64 + g * 32 + idx], wl[k * 33 + idx], d, 0, 0, 0);
                }
                }
                // through the wave's own rows of tb (no other wave reads them, and a wave's LDS accesses are ordered): accumulator layout (lane = channel,
                // register = voxel row) -> voxel-major, then 16-byte stores, consecutive lanes -> consecutive addresses
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[(wave * 64 + g * 32 + cd_row32(r, lane)) * 33 + idx] = d[r];
            }
            if (sizeof(T) == 2) {
                const int cv = hp.C / 8;                         // 16-byte vectors per voxel (1 .. 4)
                for (int i = lane; i < 64 * cv; i += 64) {
                    const int vl = i / cv, c0 = (i - vl * cv) * 8;
                    const int v = v0 + wave * 64 + vl;
                    if (v < V) *(uint4*)((T*)hp.dx + ((size_t)n * V + v) * hp.lddx + c0) = pack16<T>(tb + (wave * 64 + vl) * 33 + c0);
                }
            }
        }
        __syncthreads();
    }
    // block reduction of the four waves' 32x32 partials -> one workspace row per block (no atomics: 512 blocks hammering
    // the same 864 addresses serialised in L2 and dominated the kernel); small_wgrad_reduce_kernel sums the rows.
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    float (*bred)[32] = (float (*)[32])(smem + 4 * 16 * 64 * sizeof(float));
    if (MODE == 1) {
        bias_acc += __shfl_xor(bias_acc, 32, 64);
        if (lane < 32) bred[wave][lane] = bias_acc;
    }
    __syncthreads();
    float* row = ws + (size_t)blockIdx.x * SMALL_WS_ROW;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) row[r * 64 + lane] = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    } else if (wave == 1 && lane < 32) {
        row[1024 + lane] = MODE == 1 ? bred[0][lane] + bred[1][lane] + bred[2][lane] + bred[3][lane] : 0.f;
    }
}

// wave per element: sums the per-block rows (fixed order -> deterministic) and scatters into dW / db
template <int MODE>
__global__ __launch_bounds__(256) void small_wgrad_reduce_kernel(const float* __restrict__ ws, int nrows, StemParams sp, HeadParams hp, int k0, int c0) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= SMALL_WS_ROW) return;
    float v = 0.f;
    for (int r = lane; r < nrows; r += 64) v += ws[(size_t)r * SMALL_WS_ROW + e];
    v = wave_sum(v);
    if (lane != 0) return;
    if (e < 1024) {
        const int rr = e >> 6, ln = e & 63;
        const int i = cd_row32(rr, ln), j = ln & 31;
        if (MODE == 0) { if (c0 + i < sp.C && j < 27) sp.dw[(c0 + i) * 27 + j] = v; }
        else { if (k0 + i < hp.K && c0 + j < hp.C) hp.dw[(k0 + i) * hp.C + c0 + j] = v; }
    } else if (MODE == 1 && c0 == 0) {
        const int i = e - 1024;
        if (k0 + i < hp.K) hp.db[k0 + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ head: Conv3d(C -> K, 1x1x1) + bias
// logits are NCDHW f32 (the loss kernels reduce per (b, class) plane; the reference returns (B,C,D,H,W)).
template <typename T, int C>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = (float*)smem;                                    // [K][C] + bias[K]
    for (int i = threadIdx.x; i < p.K * C; i += 256) w[i] = p.w[i];
    for (int i = threadIdx.x; i < p.K; i += 256) w[p.K * C + i] = p.b[i];
    __syncthreads();
    const int n = blockIdx.y;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (size_t)p.vox) return;
    constexpr int KP = Elem<T>::KP;
    float f[C];
    const T* x = (const T*)p.x + ((size_t)n * p.vox + v) * p.ldx;
#pragma unroll
    for (int c = 0; c < C; c += KP) unpack16<T>(*(const uint4*)(x + c), f + c);
    for (int k = 0; k < p.K; ++k) {
        float a = w[p.K * C + k];
#pragma unroll
        for (int c = 0; c < C; ++c) a += f[c] * w[k * C + c];
        p.logits[((size_t)n * p.K + k) * p.vox + v] = a;
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void head_bwd_data_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = (float*)smem;
    for (int i = threadIdx.x; i < p.K * C; i += 256) w[i] = p.w[i];
    __syncthreads();
    const int n = blockIdx.y;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (size_t)p.vox) return;
    constexpr int KP = Elem<T>::KP;
    float f[C];
#pragma unroll
    for (int c = 0; c < C; ++c) f[c] = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const float g = p.logits[((size_t)n * p.K + k) * p.vox + v];
#pragma unroll
        for (int c = 0; c < C; ++c) f[c] += g * w[k * C + c];
    }
    T* dx = (T*)p.dx + ((size_t)n * p.vox + v) * p.lddx;
#pragma unroll
    for (int c = 0; c < C; c += KP) *(uint4*)(dx + c) = pack16<T>(f + c);
}

template <typename F> void set_smem(F k, size_t smem) {
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

}  // namespace

int rs_launch_stats_finalize(const float* part, int N, int nblk, int C, double cnt, float eps, int mode, int split, float* out, hipStream_t st) {
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((C + 31) / 32, N), dim3(1024), 0, st, part, nblk, C, cnt, eps, mode, split, out);
    return rs_check_launch();
}

int rs_elem_blocks(size_t items) {
    size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int rs_launch_in_bwd(const InBwdParams& p, int dtype, hipStream_t st) {
    const int KP = dtype == RS_F32 ? 4 : 8;
    if ((size_t)p.vox * (p.C / KP) >= 0xFFFFFFFFull) return RS_ERR_UNSUPPORTED;
    const int blocks = rs_elem_blocks((size_t)p.vox * (p.C / KP));
    if (dtype == RS_F32) hipLaunchKernelGGL(in_bwd_finalize_kernel<float>, dim3(blocks, p.N), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(in_bwd_finalize_kernel<bf16_t>, dim3(blocks, p.N), dim3(256), 0, st, p);
    return rs_check_launch();
}

int rs_launch_pool(const PoolParams& p, int dtype, int bwd, int blocks, hipStream_t st) {
    const int KP = dtype == RS_F32 ? 4 : 8;
    const int CV = p.C / KP;
    if (CV > 256) return RS_ERR_UNSUPPORTED;
    if (!bwd) {
        const size_t smem = (size_t)(256 / CV) * p.C * 2 * sizeof(float);
        if (dtype == RS_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(blocks, p.N), dim3(256), smem, st, p);
        else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(blocks, p.N), dim3(256), smem, st, p);
    } else {
        const int b = rs_elem_blocks((size_t)(p.D / 2) * (p.H / 2) * (p.W / 2) * CV);
        if (dtype == RS_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(b, p.N), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(b, p.N), dim3(256), 0, st, p);
    }
    return rs_check_launch();
}

int rs_launch_subsample(const PoolParams& p, int dtype, int bwd, int blocks, hipStream_t st) {
    const int KP = dtype == RS_F32 ? 4 : 8;
    const int CV = p.C / KP;
    if (CV > 256) return RS_ERR_UNSUPPORTED;
    if (!bwd) {
        const size_t smem = (size_t)(256 / CV) * p.C * 2 * sizeof(float);
        if (dtype == RS_F32) hipLaunchKernelGGL(subsample_fwd_kernel<float>, dim3(blocks, p.N), dim3(256), smem, st, p);
        else hipLaunchKernelGGL(subsample_fwd_kernel<bf16_t>, dim3(blocks, p.N), dim3(256), smem, st, p);
    } else {
        const int b = rs_elem_blocks((size_t)p.D * p.H * p.W * CV);
        if (dtype == RS_F32) hipLaunchKernelGGL(subsample_bwd_kernel<float>, dim3(b, p.N), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(subsample_bwd_kernel<bf16_t>, dim3(b, p.N), dim3(256), 0, st, p);
    }
    return rs_check_launch();
}

// trilinear backward on the row-sharing kernel (bf16); RS_ERR_UNSUPPORTED for scales its tile tables do not hold (the caller runs upsample_bwd2 then)
static int launch_upsample_bwd4(const void* g, int ldg, void* dx, int lddx, int N, int ID, int IH, int IW, int OD, int OH, int OW, int C, hipStream_t st) {
    static const bool off = getenv("RSUPER_UPSAMPLE_BWD4") && atoi(getenv("RSUPER_UPSAMPLE_BWD4")) == 0;
    if (off || (C % 8) || N > 65535) return RS_ERR_UNSUPPORTED;
    if ((size_t)OD * OH * OW * (size_t)ldg * 2 >= 0xFFFFFFFFull) return RS_ERR_UNSUPPORTED;                    // 32-bit byte offsets
    // the table sizes depend on the shape only: remembered per (ID, IH, IW, OD, OH, OW) (the host loops cost a few microseconds per call otherwise)
    struct Memo { int k[6]; int ok, nmax; };
    static thread_local Memo memo[8] = {};
    static thread_local int memo_next = 0;
    const int key[6] = {ID, IH, IW, OD, OH, OW};
    const Memo* hit = nullptr;
    for (int i = 0; i < 8 && !hit; ++i) if (memo[i].k[0] && !memcmp(memo[i].k, key, sizeof(key))) hit = &memo[i];
    if (!hit) {
        Memo& m = memo[memo_next++ & 7];
        memcpy(m.k, key, sizeof(key));
        m.ok = up_pair_window(ID, OD) <= UP4_ROWS && up_pair_window(IH, OH) <= UP4_ROWS;
        m.nmax = up_max_count(IW, OW);
        hit = &m;
    }
    if (!hit->ok) return RS_ERR_UNSUPPORTED;
    const int nmax = hit->nmax;
    if (nmax > 6) return RS_ERR_UNSUPPORTED;
    UpInParams p = {g, ldg, dx, lddx, N, ID, IH, IW, OD, OH, OW, C};
    const int nbx = (IW * (C / 8) + 255) / 256;
    const unsigned blocks = (unsigned)(((ID + 1) / 2) * ((IH + 1) / 2) * nbx);
    if (nmax <= 4) hipLaunchKernelGGL((upsample_bwd4_kernel<4>), dim3(blocks, N), dim3(256), 0, st, p, nbx);
    else hipLaunchKernelGGL((upsample_bwd4_kernel<6>), dim3(blocks, N), dim3(256), 0, st, p, nbx);
    return rs_check_launch();
}

int rs_launch_upsample(const UpParams& p, int dtype, int bwd, int blocks, hipStream_t st) {
    const int KP = dtype == RS_F32 ? 4 : 8;
    const int CV = p.C / KP;
    if (CV > 256) return RS_ERR_UNSUPPORTED;
    if (!bwd) {
        static const bool v1 = getenv("RSUPER_UPSAMPLE_V1") != nullptr;      // first-generation kernels (A/B, bit-identical results)
        const size_t smem = (size_t)(256 / CV) * p.C * 2 * sizeof(float);
        const bool small = (size_t)p.N * p.ID * p.IH * p.IW * p.ldx < 0x7FFFFFFFull;
        if (v1 || !small) {
            if (dtype == RS_F32) hipLaunchKernelGGL(upsample_fwd_kernel<float>, dim3(blocks, p.N), dim3(256), smem, st, p);
            else hipLaunchKernelGGL(upsample_fwd_kernel<bf16_t>, dim3(blocks, p.N), dim3(256), smem, st, p);
        } else {
            if (dtype == RS_F32) hipLaunchKernelGGL(upsample_fwd2_kernel<float>, dim3(blocks, p.N), dim3(256), smem, st, p);
            else hipLaunchKernelGGL(upsample_fwd2_kernel<bf16_t>, dim3(blocks, p.N), dim3(256), smem, st, p);
        }
    } else {
        static const bool v1 = getenv("RSUPER_UPSAMPLE_V1") != nullptr;
        // contributing outputs per input index < 2 * (O-1)/(I-1) + 2: the table kernel holds 12 (up to 4x: the aux head); the
        // first-generation kernel looks at 6 candidates (2x) and is kept for A/B runs only
        auto ratio = [](int O, int I) { return I > 1 ? (double)(O - 1) / (double)(I - 1) : (double)O; };
        const double rmax = fmax(ratio(p.OD, p.ID), fmax(ratio(p.OH, p.IH), ratio(p.OW, p.IW)));
        if (!v1 && 2.0 * rmax + 2.0 > (double)UP_MAXW + 1e-9) return RS_ERR_UNSUPPORTED;
        const int b = rs_elem_blocks((size_t)p.ID * p.IH * p.IW * CV);
        const size_t tab = (size_t)(p.ID + p.IH + p.IW) * sizeof(UpEnt);
        const bool small = (size_t)p.OD * p.OH * p.OW * p.ldy * (dtype == RS_F32 ? 4 : 2) < 0xFFFFFFFFull && (size_t)p.ID * p.IH * p.IW * CV < 0xFFFFFFFFull && tab <= 48 * 1024;
        if (!v1 && dtype == RS_BF16) {                           // row-sharing kernel (upsample_bwd4_kernel)
            const int rc = launch_upsample_bwd4(p.y, p.ldy, p.dx, p.lddx, p.N, p.ID, p.IH, p.IW, p.OD, p.OH, p.OW, p.C, st);
            if (rc != RS_ERR_UNSUPPORTED) return rc;
        }
        if (v1 || !small) {
            if (dtype == RS_F32) hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(b, p.N), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(b, p.N), dim3(256), 0, st, p);
        } else {
            if (dtype == RS_F32) hipLaunchKernelGGL(upsample_bwd2_kernel<float>, dim3(b, p.N), dim3(256), tab, st, p);
            else hipLaunchKernelGGL(upsample_bwd2_kernel<bf16_t>, dim3(b, p.N), dim3(256), tab, st, p);
        }
    }
    return rs_check_launch();
}

#define RS_DISPATCH_C(KERNEL, T, C, ...)                                                        \
    switch (C) {                                                                                \
        case 8: { auto k = KERNEL<T, 8>; set_smem(k, smem); hipLaunchKernelGGL(k, __VA_ARGS__); break; }   \
        case 16: { auto k = KERNEL<T, 16>; set_smem(k, smem); hipLaunchKernelGGL(k, __VA_ARGS__); break; } \
        case 32: { auto k = KERNEL<T, 32>; set_smem(k, smem); hipLaunchKernelGGL(k, __VA_ARGS__); break; } \
        case 64: { auto k = KERNEL<T, 64>; set_smem(k, smem); hipLaunchKernelGGL(k, __VA_ARGS__); break; } \
        default: return RS_ERR_UNSUPPORTED;                                                     \
    }

// Per-device scratch rows of the small weight-gradient kernels (allocated once, stream-ordered reuse; stem and head have
// their own so the two may overlap on different streams).
static float* small_ws(int which) {
    static float* ws[16][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!ws[dev][which] && hipMalloc((void**)&ws[dev][which], (size_t)SMALL_WS_BLOCKS * SMALL_WS_ROW * sizeof(float)) != hipSuccess) return nullptr;
    return ws[dev][which];
}

int rs_launch_stem(const StemParams& p, int dtype, int wgrad, hipStream_t st) {
    const int vox = p.D * p.H * p.W;
    if (!wgrad) {
        const size_t smem = (size_t)(256 * (p.C + 1)) * sizeof(float);
        dim3 grid((vox + 255) / 256, p.N);
        if (p.C <= 32 && (p.C % 4) == 0) {
            if (dtype == RS_F32) hipLaunchKernelGGL(stem_fwd_mfma_kernel<float>, grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL(stem_fwd_mfma_kernel<bf16_t>, grid, dim3(256), 0, st, p);
            return rs_check_launch();
        }
        if (dtype == RS_F32) { RS_DISPATCH_C(stem_fwd_kernel, float, p.C, grid, dim3(256), smem, st, p) }
        else { RS_DISPATCH_C(stem_fwd_kernel, bf16_t, p.C, grid, dim3(256), smem, st, p) }
    } else {
        HeadParams hp{};
        const long items = (long)p.N * ((vox + 255) / 256);
        dim3 grid((unsigned)(items < SMALL_WS_BLOCKS ? items : SMALL_WS_BLOCKS));
        float* ws = small_ws(0);
        if (!ws) return RS_ERR_LAUNCH;
        const size_t smem = (size_t)(256 * 33 + 256 * 29) * sizeof(float);
        for (int c0 = 0; c0 < p.C; c0 += 32) {                   // 32 output channels per pass (base_ch 64: two passes)
            if (dtype == RS_F32) hipLaunchKernelGGL((small_wgrad_mfma_kernel<float, 0>), grid, dim3(256), smem, st, p, hp, 0, c0, ws);
            else hipLaunchKernelGGL((small_wgrad_mfma_kernel<bf16_t, 0>), grid, dim3(256), smem, st, p, hp, 0, c0, ws);
            hipLaunchKernelGGL(small_wgrad_reduce_kernel<0>, dim3((SMALL_WS_ROW + 3) / 4), dim3(256), 0, st, ws, (int)grid.x, p, hp, 0, c0);
        }
    }
    return rs_check_launch();
}

int rs_launch_head(const HeadParams& p, int dtype, int which, hipStream_t st) {
    if (p.K > 256) return RS_ERR_UNSUPPORTED;                    // K * C + K floats of weights in LDS
    dim3 grid((p.vox + 255) / 256, p.N);
    if (which == 0) {
        const size_t smem = (size_t)(p.K * p.C + p.K) * sizeof(float);
        if (dtype == RS_F32) { RS_DISPATCH_C(head_fwd_kernel, float, p.C, grid, dim3(256), smem, st, p) }
        else { RS_DISPATCH_C(head_fwd_kernel, bf16_t, p.C, grid, dim3(256), smem, st, p) }
    } else if (which == 1) {
        const size_t smem = (size_t)(p.K * p.C) * sizeof(float);
        if (dtype == RS_F32) { RS_DISPATCH_C(head_bwd_data_kernel, float, p.C, grid, dim3(256), smem, st, p) }
        else { RS_DISPATCH_C(head_bwd_data_kernel, bf16_t, p.C, grid, dim3(256), smem, st, p) }
    } else {
        if (p.dx && (dtype != RS_BF16 || p.K > 32 || p.C > 32 || !p.w)) return RS_ERR_UNSUPPORTED;      // fused data gradient: one pass, bf16 activations
        StemParams sp{};
        const long items = (long)p.N * ((p.vox + 255) / 256);
        dim3 g2((unsigned)(items < SMALL_WS_BLOCKS ? items : SMALL_WS_BLOCKS));
        float* ws = small_ws(1);
        if (!ws) return RS_ERR_LAUNCH;
        for (int k0 = 0; k0 < p.K; k0 += 32)                    // 32 classes x 32 channels per pass (K = 42 in BASELINE config 5 -> 2 passes)
            for (int c0 = 0; c0 < p.C; c0 += 32) {
                const size_t smem = (size_t)(256 * 33 + 32 * 257 + 32 * 33) * sizeof(float);
                if (dtype == RS_F32) {
                    auto kf = small_wgrad_mfma_kernel<float, 1>;
                    (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                    hipLaunchKernelGGL(kf, g2, dim3(256), smem, st, sp, p, k0, c0, ws);
                } else {
                    static const bool bf = !(getenv("RSUPER_HEAD_BF16") && atoi(getenv("RSUPER_HEAD_BF16")) == 0);      // =0: the exact-f32 chains of rounds 4-5 (A/B)
                    auto kf = bf ? small_wgrad_mfma_kernel<bf16_t, 1, true> : small_wgrad_mfma_kernel<bf16_t, 1, false>;
                    (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                    hipLaunchKernelGGL(kf, g2, dim3(256), smem, st, sp, p, k0, c0, ws);
                }
                hipLaunchKernelGGL(small_wgrad_reduce_kernel<1>, dim3((SMALL_WS_ROW + 3) / 4), dim3(256), 0, st, ws, (int)g2.x, sp, p, k0, c0);
            }
    }
    return rs_check_launch();
}
