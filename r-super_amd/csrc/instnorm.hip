// Stand-alone InstanceNorm3d(affine = False) [+ ReLU] on channels-last fp32 activations, forward and backward -- the bare
// norm layers of MedFormer's attention stages (rsuper_train/model/dim3/medformer_utils.py:117-118, :160; the norm + act of the
// 1x1x1 / depthwise ConvNormAct members, conv_layers.py:46-51).  (The dense 3x3x3 convolutions never need this: their norm +
// ReLU is fused into the conv kernels' staging.)  HBM-bound: forward 2 reads + 1 write, backward 4 reads + 1 write per element.
//
//   stats  mode 0: part[n][row][c] = (sum x, sum x^2)                              -> rsuper_stats_finalize mode 0 -> (mean, rstd)
//          mode 1: part[n][row][c] = (sum g, sum g * x_hat),  g = dy * [x_hat > 0 if relu]   -> mode 1 -> (m1, m2)
//   apply  mode 0: y  = x_hat, or max(x_hat, 0) with relu                          x_hat = (x - mean) * rstd
//          mode 1: dx = rstd * (g - m1 - x_hat * m2)
//          mode 2: y  = x * mr[1] + mr[0]        (per-(sample, channel) affine map: the squeeze-excite scaling and its gradient)
// Thread = (voxel, 4 consecutive channels); a block owns 64 channels (blockIdx.y) of one sample (blockIdx.z) and walks voxels
// with stride rows * 16; per-block partial rows + the existing fixed-order f64 finalize keep the reduction deterministic.
// ATen's reductions over the middle axes of a channels-last tensor ran at ~100 us per call (21 ms per MedFormer step).
#include "common.hpp"
#include "misc.hpp"

namespace {

constexpr int CN_CG = 64, CN_VPB = 16;

struct CnParams {
    const float* x; const float* dy; const float* mr; const float* gm;
    float* out; float* part;
    long vox; int C; int relu; int rows;
};

template <int MODE>
__global__ __launch_bounds__(256) void cnorm_stats_kernel(CnParams p) {
    __shared__ float red[4][2][CN_CG];
    const int n = blockIdx.z;
    const int cv = threadIdx.x & 15, vl = threadIdx.x >> 4;
    const int c = blockIdx.y * CN_CG + cv * 4;
    const bool cok = c < p.C;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 mu = s1, rs = s1;
    if (MODE == 1 && cok) {
        const float* m = p.mr + ((size_t)n * p.C + c) * 2;
        mu = make_float4(m[0], m[2], m[4], m[6]); rs = make_float4(m[1], m[3], m[5], m[7]);
    }
    if (cok) {
        const float* xb = p.x + (size_t)n * p.vox * p.C + c;
        const float* gb = MODE == 1 ? p.dy + (size_t)n * p.vox * p.C + c : nullptr;
        for (long v = (long)blockIdx.x * CN_VPB + vl; v < p.vox; v += (long)gridDim.x * CN_VPB) {
            const float4 q = *(const float4*)(xb + (size_t)v * p.C);
            if (MODE == 0) {
                s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
                s2.x += q.x * q.x; s2.y += q.y * q.y; s2.z += q.z * q.z; s2.w += q.w * q.w;
            } else {
                float4 g = *(const float4*)(gb + (size_t)v * p.C);
                const float4 xh = make_float4((q.x - mu.x) * rs.x, (q.y - mu.y) * rs.y, (q.z - mu.z) * rs.z, (q.w - mu.w) * rs.w);
                if (p.relu) { g.x = xh.x > 0.f ? g.x : 0.f; g.y = xh.y > 0.f ? g.y : 0.f; g.z = xh.z > 0.f ? g.z : 0.f; g.w = xh.w > 0.f ? g.w : 0.f; }
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                s2.x += g.x * xh.x; s2.y += g.y * xh.y; s2.z += g.z * xh.z; s2.w += g.w * xh.w;
            }
        }
    }
    // lanes that share a channel vector sit 16 and 32 lanes apart inside a wave: two butterfly steps, then the four waves through LDS
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
    }
    if ((threadIdx.x & 63) < 16) {
        *(float4*)&red[threadIdx.x >> 6][0][cv * 4] = s1;
        *(float4*)&red[threadIdx.x >> 6][1][cv * 4] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2 * CN_CG) {
        const int k = threadIdx.x / CN_CG, cc = threadIdx.x - k * CN_CG;
        if (blockIdx.y * CN_CG + cc < p.C)
            p.part[(((size_t)n * gridDim.x + blockIdx.x) * p.C + blockIdx.y * CN_CG + cc) * 2 + k] =
                (red[0][k][cc] + red[1][k][cc]) + (red[2][k][cc] + red[3][k][cc]);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void cnorm_apply_kernel(CnParams p) {
    const int n = blockIdx.z;
    const int cv = threadIdx.x & 15, vl = threadIdx.x >> 4;
    const int c = blockIdx.y * CN_CG + cv * 4;
    if (c >= p.C) return;
    const float* m = p.mr + ((size_t)n * p.C + c) * 2;
    const float4 mu = make_float4(m[0], m[2], m[4], m[6]), rs = make_float4(m[1], m[3], m[5], m[7]);
    float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f), g2 = g1;
    if (MODE == 1) {
        const float* gm = p.gm + ((size_t)n * p.C + c) * 2;
        g1 = make_float4(gm[0], gm[2], gm[4], gm[6]); g2 = make_float4(gm[1], gm[3], gm[5], gm[7]);
    }
    const float* xb = p.x + (size_t)n * p.vox * p.C + c;
    const float* gb = MODE == 1 ? p.dy + (size_t)n * p.vox * p.C + c : nullptr;
    float* ob = p.out + (size_t)n * p.vox * p.C + c;
    for (long v = (long)blockIdx.x * CN_VPB + vl; v < p.vox; v += (long)gridDim.x * CN_VPB) {
        const float4 q = *(const float4*)(xb + (size_t)v * p.C);
        const float4 xh = make_float4((q.x - mu.x) * rs.x, (q.y - mu.y) * rs.y, (q.z - mu.z) * rs.z, (q.w - mu.w) * rs.w);
        float4 o;
        if (MODE == 2) {                                     // per-(sample, channel) affine map: out = x * mr[1] + mr[0]
            o = make_float4(fmaf(q.x, rs.x, mu.x), fmaf(q.y, rs.y, mu.y), fmaf(q.z, rs.z, mu.z), fmaf(q.w, rs.w, mu.w));
        } else if (MODE == 0) {
            o = p.relu ? make_float4(fmaxf(xh.x, 0.f), fmaxf(xh.y, 0.f), fmaxf(xh.z, 0.f), fmaxf(xh.w, 0.f)) : xh;
        } else {
            float4 g = *(const float4*)(gb + (size_t)v * p.C);
            if (p.relu) { g.x = xh.x > 0.f ? g.x : 0.f; g.y = xh.y > 0.f ? g.y : 0.f; g.z = xh.z > 0.f ? g.z : 0.f; g.w = xh.w > 0.f ? g.w : 0.f; }
            o = make_float4(rs.x * (g.x - g1.x - xh.x * g2.x), rs.y * (g.y - g1.y - xh.y * g2.y), rs.z * (g.z - g1.z - xh.z * g2.z),
                            rs.w * (g.w - g1.w - xh.w * g2.w));
        }
        *(float4*)(ob + (size_t)v * p.C) = o;
    }
}

// Small volumes (the 12^3 / 6^3 stages and the 27-token semantic maps: most of MedFormer's norm calls): one block per
// (sample, 16-channel group: 4 channel vectors x 64 voxel lanes -- round 3; 64-channel groups left 10 blocks on the chip for 320 channels and 14
// dependent trips per thread) does statistics, finalize and apply in ONE launch -- the three-launch path is pure launch latency there.
// Same arithmetic: f32 partial sums per thread, fixed-order f64 combination, mean / rstd (or m1 / m2) in f32.
constexpr int CS_CG = 16, CS_VPB = 64;
template <int MODE>
__global__ __launch_bounds__(256) void cnorm_small_kernel(CnParams p, float eps, float* mr_out) {
    __shared__ double red[4][2][CS_CG];
    __shared__ float fin[2][CS_CG];
    const int n = blockIdx.y;
    const int cv = threadIdx.x & 3, vl = threadIdx.x >> 2;
    const int c = blockIdx.x * CS_CG + cv * 4;
    const bool cok = c < p.C;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu;
    if (MODE == 1 && cok) {
        const float* m = p.mr + ((size_t)n * p.C + c) * 2;
        mu = make_float4(m[0], m[2], m[4], m[6]); rs = make_float4(m[1], m[3], m[5], m[7]);
    }
    const float* xb = p.x + (size_t)n * p.vox * p.C + c;
    const float* gb = MODE == 1 ? p.dy + (size_t)n * p.vox * p.C + c : nullptr;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (cok)
        for (long v = vl; v < p.vox; v += CS_VPB) {
            const float4 q = *(const float4*)(xb + (size_t)v * p.C);
            if (MODE == 0) {
                s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
                s2.x += q.x * q.x; s2.y += q.y * q.y; s2.z += q.z * q.z; s2.w += q.w * q.w;
            } else {
                float4 g = *(const float4*)(gb + (size_t)v * p.C);
                const float4 xh = make_float4((q.x - mu.x) * rs.x, (q.y - mu.y) * rs.y, (q.z - mu.z) * rs.z, (q.w - mu.w) * rs.w);
                if (p.relu) { g.x = xh.x > 0.f ? g.x : 0.f; g.y = xh.y > 0.f ? g.y : 0.f; g.z = xh.z > 0.f ? g.z : 0.f; g.w = xh.w > 0.f ? g.w : 0.f; }
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                s2.x += g.x * xh.x; s2.y += g.y * xh.y; s2.z += g.z * xh.z; s2.w += g.w * xh.w;
            }
        }
    double d1[4] = {s1.x, s1.y, s1.z, s1.w}, d2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { d1[j] += __shfl_xor(d1[j], o, 64); d2[j] += __shfl_xor(d2[j], o, 64); }
    if ((threadIdx.x & 63) < 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[threadIdx.x >> 6][0][cv * 4 + j] = d1[j]; red[threadIdx.x >> 6][1][cv * 4 + j] = d2[j]; }
    __syncthreads();
    if (threadIdx.x < CS_CG) {
        const int cc = threadIdx.x;
        const double a = (red[0][0][cc] + red[1][0][cc]) + (red[2][0][cc] + red[3][0][cc]);
        const double b = (red[0][1][cc] + red[1][1][cc]) + (red[2][1][cc] + red[3][1][cc]);
        const double cnt = (double)p.vox;
        float f0, f1;
        if (MODE == 0) {
            const double mean = a / cnt;
            double var = b / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            f0 = (float)mean; f1 = (float)(1.0 / sqrt(var + (double)eps));
            if (blockIdx.x * CS_CG + cc < p.C) { float* o = mr_out + ((size_t)n * p.C + blockIdx.x * CS_CG + cc) * 2; o[0] = f0; o[1] = f1; }
        } else {
            f0 = (float)(a / cnt); f1 = (float)(b / cnt);
        }
        fin[0][cc] = f0; fin[1][cc] = f1;
    }
    __syncthreads();
    if (!cok) return;
    const float4 a0 = *(const float4*)&fin[0][cv * 4], a1 = *(const float4*)&fin[1][cv * 4];
    if (MODE == 0) { mu = a0; rs = a1; }
    float* ob = p.out + (size_t)n * p.vox * p.C + c;
    for (long v = vl; v < p.vox; v += CS_VPB) {
        const float4 q = *(const float4*)(xb + (size_t)v * p.C);
        const float4 xh = make_float4((q.x - mu.x) * rs.x, (q.y - mu.y) * rs.y, (q.z - mu.z) * rs.z, (q.w - mu.w) * rs.w);
        float4 o;
        if (MODE == 0) {
            o = p.relu ? make_float4(fmaxf(xh.x, 0.f), fmaxf(xh.y, 0.f), fmaxf(xh.z, 0.f), fmaxf(xh.w, 0.f)) : xh;
        } else {
            float4 g = *(const float4*)(gb + (size_t)v * p.C);
            if (p.relu) { g.x = xh.x > 0.f ? g.x : 0.f; g.y = xh.y > 0.f ? g.y : 0.f; g.z = xh.z > 0.f ? g.z : 0.f; g.w = xh.w > 0.f ? g.w : 0.f; }
            o = make_float4(rs.x * (g.x - a0.x - xh.x * a1.x), rs.y * (g.y - a0.y - xh.y * a1.y), rs.z * (g.z - a0.z - xh.z * a1.z),
                            rs.w * (g.w - a0.w - xh.w * a1.w));
        }
        *(float4*)(ob + (size_t)v * p.C) = o;
    }
}

}  // namespace

int rs_cnorm_rows(long vox) {
    long b = (vox + CN_VPB * 8 - 1) / (CN_VPB * 8);
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

int rs_launch_cnorm_stats(const float* x, const float* dy, const float* mr, float* part, int N, long vox, int C, int relu, int mode, hipStream_t st) {
    const int rows = rs_cnorm_rows(vox);
    CnParams p = {x, dy, mr, nullptr, nullptr, part, vox, C, relu, rows};
    dim3 grid(rows, (C + CN_CG - 1) / CN_CG, N);
    if (mode == 0) hipLaunchKernelGGL(cnorm_stats_kernel<0>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(cnorm_stats_kernel<1>, grid, dim3(256), 0, st, p);
    return rs_check_launch();
}

int rs_launch_cnorm_apply(const float* x, const float* dy, const float* mr, const float* gm, float* out, int N, long vox, int C, int relu, int mode,
                          hipStream_t st) {
    CnParams p = {x, dy, mr, gm, out, nullptr, vox, C, relu, 0};
    long bx = (vox + CN_VPB - 1) / CN_VPB;
    if (bx > 2048) bx = 2048;
    dim3 grid((unsigned)bx, (C + CN_CG - 1) / CN_CG, N);
    if (mode == 0) hipLaunchKernelGGL(cnorm_apply_kernel<0>, grid, dim3(256), 0, st, p);
    else if (mode == 1) hipLaunchKernelGGL(cnorm_apply_kernel<1>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(cnorm_apply_kernel<2>, grid, dim3(256), 0, st, p);
    return rs_check_launch();
}

namespace {

// Squeeze-excite excitation (conv_layers.py:159-174) on one (N, C) vector of channel means: s = sigmoid(W2 relu(W1 m + b1) + b2), r = C / ratio
// hidden units.  Two small kernels, a wave per output with the lanes striding the reduction (coalesced weight rows, shuffle reduce) -- a
// single block per sample took 269 us (one memory latency per output, 80 outputs in sequence per wave); spread over 32-40 blocks it takes a few us.
// The ATen form was ~8 forward and ~14 backward one-element-sized launches per block of the network (36 blocks per step).
// Writes the affine table (0, s) that `cnorm_apply_kernel<2>` scales x with, and the hidden activations for the backward pass.
__global__ __launch_bounds__(1024) void se_hidden_fwd_kernel(const float* __restrict__ ms, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             float* __restrict__ hbuf, int C, int r) {
    // grid (N, r / 16): one hidden unit per wave, lanes stride the C-long reduction (coalesced W1 row)
    const int n = blockIdx.x, lane = threadIdx.x & 63, j = blockIdx.y * 16 + (threadIdx.x >> 6);
    if (j >= r) return;
    const float* row = w1 + (size_t)j * C;
    // four independent partial sums, every trip's eight loads issued before the first multiply-add (round 3: one load per trip in flight was
    // 20-32 memory latencies in sequence -- 8.9 us for a 100 KB reduction)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = lane;
    for (; c + 192 < C; c += 256) {
        const float w0 = row[c], w1_ = row[c + 64], w2_ = row[c + 128], w3 = row[c + 192];
        const float m0 = ms[((size_t)n * C + c) * 2], m1 = ms[((size_t)n * C + c + 64) * 2], m2 = ms[((size_t)n * C + c + 128) * 2],
                    m3 = ms[((size_t)n * C + c + 192) * 2];
        a0 = fmaf(w0, m0, a0); a1 = fmaf(w1_, m1, a1); a2 = fmaf(w2_, m2, a2); a3 = fmaf(w3, m3, a3);
    }
    for (; c < C; c += 64) a0 = fmaf(row[c], ms[((size_t)n * C + c) * 2], a0);
    float a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) hbuf[(size_t)n * r + j] = fmaxf(a + b1[j], 0.f);
}

__global__ __launch_bounds__(1024) void se_gate_fwd_kernel(const float* __restrict__ hbuf, const float* __restrict__ w2, const float* __restrict__ b2,
                                                           float* __restrict__ tab, int C, int r) {
    // grid (N, C / 64): four gates per wave, lanes stride the r-long reduction (coalesced W2 row); writes the affine table (0, s)
    const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the four gates of a wave advance together (their loads are independent: one latency per trip instead of four)
    const int c0 = blockIdx.y * 64 + wv * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int jj = lane; jj < r; jj += 64) {
        const float hv = hbuf[(size_t)n * r + jj];
        float wv4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv4[k] = c0 + k < C ? w2[(size_t)(c0 + k) * r + jj] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fmaf(wv4[k], hv, a[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
        if (lane == 0 && c0 + k < C) {
            float* t = tab + ((size_t)n * C + c0 + k) * 2;
            t[0] = 0.f; t[1] = 1.f / (1.f + expf(-(a[k] + b2[c0 + k])));
        }
    }
}

// Backward, stage 1: dz2 = ds * s (1 - s) with ds = vox * gm[., 1] (gm = per-channel means of (dy, dy * x) from the statistics kernel),
// dh = dz2 W2, dz1 = dh [h > 0].  Grid (N, r / 32): a block owns 32 hidden units; its 1024 threads are 32 units x 32 slices of the C-long
// reduction (the lanes of a unit row read 128 contiguous bytes of a W2 row), slices summed through LDS in a fixed order.
__global__ __launch_bounds__(1024) void se_excite_bwd_hidden_kernel(const float* __restrict__ gm, const float* __restrict__ stab,
                                                                    const float* __restrict__ hbuf, const float* __restrict__ w2, float vox,
                                                                    float* __restrict__ dz1buf, int C, int r) {
    extern __shared__ float se_sm[];
    float* dz2 = se_sm;                 // [C]
    float* red = se_sm + C;             // [32][33]
    const int n = blockIdx.x, j0 = blockIdx.y * 32;
    for (int c = threadIdx.x; c < C; c += 1024) {
        const float sv = stab[((size_t)n * C + c) * 2 + 1];
        dz2[c] = gm[((size_t)n * C + c) * 2 + 1] * vox * sv * (1.f - sv);
    }
    __syncthreads();
    const int jl = threadIdx.x & 31, sl = threadIdx.x >> 5, j = j0 + jl;
    float a = 0.f;
    if (j < r) {                                                 // four loads in flight per trip (was one: C / 32 latencies in sequence, 16.5 us)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int c = sl;
        for (; c + 96 < C; c += 128) {
            const float w0 = w2[(size_t)c * r + j], w1_ = w2[(size_t)(c + 32) * r + j], w2_ = w2[(size_t)(c + 64) * r + j], w3 = w2[(size_t)(c + 96) * r + j];
            a0 = fmaf(dz2[c], w0, a0); a1 = fmaf(dz2[c + 32], w1_, a1); a2 = fmaf(dz2[c + 64], w2_, a2); a3 = fmaf(dz2[c + 96], w3, a3);
        }
        for (; c < C; c += 32) a0 = fmaf(dz2[c], w2[(size_t)c * r + j], a0);
        a = (a0 + a1) + (a2 + a3);
    }
    red[sl * 33 + jl] = a;
    __syncthreads();
    if (threadIdx.x < 32 && j0 + threadIdx.x < r) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k * 33 + threadIdx.x];
        dz1buf[(size_t)n * r + j0 + threadIdx.x] = hbuf[(size_t)n * r + j0 + threadIdx.x] > 0.f ? t : 0.f;
    }
}

// Backward: dm[n][c] = sum_j dz1[n][j] W1[j][c] into the affine table (dm / vox, s) of dx = dy * s + dm / vox.  Grid (N, C / 32): 32 channels x 32
// slices of the r-long reduction per block, slices summed through LDS in a fixed order.
__global__ __launch_bounds__(1024) void se_dm_kernel(const float* __restrict__ dz1buf, const float* __restrict__ w1, const float* __restrict__ stab, float vox,
                                                     float* __restrict__ tab, int C, int r) {
    __shared__ float red[32 * 33];
    const int n = blockIdx.x, cl = threadIdx.x & 31, sl = threadIdx.x >> 5, c = blockIdx.y * 32 + cl;
    float a = 0.f;
    if (c < C)
        for (int jj = sl; jj < r; jj += 32) a = fmaf(dz1buf[(size_t)n * r + jj], w1[(size_t)jj * C + c], a);
    red[sl * 33 + cl] = a;
    __syncthreads();
    if (threadIdx.x < 32 && blockIdx.y * 32 + threadIdx.x < C) {
        const int cc = blockIdx.y * 32 + threadIdx.x;
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k * 33 + threadIdx.x];
        float* o = tab + ((size_t)n * C + cc) * 2;
        o[0] = t / vox; o[1] = stab[((size_t)n * C + cc) * 2 + 1];
    }
}

// Backward, stage 2: weight / bias gradients summed over the samples in a fixed order -- dW2[c][j] = sum_n dz2[n][c] h[n][j],
// dW1[j][c] = sum_n dz1[n][j] m[n][c], db2 = sum_n dz2, db1 = sum_n dz1.  Element-parallel (threads walk the contiguous axis of each output).
__global__ __launch_bounds__(256) void se_excite_wgrad_kernel(const float* __restrict__ gm, const float* __restrict__ stab, const float* __restrict__ hbuf,
                                                              const float* __restrict__ ms, const float* __restrict__ dz1buf,
                                                              float vox, int N, int C, int r, float* __restrict__ dw1, float* __restrict__ db1,
                                                              float* __restrict__ dw2, float* __restrict__ db2) {
    const long total = (long)C * r;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        {   // dW2 (C, r)
            const int c = (int)(i / r), j = (int)(i - (long)c * r);
            float a = 0.f;
            for (int n = 0; n < N; ++n) {
                const float sv = stab[((size_t)n * C + c) * 2 + 1];
                a = fmaf(gm[((size_t)n * C + c) * 2 + 1] * vox * sv * (1.f - sv), hbuf[(size_t)n * r + j], a);
            }
            dw2[i] = a;
        }
        {   // dW1 (r, C)
            const int j = (int)(i / C), c = (int)(i - (long)j * C);
            float a = 0.f;
            for (int n = 0; n < N; ++n) a = fmaf(dz1buf[(size_t)n * r + j], ms[((size_t)n * C + c) * 2], a);
            dw1[i] = a;
        }
        if (i < C) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) {
                const float sv = stab[((size_t)n * C + i) * 2 + 1];
                a += gm[((size_t)n * C + i) * 2 + 1] * vox * sv * (1.f - sv);
            }
            db2[i] = a;
        }
        if (i < r) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) a += dz1buf[(size_t)n * r + i];
            db1[i] = a;
        }
    }
}

// Channels-last <-> planar re-layout of a logits-like f32 tensor: [N][vox][C] (C % 4 == 0, C <= 64, the first K channels real) <->
// [N][K][vox].  The deep-supervision head of MedFormer up-samples channels-last and hands (N, K, D, H, W) planes to the loss
// (medformer.py:190-194); ATen's permute copy ran this at 0.45 TB/s.  A block moves 128 voxels x C channels through LDS: 16-byte row
// accesses on the channels-last side, 256-byte plane runs on the planar side.  dir 0: planar <- channels-last; dir 1: channels-last
// <- planar with the C - K padding channels written as zero.
__global__ __launch_bounds__(256) void cl_planar_kernel(const float* __restrict__ src, float* __restrict__ dst, long vox, int C, int K, int dir) {
    __shared__ float tile[128][65];
    const int n = blockIdx.y;
    const long v0 = (long)blockIdx.x * 128;
    const int nv = (int)min((long)128, vox - v0);
    const int CV = C >> 2;
    const float* cl_src = src + ((size_t)n * vox + v0) * C;
    float* cl_dst = dst + ((size_t)n * vox + v0) * C;
    if (dir == 0) {
        for (int i = threadIdx.x; i < nv * CV; i += 256) {
            const int v = i / CV, c = (i - v * CV) * 4;
            const float4 q = *(const float4*)(cl_src + (size_t)v * C + c);
            tile[v][c] = q.x; tile[v][c + 1] = q.y; tile[v][c + 2] = q.z; tile[v][c + 3] = q.w;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K * 128; i += 256) {
            const int k = i >> 7, v = i & 127;
            if (v < nv) dst[((size_t)n * K + k) * vox + v0 + v] = tile[v][k];
        }
    } else {
        for (int i = threadIdx.x; i < K * 128; i += 256) {
            const int k = i >> 7, v = i & 127;
            if (v < nv) tile[v][k] = src[((size_t)n * K + k) * vox + v0 + v];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nv * CV; i += 256) {
            const int v = i / CV, c = (i - v * CV) * 4;
            const float4 q = make_float4(c < K ? tile[v][c] : 0.f, c + 1 < K ? tile[v][c + 1] : 0.f, c + 2 < K ? tile[v][c + 2] : 0.f,
                                         c + 3 < K ? tile[v][c + 3] : 0.f);
            *(float4*)(cl_dst + (size_t)v * C + c) = q;
        }
    }
}

}  // namespace

// SEBlock forward in one host call: channel means (statistics + finalize), excitation, y = x * s.  scratch: part (cnorm rows), ms (N, C, 2),
// tab (N, C, 2) = (0, s) and hbuf (N, r) are kept by the caller for the backward pass.
int rs_launch_se_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* part, float* ms, float* tab,
                         float* hbuf, float* y, int N, long vox, int C, int r, hipStream_t st) {
    if (C > 8192 || r > 8192 || r < 1) return RS_ERR_UNSUPPORTED;
    const int rows = rs_cnorm_rows(vox);
    int rc = rs_launch_cnorm_stats(x, nullptr, nullptr, part, N, vox, C, 0, 0, st);
    if (rc != RS_OK) return rc;
    rc = rs_launch_stats_finalize(part, N, rows, C, (double)vox, 0.f, 1, 0, ms, st);
    if (rc != RS_OK) return rc;
    hipLaunchKernelGGL(se_hidden_fwd_kernel, dim3(N, (r + 15) / 16), dim3(1024), 0, st, (const float*)ms, w1, b1, hbuf, C, r);
    hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(N, (C + 63) / 64), dim3(1024), 0, st, (const float*)hbuf, w2, b2, tab, C, r);
    rc = rs_check_launch();
    if (rc != RS_OK) return rc;
    return rs_launch_cnorm_apply(x, nullptr, tab, nullptr, y, N, vox, C, 0, 2, st);
}

// SEBlock backward in one host call.  ident: (N, C, 2) table of (0, 1) (the statistics kernel then yields the means of (dy, dy * x)).
int rs_launch_se_backward(const float* x, const float* dy, const float* ident, const float* w1, const float* w2, const float* ms, const float* tab,
                          const float* hbuf, float* part, float* gm, float* dz1buf, float* tab2, float* dx, float* dw1, float* db1, float* dw2,
                          float* db2, int N, long vox, int C, int r, hipStream_t st) {
    if (C > 8192 || r > 8192 || r < 1) return RS_ERR_UNSUPPORTED;
    const int rows = rs_cnorm_rows(vox);
    int rc = rs_launch_cnorm_stats(x, dy, ident, part, N, vox, C, 0, 1, st);
    if (rc != RS_OK) return rc;
    rc = rs_launch_stats_finalize(part, N, rows, C, (double)vox, 0.f, 1, 0, gm, st);
    if (rc != RS_OK) return rc;
    hipLaunchKernelGGL(se_excite_bwd_hidden_kernel, dim3(N, (r + 31) / 32), dim3(1024), (size_t)(C + 32 * 33) * sizeof(float), st, (const float*)gm, tab,
                       hbuf, w2, (float)vox, dz1buf, C, r);
    const long total = (long)C * r;
    hipLaunchKernelGGL(se_excite_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)gm, tab, hbuf, ms,
                       (const float*)dz1buf, (float)vox, N, C, r, dw1, db1, dw2, db2);
    hipLaunchKernelGGL(se_dm_kernel, dim3(N, (C + 31) / 32), dim3(1024), 0, st, (const float*)dz1buf, w1, tab, (float)vox, tab2, C, r);
    rc = rs_check_launch();
    if (rc != RS_OK) return rc;
    return rs_launch_cnorm_apply(dy, nullptr, tab2, nullptr, dx, N, vox, C, 0, 2, st);
}

int rs_launch_cl_planar(const float* src, float* dst, int N, long vox, int C, int K, int dir, hipStream_t st) {
    if (C > 64 || (C & 3) || K > C || K < 1) return RS_ERR_UNSUPPORTED;
    const long bx = (vox + 127) / 128;
    if (bx > 0x7FFFFFFFL) return RS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cl_planar_kernel, dim3((unsigned)bx, N), dim3(256), 0, st, src, dst, vox, C, K, dir);
    return rs_check_launch();
}

// One-launch path for small volumes.  mode 0: out = norm(x) (+relu), mr_out = (mean, rstd); mode 1: out = dx from (x, dy, mr).
int rs_launch_cnorm_small(const float* x, const float* dy, const float* mr, float* out, float* mr_out, int N, long vox, int C, int relu, float eps,
                          int mode, hipStream_t st) {
    CnParams p = {x, dy, mr, nullptr, out, nullptr, vox, C, relu, 0};
    dim3 grid((C + 15) / 16, N);
    if (mode == 0) hipLaunchKernelGGL(cnorm_small_kernel<0>, grid, dim3(256), 0, st, p, eps, mr_out);
    else hipLaunchKernelGGL(cnorm_small_kernel<1>, grid, dim3(256), 0, st, p, eps, mr_out);
    return rs_check_launch();
}
