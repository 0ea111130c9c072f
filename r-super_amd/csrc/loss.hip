// Loss reductions of the R-Super step (gfx950, HBM-bound): one pass over logits + byte masks producing the
// per-(sample, class) partial sums every loss term needs, with wave-shuffle + LDS block reductions, and the
// matching one-pass backward that writes / accumulates d(loss)/d(logits).
//
// Restates, for planes of V voxels (rsuper_train/training/losses_foundation.py):
//   masked BCE-with-logits          :945-956   S  = sum bce(x,t) * k
//   DiceLossMultiClass              :541-607   A  = sum sig(x)*k ; B = sum sig(x)*t*k ; Cn = sum t*k
//                                              (TP = B, FP = A - B, FN = Cn - B because k is binary)
//   soft volume of volume_loss_basic:329,:367  A with k = dilated segment mask
//   ball_loss weighted BCE          :1793-1811 F1 = sum bce*k*w1 (GWRP foreground weights)
//                                              F2 = sum bce*k*(1-w2) (background = outside dilated pseudo mask)
// The tiny (B,C) algebra on these sums (adaptive alpha, clamps, means) stays in autograd on the host side so
// d(alpha) flows exactly as in the reference; this file supplies d(sum)/d(x) per voxel.
#include "common.hpp"
#include "misc.hpp"
#include "loss.hpp"

namespace {

// One exp per voxel: e = exp(-|x|) gives both sigmoid(x) = (x >= 0 ? 1 : e) / (1 + e) and the softplus term
// log1p(e) of ATen's binary_cross_entropy_with_logits: max(x,0) - x*t + log1p(exp(-|x|)).
__device__ __forceinline__ void sig_bce(float x, float t, float& sg, float& bce) {
    const float e = __expf(-fabsf(x));
    const float r = __frcp_rn(1.f + e);
    sg = x >= 0.f ? r : e * r;
    bce = fmaxf(x, 0.f) - x * t + __logf(1.f + e);             // e in (0, 1]: 1 + e is exact enough (abs error < 1.2e-7 per voxel)
}
__device__ __forceinline__ float sigmoidf(float x) { float s, b; sig_bce(x, 0.f, s, b); return s; }

// grid: (blocks, planes).  p.pblk (calculate_loss, round 6): every block stores its six sums to pblk[plane][block][6], plane_sums_reduce_kernel adds them in block
// order; p.sums (the one-launch form of rsuper_plane_partials_fwd / _fwd2): [planes][6] doubles, pre-zeroed, accumulated with f64 atomics.
__global__ __launch_bounds__(256) void plane_partials_fwd_kernel(PlaneParams p) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * p.V;
    const float* x = p.x + (size_t)plane * p.xstride;
    // target: one byte per voxel (p.t) or the dataset's bit-packed form (p.tpk: np.packbits along the class axis, dataset_abdomenatlas_UFO.py:955 --
    // class c of sample b is bit 7 - (c & 7) of byte plane b * tP + (c >> 3)); either way "byte & tm8 != 0"
    const uint8_t* t = p.tpk ? p.tpk + ((size_t)(plane / p.tC) * p.tP + (size_t)((plane % p.tC) >> 3)) * p.V : (p.t ? p.t + base : nullptr);
    const uint32_t tm8 = p.tpk ? (0x80u >> ((plane % p.tC) & 7)) : 0xFFu;
    // unknown-voxel plane without a single unknown voxel (kflags from rsuper_plane_any of the UNdilated map): weight 1 everywhere, nothing to read
    const bool kskip = p.kinv && p.kflags && !p.kflags[plane];
    const uint8_t* k = (p.k && !kskip) ? p.k + base : nullptr;
    const int kinv = k ? p.kinv : 0;
    const float* w1 = p.w1 ? p.w1 + base : nullptr;
    const uint8_t* w2 = p.w2 ? p.w2 + base : nullptr;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool vec_ok = (p.V & 3) == 0 && (p.xstride & 3) == 0;   // 16-byte alignment of every plane
    // wide path: 16 voxels per thread and trip (four 16-byte logit loads, one 16-byte load per mask / weight plane) -- the 4-voxel trips below read the
    // byte planes with 4-byte loads (a quarter of the bytes per request): 84 us for 276 MB at 96^3 x 26 classes
    size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    // the 16-byte loads need 16-byte aligned bases too (a mask handed over as a view at an odd storage offset keeps the 4-voxel path)
    const bool wide_ok = vec_ok && (p.V & 15) == 0 && !w1 &&
                         ((((uintptr_t)p.x | (uintptr_t)p.t | (uintptr_t)p.tpk | (uintptr_t)p.k | (uintptr_t)p.w2) & 15) == 0);
    if (wide_ok) {
        auto term = [&](float xv, uint32_t tb, uint32_t kb, uint32_t w2b) {
            const float tt = (tb & tm8) ? 1.f : 0.f, kk = ((kb != 0u) != (kinv != 0)) ? 1.f : 0.f;
            float sg, b;
            sig_bce(xv, tt, sg, b);
            b *= kk;
            s[0] += b; s[1] += sg * kk; s[2] += sg * tt * kk; s[3] += tt * kk; s[5] += b * (w2b ? 0.f : 1.f);
        };
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < (size_t)p.V; i += (size_t)gridDim.x * 256 * 16) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *(const float4*)(x + i + 4 * u);
            const uint4 tq = t ? *(const uint4*)(t + i) : make_uint4(0, 0, 0, 0);
            const uint4 kq = k ? *(const uint4*)(k + i) : make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
            const uint4 wq = w2 ? *(const uint4*)(w2 + i) : make_uint4(0, 0, 0, 0);
            const uint32_t tw[4] = {tq.x, tq.y, tq.z, tq.w}, kw[4] = {kq.x, kq.y, kq.z, kq.w}, ww[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xv[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) term(xv[j], (tw[u] >> (8 * j)) & 0xFFu, (kw[u] >> (8 * j)) & 0xFFu, (ww[u] >> (8 * j)) & 0xFFu);
            }
        }
        i0 = (size_t)p.V;                                          // nothing left for the narrow loop
    }
    for (size_t i = i0; i < (size_t)p.V; i += (size_t)gridDim.x * 256 * 4) {
        float xv[4]; uint8_t tv[4] = {0, 0, 0, 0}, kv[4] = {1, 1, 1, 1}, w2v[4] = {0, 0, 0, 0}; float w1v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = vec_ok && i + 4 <= (size_t)p.V;
        if (full) {
            const float4 q = *(const float4*)(x + i);
            xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
            if (t) { const uchar4 u = *(const uchar4*)(t + i); tv[0] = u.x; tv[1] = u.y; tv[2] = u.z; tv[3] = u.w; }
            if (k) { const uchar4 u = *(const uchar4*)(k + i); kv[0] = u.x; kv[1] = u.y; kv[2] = u.z; kv[3] = u.w; }
            if (w2) { const uchar4 u = *(const uchar4*)(w2 + i); w2v[0] = u.x; w2v[1] = u.y; w2v[2] = u.z; w2v[3] = u.w; }
            if (w1) { const float4 u = *(const float4*)(w1 + i); w1v[0] = u.x; w1v[1] = u.y; w1v[2] = u.z; w1v[3] = u.w; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!full) {
                if (i + j >= (size_t)p.V) break;
                xv[j] = x[i + j];
                if (t) tv[j] = t[i + j];
                if (k) kv[j] = k[i + j];
                if (w2) w2v[j] = w2[i + j];
                if (w1) w1v[j] = w1[i + j];
            }
            const float tt = (tv[j] & tm8) ? 1.f : 0.f, kk = ((kv[j] != 0) != (kinv != 0)) ? 1.f : 0.f;
            float sg, b;
            sig_bce(xv[j], tt, sg, b);
            b *= kk;
            s[0] += b;
            s[1] += sg * kk;
            s[2] += sg * tt * kk;
            s[3] += tt * kk;
            s[4] += b * w1v[j];
            s[5] += b * (w2v[j] ? 0.f : 1.f);
        }
    }
    __shared__ float red[4][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float v = wave_sum(s[q]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double v = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
        if (p.pblk) p.pblk[((size_t)plane * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = v;      // fixed-order reduction follows (plane_sums_reduce_kernel)
        else atomicAdd(p.sums + (size_t)plane * 6 + threadIdx.x, v);
    }
}

// out[row][q] = (float) sum_b pblk[row][b][q], b ascending: the per-block sums of plane_partials_fwd_kernel added in a fixed order -- no float atomic is left on
// the training path, and neither is the zero-fill of the accumulator nor the f64 -> f32 conversion pass of the atomic form.
__global__ __launch_bounds__(256) void plane_sums_reduce_kernel(const double* __restrict__ pblk, int rows, int nb, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * 6) return;
    const int row = e / 6, q = e - row * 6;
    const double* src = pblk + (size_t)row * nb * 6 + q;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += src[(size_t)b * 6];
    out[e] = (float)s;
}

// dx = k * ( (gS + gF1*w1 + gF2*(1-w2)) * (sig - t) + sig*(1-sig) * (gA + gB*t) ),  g: [planes][6] floats
__global__ __launch_bounds__(256) void plane_partials_bwd_kernel(PlaneParams p) {
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * p.V;
    const float* x = p.x + (size_t)plane * p.xstride;
    float* dx = p.dx + (size_t)plane * p.xstride;
    const uint8_t* t = p.tpk ? p.tpk + ((size_t)(plane / p.tC) * p.tP + (size_t)((plane % p.tC) >> 3)) * p.V : (p.t ? p.t + base : nullptr);
    const uint32_t tm8 = p.tpk ? (0x80u >> ((plane % p.tC) & 7)) : 0xFFu;
    const bool kskip = p.kinv && p.kflags && !p.kflags[plane];
    const uint8_t* k = (p.k && !kskip) ? p.k + base : nullptr;
    const float* w1 = p.w1 ? p.w1 + base : nullptr;
    const uint8_t* w2 = p.w2 ? p.w2 + base : nullptr;
    const float gS = p.g[plane * 6], gA = p.g[plane * 6 + 1], gB = p.g[plane * 6 + 2], gF1 = p.g[plane * 6 + 4], gF2 = p.g[plane * 6 + 5];
    const bool kinv = k && p.kinv != 0;
    auto one = [&](size_t i, float xv, bool tb, bool kb, float w1v, bool w2b, float old) {
        const float tt = tb ? 1.f : 0.f;
        const float e = __expf(-fabsf(xv));
        const float r = __frcp_rn(1.f + e);
        const float sg = xv >= 0.f ? r : e * r;
        const float gb = gS + gF1 * w1v + gF2 * (w2b ? 0.f : 1.f);
        const float v = kb ? (gb * (sg - tt) + sg * (1.f - sg) * (gA + gB * tt)) : 0.f;
        return p.accumulate ? old + v : v;
    };
    if ((p.V & 3) == 0 && (p.xstride & 3) == 0) {                        // 16-byte aligned planes: four voxels per thread
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < (size_t)p.V; i += (size_t)gridDim.x * 1024) {
            const float4 xq = *(const float4*)(x + i);
            uchar4 tq = {0, 0, 0, 0}, kq = {1, 1, 1, 1}, w2q = {0, 0, 0, 0};
            float4 w1q = {0.f, 0.f, 0.f, 0.f}, oq = {0.f, 0.f, 0.f, 0.f};
            if (t) tq = *(const uchar4*)(t + i);
            if (k) kq = *(const uchar4*)(k + i);
            if (w2) w2q = *(const uchar4*)(w2 + i);
            if (w1) w1q = *(const float4*)(w1 + i);
            if (p.accumulate) oq = *(const float4*)(dx + i);
            float4 o;
            o.x = one(i, xq.x, (tq.x & tm8) != 0, k ? ((kq.x != 0) != kinv) : true, w1q.x, w2q.x, oq.x);
            o.y = one(i, xq.y, (tq.y & tm8) != 0, k ? ((kq.y != 0) != kinv) : true, w1q.y, w2q.y, oq.y);
            o.z = one(i, xq.z, (tq.z & tm8) != 0, k ? ((kq.z != 0) != kinv) : true, w1q.z, w2q.z, oq.z);
            o.w = one(i, xq.w, (tq.w & tm8) != 0, k ? ((kq.w != 0) != kinv) : true, w1q.w, w2q.w, oq.w);
            *(float4*)(dx + i) = o;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)p.V; i += (size_t)gridDim.x * 256)
        dx[i] = one(i, x[i], t && (t[i] & tm8), k ? ((k[i] != 0) != kinv) : true, w1 ? w1[i] : 0.f, w2 && w2[i], p.accumulate ? dx[i] : 0.f);
}

// One block.  Thread c owns class c: column sums over the batch give the adaptive Tversky alpha (clamped to [0.2, 0.8];
// the clamp passes gradient inside the interval, like torch.clamp), then loss and Jacobian per (b, c).  f64 arithmetic on
// the f32 sums: (B, C) is tiny and the ~60 ATen launches this replaces were pure launch latency.
__global__ __launch_bounds__(256) void seg_from_sums_kernel(SegSumsParams p) {
    const int B = p.B, C = p.C;
    const double inv_n = 1.0 / ((double)B * C);
    double acc = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        double sFP = 0.0, sFN = 0.0;
        for (int b = 0; b < B; ++b) {
            const float* s = p.sums + ((size_t)b * C + c) * 6;
            sFP += (double)s[1] - (double)s[2];
            sFN += (double)s[3] - (double)s[2];
        }
        const double den_a = sFP + sFN + 1e-5;
        const double a_raw = sFP / den_a;
        const double alpha = fmin(fmax(a_raw, 0.2), 0.8);
        const bool pass = a_raw >= 0.2 && a_raw <= 0.8;
        double g_alpha = 0.0;
        for (int b = 0; b < B; ++b) {
            const float* s = p.sums + ((size_t)b * C + c) * 6;
            const double w = p.cw ? (double)p.cw[(size_t)b * C + c] : 1.0;
            const double TP = s[2], FP = (double)s[1] - TP, FN = (double)s[3] - TP;
            const double den = TP + alpha * FP + (1.0 - alpha) * FN + 1e-5;
            acc += (double)s[0] * w * p.inv_bcv + (1.0 - TP / den) * w * inv_n;
            g_alpha += w * inv_n * TP * (FP - FN) / (den * den);           // d(1 - dice)/d alpha
        }
        const double ga_fp = pass ? g_alpha * (sFN + 1e-5) / (den_a * den_a) : 0.0;
        const double ga_fn = pass ? -g_alpha * sFP / (den_a * den_a) : 0.0;
        for (int b = 0; b < B; ++b) {
            const size_t o = ((size_t)b * C + c) * 6;
            const float* s = p.sums + o;
            const double w = p.cw ? (double)p.cw[(size_t)b * C + c] : 1.0;
            const double TP = s[2], FP = (double)s[1] - TP, FN = (double)s[3] - TP;
            const double den = TP + alpha * FP + (1.0 - alpha) * FN + 1e-5;
            const double k = w * inv_n / (den * den);
            const double dTP = -k * (den - TP), dFP = k * TP * alpha + ga_fp, dFN = k * TP * (1.0 - alpha) + ga_fn;
            p.dsums[o + 0] = (float)(p.scale * w * p.inv_bcv);
            p.dsums[o + 1] = (float)(p.scale * dFP);                         // A  = FP + TP
            p.dsums[o + 2] = (float)(p.scale * (dTP - dFP - dFN));           // Bs = TP
            p.dsums[o + 3] = (float)(p.scale * dFN);                         // Cn = FN + TP
            p.dsums[o + 4] = 0.f;
            p.dsums[o + 5] = 0.f;
        }
    }
    __shared__ double red[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.loss[0] = (float)(p.scale * (red[0] + red[1] + red[2] + red[3]));
}

// One thread: the quantities are a few dozen scalars (f64 arithmetic on the f32 sums); what matters is that nothing leaves the device.
//   volume:  lv[b, li] = clamp(|x - y| / (x + y + E) - |v - y| / (v + y + E), 0, 1) * w,  x = sums[li B + b][1] * (1 - annotated), y = rvol[b] * gate,
//            v = max((1 - tol) y, min(y, 100));  dice_volume_loss = mean over (b, li)
//   plan without tumour: bce = sum_li S w / (L V); dice = mean_li (1 - TP / (TP + a FP + (1 - a) FN + 1e-5)) w,  a = clamp(FP / (FP + FN + 1e-5), .2, .8)
//   plan with tumour:    bce = (F1 + F2) / V * w  (S / V with standard_ce); dice = (1 - TP / (...)) w on its single row
//   ball_loss_bce / _dice = mean over the plans.  clamp passes the gradient on [min, max] like torch.clamp, |.| has derivative sign(.) (0 at 0).
__global__ void report_from_sums_kernel(ReportSumsParams p, int R) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float* J0 = p.jac; float* J1 = p.jac + (size_t)R * 6; float* J2 = p.jac + (size_t)2 * R * 6;
    for (int i = 0; i < 3 * R * 6; ++i) p.jac[i] = 0.f;
    const int B = p.B, L = p.L;
    double lvol = 0.0;
    int row = 0;
    if (p.use_vol) {
        const double inv = 1.0 / ((double)B * L);
        for (int li = 0; li < L; ++li)
            for (int b = 0; b < B; ++b) {
                const int r = li * B + b;
                const double keep = 1.0 - (double)p.flags[b * 2 * L + li], gate = (double)p.flags[b * 2 * L + L + li];
                const double x = (double)p.sums[r * 6 + 1] * keep, y = (double)p.rvol[b] * gate, w = (double)p.roww[r];
                const double s = x + y + p.E, d = x - y;
                const double v = fmax((1.0 - p.tol) * y, fmin(y, 100.0));
                const double raw = fabs(d) / s - fabs(v - y) / (v + y + p.E);
                const double lv = fmin(fmax(raw, 0.0), 1.0);
                lvol += lv * w * inv;
                const double sg = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
                const double draw = (raw >= 0.0 && raw <= 1.0) ? (sg / s - fabs(d) / (s * s)) : 0.0;
                J2[r * 6 + 1] = (float)(draw * keep * w * inv);
            }
        row = L * B;
    }
    double lbce = 0.0, ldice = 0.0;
    const double invp = p.nplans > 0 ? 1.0 / p.nplans : 0.0;
    auto dice_row = [&](int r, double w, double scale) {         // (1 - dice) * w of one row; Jacobian scaled by `scale` into J1; returns the loss
        const double A = p.sums[r * 6 + 1], Bs = p.sums[r * 6 + 2], Cn = p.sums[r * 6 + 3];
        const double TP = Bs, FP = A - Bs, FN = Cn - Bs;
        const double den_a = FP + FN + 1e-5, a_raw = FP / den_a;
        const double al = fmin(fmax(a_raw, 0.2), 0.8);
        const bool pass = a_raw >= 0.2 && a_raw <= 0.8;
        const double den = TP + al * FP + (1.0 - al) * FN + 1e-5;
        const double g_al = w * TP * (FP - FN) / (den * den);    // d((1 - dice) w) / d alpha
        const double ga_fp = pass ? g_al * (FN + 1e-5) / (den_a * den_a) : 0.0, ga_fn = pass ? -g_al * FP / (den_a * den_a) : 0.0;
        const double k = w / (den * den);
        const double dTP = -k * (den - TP), dFP = k * TP * al + ga_fp, dFN = k * TP * (1.0 - al) + ga_fn;
        J1[r * 6 + 1] += (float)(scale * dFP);
        J1[r * 6 + 2] += (float)(scale * (dTP - dFP - dFN));
        J1[r * 6 + 3] += (float)(scale * dFN);
        return (1.0 - TP / den) * w;
    };
    for (int q = 0; q < p.nplans; ++q) {
        const int kind = p.plan[2 * q], r0 = p.plan[2 * q + 1];
        if (kind == 0) {
            double sb = 0.0, sd = 0.0;
            for (int li = 0; li < L; ++li) {
                const int r = r0 + li;
                const double w = (double)p.roww[r];
                sb += (double)p.sums[r * 6 + 0] * w;
                J0[r * 6 + 0] = (float)(w / ((double)L * p.V) * invp);
                if (p.apply_dice) sd += dice_row(r, w, invp / L);
            }
            lbce += sb / ((double)L * p.V) * invp;
            ldice += sd / L * invp;
        } else {
            const int r = r0;
            const double w = (double)p.roww[r];
            if (p.standard_ce) {
                lbce += (double)p.sums[r * 6 + 0] / p.V * w * invp;
                J0[r * 6 + 0] = (float)(w / p.V * invp);
            } else {
                lbce += ((double)p.sums[r * 6 + 4] + (double)p.sums[r * 6 + 5]) / p.V * w * invp;
                J0[r * 6 + 4] = (float)(w / p.V * invp);
                J0[r * 6 + 5] = (float)(w / p.V * invp);
            }
            if (p.apply_dice) ldice += dice_row(r, w, invp) * invp;
        }
    }
    (void)row;
    p.loss[0] = (float)lbce; p.loss[1] = (float)ldice; p.loss[2] = (float)lvol;
}

// out[i] = sigmoid(x[i]) * (m ? m[i] : 1)      (x_iter of ball_loss, :1691-1694)
__global__ void sigmoid_mask_kernel(const float* x, const uint8_t* m, float* out, size_t V) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (1.f / (1.f + expf(-x[i]))) * ((!m || m[i]) ? 1.f : 0.f);   // accurate exp: feeds argmax / top-k selection
}


// Sliding-window inference (inference/inference3d.py:28-107).  The reference moves every window's sigmoid
// probabilities to the host and accumulates there; here the accumulator stays in HBM.
//   acc[b][k][d0+d][h0+h][w0+w] (+)= sigmoid(logits[b][k][d][h][w])      thread = one w-row segment of 4 voxels
__global__ __launch_bounds__(256) void window_accumulate_kernel(const float* __restrict__ logits, float* __restrict__ acc, int BK,
                                                                int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0, int assign) {
    const int wq = (ww + 3) / 4;
    const long total = (long)BK * wd * wh * wq;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        long r = t;
        const int q = (int)(r % wq); r /= wq;
        const int h = (int)(r % wh); r /= wh;
        const int d = (int)(r % wd); r /= wd;
        const int bk = (int)r;
        const float* src = logits + (((size_t)bk * wd + d) * wh + h) * ww + q * 4;
        float* dst = acc + (((size_t)bk * D + d0 + d) * H + h0 + h) * W + w0 + q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (q * 4 + j < ww) {
                const float p = 1.f / (1.f + expf(-src[j]));
                dst[j] = assign ? p : dst[j] + p;
            }
        }
    }
}

// out = acc / (cd[d] * ch[h] * cw[w]): the per-voxel window count of a half-overlapping grid is separable
__global__ __launch_bounds__(256) void window_normalize_kernel(float* __restrict__ acc, const float* __restrict__ cd, const float* __restrict__ ch,
                                                               const float* __restrict__ cw, long BK, int D, int H, int W) {
    const long total = BK * D * H * W;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int w = (int)(t % W);
        const int h = (int)((t / W) % H);
        const int d = (int)((t / ((long)W * H)) % D);
        acc[t] = acc[t] / (cd[d] * ch[h] * cw[w]);
    }
}
}  // namespace

int rs_plane_partials_blocks(size_t V) {
    int blocks = (int)((V + 4095) / 4096);
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    return blocks;
}
int rs_launch_plane_sums_reduce(const double* pblk, int rows, int nb, float* out, hipStream_t st) {
    hipLaunchKernelGGL(plane_sums_reduce_kernel, dim3((unsigned)((rows * 6 + 255) / 256)), dim3(256), 0, st, pblk, rows, nb, out);
    return rs_check_launch();
}
int rs_launch_plane_partials(const PlaneParams& p, int planes, int bwd, hipStream_t st) {
    const int blocks = rs_plane_partials_blocks(p.V);
    if (!bwd) {
        hipLaunchKernelGGL(plane_partials_fwd_kernel, dim3(blocks, planes), dim3(256), 0, st, p);
    } else {
        int b2 = (int)((p.V + 1023) / 1024);
        if (b2 > 256) b2 = 256;
        hipLaunchKernelGGL(plane_partials_bwd_kernel, dim3(b2, planes), dim3(256), 0, st, p);
    }
    return rs_check_launch();
}

int rs_launch_seg_from_sums(const SegSumsParams& p, hipStream_t st) {
    hipLaunchKernelGGL(seg_from_sums_kernel, dim3(1), dim3(256), 0, st, p);
    return rs_check_launch();
}

int rs_launch_report_from_sums(const ReportSumsParams& p, int R, hipStream_t st) {
    hipLaunchKernelGGL(report_from_sums_kernel, dim3(1), dim3(64), 0, st, p, R);
    return rs_check_launch();
}

int rs_launch_sigmoid_mask(const float* x, const uint8_t* m, float* out, size_t V, hipStream_t st) {
    hipLaunchKernelGGL(sigmoid_mask_kernel, dim3(rs_elem_blocks(V)), dim3(256), 0, st, x, m, out, V);
    return rs_check_launch();
}

int rs_launch_window_accumulate(const float* logits, float* acc, int BK, int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0,
                                int assign, hipStream_t st) {
    const long total = (long)BK * wd * wh * ((ww + 3) / 4);
    hipLaunchKernelGGL(window_accumulate_kernel, dim3(rs_elem_blocks((size_t)total)), dim3(256), 0, st, logits, acc, BK, wd, wh, ww, D, H, W, d0, h0, w0, assign);
    return rs_check_launch();
}
int rs_launch_window_normalize(float* acc, const float* cd, const float* ch, const float* cw, long BK, int D, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(window_normalize_kernel, dim3(rs_elem_blocks((size_t)(BK * D * H * W))), dim3(256), 0, st, acc, cd, ch, cw, BK, D, H, W);
    return rs_check_launch();
}
