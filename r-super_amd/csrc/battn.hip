// Bidirectional attention between the voxels of a stage and its semantic-map tokens, forward and backward, fp32 channels-last --
// the core of MedFormer's BidirectionAttention (rsuper_train/model/dim3/medformer_utils.py:13-99 of the reference): ONE score
// matrix S[t, l] = scale * <mq[t], fq[l]> per (sample, head), soft-maxed over the T map tokens for the voxel update and over
// the L voxels for the map update:
//
//     f_out[l, :] = sum_t softmax_t(S[:, l])[t] * mv[t, :]            m_out[t, :] = sum_l softmax_l(S[t, :])[l] * fv[l, :]
//
// Layout: fqv (B, L, 2*inner) and mqv (B, T, 2*inner) are the outputs of the q/v projections as they leave the GEMMs (q in the
// first `inner` channels, v in the second), channel = dim_head_index * heads + head (the reference's 'b (dim_head heads) ...'
// rearrange); f_out (B, L, inner) / m_out (B, T, inner) use the same channel order, so no head split / merge copies exist.
// T is tiny (27 = 3x3x3 tokens): a thread owns one (voxel, head) pair with its score column in registers; the token rows of the
// map sit in LDS.  The voxel soft-max is local to the thread; the soft-max over the voxels is the flash-attention split: each
// block keeps (max, sum, weighted sum) per token for its voxels, a second kernel merges the blocks in a fixed order (no atomics:
// bit-reproducible).  Backward recomputes S from q and the saved (max, sum) per token, uses D[t] = <dM[t], m_out[t]> for the
// voxel-axis soft-max, writes dq / dv of the voxels directly and reduces dmq / dmv through per-block partial rows.
// The ATen composition this replaces issued ~25 forward and ~40 backward launches per attention (head split / merge copies, two
// soft-max passes over a (B, heads, T, L) tensor and its transpose, three batched GEMMs of width 27).
#include "common.hpp"
#include "misc.hpp"

namespace {

struct BaParams {
    const float* fqv; const float* mqv;
    float* fout; float* mout; float* lse;                 // lse (B, heads, T, 2) = (max, sum) of the voxel-axis soft-max
    const float* dfo; const float* dmo; float* dfqv; float* dmqv;
    float* part; float* pms;
    int B, L, heads, inner, chunks; float scale;
};

template <int T, int DH>
struct BaCfg {
    static constexpr int TP = (T + 3) & ~3;               // row pitch of the per-thread score rows in LDS
    static constexpr int HS = T * DH + 4;                 // pitch of one head's token table: heads land on disjoint banks
};

// Token rows (T, ld) -> per-head tables [head][t][d] in LDS.  Four channels per thread and iteration as one 16-byte load; the loop is
// unrolled so the loads of several iterations are in flight together (one load per iteration was pure latency: 70 us per block).
template <int T, int DH>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int ld, int c0, int nc, int heads, float* __restrict__ dst, float mul) {
    constexpr int HS = BaCfg<T, DH>::HS;
    const int nq = nc >> 2, total = T * nq;
#pragma unroll 4
    for (int i = threadIdx.x; i < total; i += 256) {
        const int t = i / nq, c = (i - t * nq) * 4;
        const float4 x = *(const float4*)(src + (size_t)t * ld + c0 + c);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = c + k, d = cc / heads, h = cc - d * heads;
            dst[h * HS + t * DH + d] = xs[k] * mul;
        }
    }
}

template <int T, int DH>
__device__ __forceinline__ void stage_tokens(const BaParams& p, const float* __restrict__ src, float* q, float* v, float qscale) {
    stage_rows<T, DH>(src, 2 * p.inner, 0, p.inner, p.heads, q, qscale);
    stage_rows<T, DH>(src, 2 * p.inner, p.inner, p.inner, p.heads, v, 1.f);
}

template <int T, int DH>
__device__ __forceinline__ void dots(const float (&a)[DH], const float* __restrict__ rows, float (&out)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 m = *(const float4*)(rows + t * DH + d);
            s = fmaf(a[d], m.x, s); s = fmaf(a[d + 1], m.y, s); s = fmaf(a[d + 2], m.z, s); s = fmaf(a[d + 3], m.w, s);
        }
        out[t] = s;
    }
}

template <int T, int DH>
__device__ __forceinline__ void mix(const float (&w)[T], const float* __restrict__ rows, float (&out)[DH]) {
#pragma unroll
    for (int d = 0; d < DH; ++d) out[d] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 m = *(const float4*)(rows + t * DH + d);
            out[d] = fmaf(w[t], m.x, out[d]); out[d + 1] = fmaf(w[t], m.y, out[d + 1]);
            out[d + 2] = fmaf(w[t], m.z, out[d + 2]); out[d + 3] = fmaf(w[t], m.w, out[d + 3]);
        }
    }
}

// sum over the block's voxels of w[t][voxel, head(c)] * x[voxel][c] for the channels c = tid, tid + 256, ...; lanes walk channels, so
// the rows of x (just read by this block: L1 / L2 hits) stream coalesced and the weights come from LDS as 16-byte reads
template <int T>
__device__ __forceinline__ void reduce_voxels(const float* __restrict__ pbuf, const float* __restrict__ xrows, int ldx, int nv, int heads,
                                              int inner, float* __restrict__ dst, int ldd) {
    constexpr int TP = (T + 3) & ~3;
    for (int c = threadIdx.x; c < inner; c += 256) {
        const int h = c % heads;
        float acc[TP];
#pragma unroll
        for (int t = 0; t < TP; ++t) acc[t] = 0.f;
        const float* w = pbuf + h * TP;
#pragma unroll 4
        for (int vl = 0; vl < nv; ++vl) {
            const float x = xrows[(size_t)vl * ldx + c];
            const float* wr = w + vl * heads * TP;
#pragma unroll
            for (int t = 0; t < TP; t += 4) {
                const float4 m = *(const float4*)(wr + t);
                acc[t] = fmaf(m.x, x, acc[t]); acc[t + 1] = fmaf(m.y, x, acc[t + 1]);
                acc[t + 2] = fmaf(m.z, x, acc[t + 2]); acc[t + 3] = fmaf(m.w, x, acc[t + 3]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) dst[(size_t)t * ldd + c] = acc[t];
    }
}

template <int T, int DH>
__global__ __launch_bounds__(256) void battn_fwd_kernel(BaParams p) {
    extern __shared__ __attribute__((aligned(16))) float ba_sm[];
    constexpr int TP = BaCfg<T, DH>::TP, HS = BaCfg<T, DH>::HS;
    const int heads = p.heads, inner = p.inner, ld = 2 * inner;
    float* mq = ba_sm;
    float* mv = mq + heads * HS;
    float* pbuf = mv + heads * HS;                          // [256][TP]
    float* red = pbuf + 256 * TP;                           // [heads * T]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int VPB = 256 / heads, nthr = VPB * heads;
    const int tid = threadIdx.x, vl = tid / heads, h = tid - vl * heads;
    const int v0 = chunk * VPB, v = v0 + vl;
    const bool act = tid < nthr && v < p.L;
    const int vc = v < p.L ? v : p.L - 1;
    float q[DH];
    {
        const float* qr = p.fqv + ((size_t)b * p.L + vc) * ld + h;
#pragma unroll
        for (int d = 0; d < DH; ++d) q[d] = qr[d * heads];
    }
    stage_tokens<T, DH>(p, p.mqv + (size_t)b * T * ld, mq, mv, p.scale);
    __syncthreads();
    float S[T];
    dots<T, DH>(q, mq + h * HS, S);
    {   // voxel update: soft-max over the tokens, local to the thread
        float mx = S[0];
#pragma unroll
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, S[t]);
        float e[T], den = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) { e[t] = expf(S[t] - mx); den += e[t]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int t = 0; t < T; ++t) e[t] *= inv;
        float o[DH];
        mix<T, DH>(e, mv + h * HS, o);
        if (act) {
            float* orow = p.fout + ((size_t)b * p.L + v) * inner + h;
#pragma unroll
            for (int d = 0; d < DH; ++d) orow[d * heads] = o[d];
        }
    }
    // map update: block-local (max, sum, weighted sum) per (head, token) over this block's voxels
#pragma unroll
    for (int t = 0; t < T; ++t) pbuf[tid * TP + t] = act ? S[t] : -INFINITY;
    __syncthreads();
    for (int i = tid; i < heads * T; i += 256) {
        const int hh = i / T, t = i - hh * T;
        float m = -INFINITY;
        for (int k = 0; k < VPB; ++k) m = fmaxf(m, pbuf[(k * heads + hh) * TP + t]);
        red[i] = m;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TP; ++t) pbuf[tid * TP + t] = (t < T && act) ? expf(S[t < T ? t : 0] - red[h * T + (t < T ? t : 0)]) : 0.f;
    __syncthreads();
    for (int i = tid; i < heads * T; i += 256) {
        const int hh = i / T, t = i - hh * T;
        float s = 0.f;
        for (int k = 0; k < VPB; ++k) s += pbuf[(k * heads + hh) * TP + t];
        float* ms = p.pms + ((((size_t)b * p.chunks + chunk) * heads + hh) * T + t) * 2;
        ms[0] = red[i]; ms[1] = s;
    }
    const int nv = p.L - v0 < VPB ? p.L - v0 : VPB;
    reduce_voxels<T>(pbuf, p.fqv + ((size_t)b * p.L + v0) * ld + inner, ld, nv, heads, inner,
                     p.part + ((size_t)b * p.chunks + chunk) * T * inner, inner);
}

// merge of the blocks' soft-max partials: grid (T, B, inner / 64); thread = (channel, one of four interleaved chunk lanes), each lane an
// online (max, sum, weighted sum) over its chunks, the four lanes merged through LDS
__global__ __launch_bounds__(256) void battn_fwd_merge_kernel(BaParams p, int T) {
    __shared__ float red[3][4][64];
    const int t = blockIdx.x, b = blockIdx.y, cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int c = blockIdx.z * 64 + cl;
    const bool ok = c < p.inner;
    const int cc = ok ? c : 0, h = cc % p.heads;
    const float* ms = p.pms + (((size_t)b * p.chunks * p.heads + h) * T + t) * 2;
    const size_t mstep = (size_t)p.heads * T * 2, pstep = (size_t)T * p.inner;
    const float* pr = p.part + ((size_t)b * p.chunks * T + t) * p.inner + cc;
    float m = -INFINITY, s = 0.f, o = 0.f;
    for (int k = kg; k < p.chunks; k += 4) {
        const float mc = ms[k * mstep], sc = ms[k * mstep + 1], oc = pr[k * pstep];
        const float mn = fmaxf(m, mc);
        const float e1 = expf(m - mn), e2 = expf(mc - mn);
        s = fmaf(sc, e2, s * e1);
        o = fmaf(oc, e2, o * e1);
        m = mn;
    }
    red[0][kg][cl] = m; red[1][kg][cl] = s; red[2][kg][cl] = o;
    __syncthreads();
    if (kg == 0 && ok) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float mc = red[0][k][cl];
            if (mc == -INFINITY) continue;                   // a lane without chunks
            const float mn = fmaxf(m, mc);
            const float e1 = expf(m - mn), e2 = expf(mc - mn);
            s = fmaf(red[1][k][cl], e2, s * e1);
            o = fmaf(red[2][k][cl], e2, o * e1);
            m = mn;
        }
        p.mout[((size_t)b * T + t) * p.inner + c] = o / s;
        if (c < p.heads) {
            float* l = p.lse + (((size_t)b * p.heads + h) * T + t) * 2;
            l[0] = m; l[1] = s;
        }
    }
}

template <int T, int DH>
__global__ __launch_bounds__(256) void battn_bwd_kernel(BaParams p) {
    extern __shared__ __attribute__((aligned(16))) float ba_sm[];
    constexpr int TP = BaCfg<T, DH>::TP, HS = BaCfg<T, DH>::HS;
    const int heads = p.heads, inner = p.inner, ld = 2 * inner;
    float* mq = ba_sm;
    float* mv = mq + heads * HS;
    float* dm = mv + heads * HS;
    float* pbuf = dm + heads * HS;                          // [256][TP]
    float* lm = pbuf + 256 * TP;                            // [heads * T] max, 1 / sum, D
    float* ls = lm + heads * T;
    float* dv = ls + heads * T;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int VPB = 256 / heads, nthr = VPB * heads;
    const int tid = threadIdx.x, vl = tid / heads, h = tid - vl * heads;
    const int v0 = chunk * VPB, v = v0 + vl;
    const bool act = tid < nthr && v < p.L;
    const int vc = v < p.L ? v : p.L - 1;
    const float* frow = p.fqv + ((size_t)b * p.L + vc) * ld + h;
    const float* grow = p.dfo + ((size_t)b * p.L + vc) * inner + h;
    float a[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) a[d] = frow[d * heads];
    stage_tokens<T, DH>(p, p.mqv + (size_t)b * T * ld, mq, mv, p.scale);
    stage_rows<T, DH>(p.dmo + (size_t)b * T * inner, inner, 0, inner, heads, dm, 1.f);   // dM (B, T, inner) -> [head][t][d]
    __syncthreads();
    for (int i = tid; i < heads * T; i += 256) {
        const int hh = i / T, t = i - hh * T;
        const float* l = p.lse + (((size_t)b * heads + hh) * T + t) * 2;
        lm[i] = l[0]; ls[i] = 1.f / l[1];
        const float* mo = p.mout + ((size_t)b * T + t) * inner + hh;
        float s = 0.f;
        for (int d = 0; d < DH; ++d) s = fmaf(dm[hh * HS + t * DH + d], mo[d * heads], s);
        dv[i] = s;
    }
    __syncthreads();
    float Af[T], Am[T], dS[T];
    {
        float S[T];
        dots<T, DH>(a, mq + h * HS, S);
        float mx = S[0];
#pragma unroll
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, S[t]);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) { Af[t] = expf(S[t] - mx); den += Af[t]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int t = 0; t < T; ++t) { Af[t] *= inv; Am[t] = expf(S[t] - lm[h * T + t]) * ls[h * T + t]; }
    }
    asm volatile("" ::: "memory");                           // one operand row live at a time: keeps the later rows' loads below this point
#pragma unroll
    for (int d = 0; d < DH; ++d) a[d] = grow[d * heads];     // dF of this (voxel, head)
    {
        float g[T];
        dots<T, DH>(a, mv + h * HS, g);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) rs = fmaf(Af[t], g[t], rs);
#pragma unroll
        for (int t = 0; t < T; ++t) dS[t] = Af[t] * (g[t] - rs);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < DH; ++d) a[d] = frow[inner + d * heads];   // fv
    {
        float g[T];
        dots<T, DH>(a, dm + h * HS, g);
#pragma unroll
        for (int t = 0; t < T; ++t) dS[t] = fmaf(Am[t], g[t] - dv[h * T + t], dS[t]);
    }
    asm volatile("" ::: "memory");
    float* drow = p.dfqv + ((size_t)b * p.L + vc) * ld + h;
    mix<T, DH>(dS, mq + h * HS, a);                          // dq = sum_t dS[t] * scale * mq[t]
    if (act) {
#pragma unroll
        for (int d = 0; d < DH; ++d) drow[d * heads] = a[d];
    }
    asm volatile("" ::: "memory");
    mix<T, DH>(Am, dm + h * HS, a);                          // dfv = sum_t Am[t] * dM[t]
    if (act) {
#pragma unroll
        for (int d = 0; d < DH; ++d) drow[inner + d * heads] = a[d];
    }
    const int nv = p.L - v0 < VPB ? p.L - v0 : VPB;
    float* prow = p.part + ((size_t)b * p.chunks + chunk) * T * ld;
#pragma unroll
    for (int t = 0; t < TP; ++t) pbuf[tid * TP + t] = (t < T && act) ? dS[t < T ? t : 0] : 0.f;
    __syncthreads();
    reduce_voxels<T>(pbuf, p.fqv + ((size_t)b * p.L + v0) * ld, ld, nv, heads, inner, prow, ld);              // dmq / scale
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TP; ++t) pbuf[tid * TP + t] = (t < T && act) ? Af[t < T ? t : 0] : 0.f;
    __syncthreads();
    reduce_voxels<T>(pbuf, p.dfo + ((size_t)b * p.L + v0) * inner, inner, nv, heads, inner, prow + inner, ld); // dmv
}

__global__ __launch_bounds__(256) void battn_bwd_merge_kernel(BaParams p, int T) {
    __shared__ float red[4][64];
    const int t = blockIdx.x, b = blockIdx.y, ld = 2 * p.inner, cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int c = blockIdx.z * 64 + cl;
    const bool ok = c < ld;
    const float* pr = p.part + ((size_t)b * p.chunks * T + t) * ld + (ok ? c : 0);
    const size_t step = (size_t)T * ld;
    float s0 = 0.f, s1 = 0.f;
    int k = kg;
    for (; k + 4 < p.chunks; k += 8) { s0 += pr[k * step]; s1 += pr[(k + 4) * step]; }
    if (k < p.chunks) s0 += pr[k * step];
    red[kg][cl] = s0 + s1;
    __syncthreads();
    if (kg == 0 && ok) {
        const float s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        p.dmqv[((size_t)b * T + t) * ld + c] = c < p.inner ? s * p.scale : s;
    }
}

template <int T, int DH>
size_t ba_smem(int heads, int bwd) {
    return ((size_t)(bwd ? 3 : 2) * heads * BaCfg<T, DH>::HS + 256 * BaCfg<T, DH>::TP + (size_t)(bwd ? 3 : 1) * heads * T) * sizeof(float);
}

template <int T, int DH>
int launch(const BaParams& p, int bwd, hipStream_t st) {
    const size_t smem = ba_smem<T, DH>(p.heads, bwd);
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    dim3 grid(p.chunks, p.B);
    if (!bwd) {
        auto k = battn_fwd_kernel<T, DH>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
        hipLaunchKernelGGL(battn_fwd_merge_kernel, dim3(T, p.B, (p.inner + 63) / 64), dim3(256), 0, st, p, T);
    } else {
        auto k = battn_bwd_kernel<T, DH>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
        hipLaunchKernelGGL(battn_bwd_merge_kernel, dim3(T, p.B, (2 * p.inner + 63) / 64), dim3(256), 0, st, p, T);
    }
    return rs_check_launch();
}

}  // namespace

int rs_battn_supported(int T, int dh, int heads) {
    return ((T == 27 && dh == 32) || (T == 8 && dh == 16)) && heads >= 1 && heads <= 10;
}

int rs_battn_chunks(int L, int heads) {
    if (heads < 1 || heads > 256) return 0;
    const int vpb = 256 / heads;
    return (L + vpb - 1) / vpb;
}

int rs_launch_battn(const float* fqv, const float* mqv, float* fout, float* mout, float* lse, const float* dfo, const float* dmo,
                    float* dfqv, float* dmqv, float* part, float* pms, int B, int L, int T, int heads, int dh, float scale, int bwd,
                    hipStream_t st) {
    if (!rs_battn_supported(T, dh, heads)) return RS_ERR_UNSUPPORTED;
    if ((size_t)B * L * heads * dh * 2 >= 0x7FFFFFFFull) return RS_ERR_UNSUPPORTED;
    BaParams p = {fqv, mqv, fout, mout, lse, dfo, dmo, dfqv, dmqv, part, pms, B, L, heads, heads * dh, rs_battn_chunks(L, heads), scale};
    if (T == 27) return launch<27, 32>(p, bwd, st);
    return launch<8, 16>(p, bwd, st);
}
