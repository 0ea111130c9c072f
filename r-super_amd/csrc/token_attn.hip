// Self-attention core of a short token sequence: softmax(q k^T * scale) v per (sample, head), forward and backward, f32.
//
// Replaces the ATen chain of Attention.forward in MedFormer's SemanticMapFusion transformer
// (rsuper_train/model/dim3/trans_layers.py:52-84 under medformer_utils.py:239-273: chunk -> rearrange -> matmul -> scale -> softmax ->
// matmul -> rearrange, eight launches forward and about twelve backward on 81 tokens) by one launch per direction.
// qkv (B, L, 3 * H * Dh): q | k | v thirds, head h at columns [h * Dh, (h + 1) * Dh) of its third -- the layout
// `to_qkv(x).chunk(3, -1)` + `b l (h d) -> b h l d` reads; o (B, L, H * Dh) in the layout `b h l d -> b l (h d)` writes, so neither side
// needs a transposing copy.  Per (head, sample) a few blocks split the score matrix by row chunks (forward, dQ) and column chunks (dK, dV):
// the operands live in LDS, a wave owns one row / column (lane = the other index), sums run in a fixed order (deterministic).  The sequence is tiny (81 tokens x 10 heads x 2 samples:
// 0.8 MFLOP per block) -- this is launch-count work, FMA on the vector ALU, not an MFMA kernel.
#include "common.hpp"
#include "kernels.hpp"

namespace {

struct TaParams {
    const float* qkv; float* o; float* p;        // p: (B, H, L, L) soft-max rows (written forward, read backward)
    const float* d_o; float* d_qkv;
    int B, L, H, Dh; float scale;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int TA_MAXL = 128;                                      // two score columns per lane
constexpr int TA_NT = 512, TA_NW = TA_NT / 64;                    // 8 waves (register budget 256: the column rows below): a (head, sample) pair is one block, its rows are the only parallelism

// sum over the lanes that hold the same d (lane = part * DP2 + d): parts are DP2 apart
template <int DP2> __device__ __forceinline__ float part_sum(float v) {
#pragma unroll
    for (int o = 32; o >= DP2; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// rows (or columns) of the score matrix a block owns: chunk z of nz
__device__ __forceinline__ void chunk_range(int L, int z, int nz, int& lo, int& hi) {
    const int per = (L + nz - 1) / nz;
    lo = z * per; hi = lo + per < L ? lo + per : L;
}

// ---- forward: grid (H, B, nz); a block owns a chunk of score rows (a wave per row, lane = column)
template <int DP2>
__global__ __launch_bounds__(TA_NT) void token_attn_fwd_kernel(TaParams a, int nz) {
    extern __shared__ float sm[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = a.L, Dh = a.Dh, DP = Dh + 1, LP = L + 1, HD = a.H * Dh;
    float* Q = sm; float* K = Q + L * DP; float* V = K + L * DP; float* P = V + L * DP;      // P: one row per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lo, hi;
    chunk_range(L, blockIdx.z, nz, lo, hi);
    const float* src = a.qkv + (size_t)b * L * 3 * HD + h * Dh;
    for (int e = tid; e < L * Dh; e += TA_NT) {
        const int i = e / Dh, d = e - i * Dh;
        const float* r = src + (size_t)i * 3 * HD + d;
        if (i >= lo && i < hi) Q[i * DP + d] = r[0];
        K[i * DP + d] = r[HD]; V[i * DP + d] = r[2 * HD];
    }
    __syncthreads();
    constexpr int NP = 64 / DP2;                                 // lanes per d: the j range is dealt round-robin to NP parts
    const int d = lane % DP2, part = lane / DP2;
    const bool d_ok = d < Dh;
    // the lane's two score columns never change: their K rows stay in registers, a score costs one broadcast LDS read per multiply-add
    constexpr bool REG = DP2 <= 32;
    float cr[2][REG ? DP2 : 1];
    if (REG) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < DP2; ++k) cr[c][k] = (lane + 64 * c < L && k < Dh) ? K[(lane + 64 * c) * DP + k] : 0.f;
    }
    float* pg = a.p + ((size_t)b * a.H + h) * L * L;
    float* prow = P + wave * LP;
    for (int i = lo + wave; i < hi; i += TA_NW) {
        float s[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            float t = 0.f;
            if (REG) {
#pragma unroll
                for (int k = 0; k < DP2; ++k) t = fmaf(k < Dh ? Q[i * DP + k] : 0.f, cr[c][REG ? k : 0], t);
            } else if (j < L) {
                for (int k = 0; k < Dh; ++k) t = fmaf(Q[i * DP + k], K[j * DP + k], t);
            }
            s[c] = j < L ? t * a.scale : -INFINITY;
        }
        const float m = wave_max(fmaxf(s[0], s[1]));
        const float e0 = lane < L ? __expf(s[0] - m) : 0.f, e1 = lane + 64 < L ? __expf(s[1] - m) : 0.f;
        const float inv = 1.f / wave_sum(e0 + e1);
        if (lane < L) { prow[lane] = e0 * inv; pg[(size_t)i * L + lane] = e0 * inv; }
        if (lane + 64 < L) { prow[lane + 64] = e1 * inv; pg[(size_t)i * L + lane + 64] = e1 * inv; }
        __builtin_amdgcn_wave_barrier();                         // the row is written and read by this wave only (LDS is in order per wave)
        float acc = 0.f;
        if (d_ok) {
#pragma unroll 8
            for (int j = part; j < L; j += NP) acc = fmaf(prow[j], V[j * DP + d], acc);   // unrolled: eight LDS reads in flight (two waves per SIMD hide nothing)
        }
        acc = part_sum<DP2>(acc);
        if (part == 0 && d_ok) a.o[((size_t)b * L + i) * HD + h * Dh + d] = acc;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- backward: grid (H, B, 2 nz).  z < nz: ROW role -- dS rows of the chunk (delta = row sum of dP o P) -> dQ.  z >= nz: COLUMN role -- dS
// columns of the chunk for all rows (delta_i = dO_i . O_i, the same number up to rounding) -> dK, dV.  The two roles share nothing, so the
// 2 nz blocks of a (head, sample) pair run side by side; dP = dO V^T is evaluated by both (0.2 MFLOP).
template <int DP2>
__global__ __launch_bounds__(TA_NT) void token_attn_bwd_kernel(TaParams a, int nz) {
    extern __shared__ float sm[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = a.L, Dh = a.Dh, DP = Dh + 1, LP = L + 1, HD = a.H * Dh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool rows = (int)blockIdx.z < nz;
    int lo, hi;
    chunk_range(L, rows ? blockIdx.z : blockIdx.z - nz, nz, lo, hi);
    // LDS: A = K (rows) / Q (columns), all L rows; B = V (rows: all rows; columns: only the chunk is used); dO all rows; then per-wave vectors
    float* A = sm; float* Bm = A + L * DP; float* dO = Bm + L * DP;
    float* delta = dO + L * DP;                                  // [L]  (column role)
    float* wv = delta + LP;                                      // [TA_NW][2][LP]: a wave's dS row / column and P column
    const float* src = a.qkv + (size_t)b * L * 3 * HD + h * Dh;
    for (int e = tid; e < L * Dh; e += TA_NT) {
        const int i = e / Dh, d = e - i * Dh;
        const float* r = src + (size_t)i * 3 * HD + d;
        A[i * DP + d] = rows ? r[HD] : r[0];
        Bm[i * DP + d] = r[2 * HD];
        dO[i * DP + d] = a.d_o[((size_t)b * L + i) * HD + h * Dh + d];
    }
    __syncthreads();
    if (!rows) {
        for (int i = tid; i < L; i += TA_NT) {
            const float* orow = a.o + ((size_t)b * L + i) * HD + h * Dh;
            float t = 0.f;
            for (int k = 0; k < Dh; ++k) t = fmaf(dO[i * DP + k], orow[k], t);
            delta[i] = t;
        }
        __syncthreads();
    }
    constexpr int NP = 64 / DP2;
    const int d = lane % DP2, part = lane / DP2;
    const bool d_ok = d < Dh;
    const float* pg = a.p + ((size_t)b * a.H + h) * L * L;
    float* dst = a.d_qkv + (size_t)b * L * 3 * HD + h * Dh;
    float* ds_ = wv + wave * 2 * LP; float* pc = ds_ + LP;
    constexpr bool REG = DP2 <= 32;
    float cr[2][REG ? DP2 : 1];                                  // rows role: V rows of the lane's two columns; columns role: dO rows of its two rows
    if (REG) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < DP2; ++k) cr[c][k] = (lane + 64 * c < L && k < Dh) ? (rows ? Bm : dO)[(lane + 64 * c) * DP + k] : 0.f;
    }
    auto dots = [&](const float* bc, const float* M, float* out) {   // out[c] = sum_k bc[k] * M[lane + 64 c][k]   (bc: a broadcast row)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            float t = 0.f;
            if (REG) {
#pragma unroll
                for (int k = 0; k < DP2; ++k) t = fmaf(k < Dh ? bc[k] : 0.f, cr[c][REG ? k : 0], t);
            } else if (j < L) {
                for (int k = 0; k < Dh; ++k) t = fmaf(bc[k], M[j * DP + k], t);
            }
            out[c] = t;
        }
    };
    if (rows) {
        for (int i = lo + wave; i < hi; i += TA_NW) {
            float dp[2], pr[2];
            dots(dO + i * DP, Bm, dp);
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) { pr[c] = lane + 64 * c < L ? pg[(size_t)i * L + lane + 64 * c] : 0.f; dot = fmaf(dp[c], pr[c], dot); }
            dot = wave_sum(dot);
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (lane + 64 * c < L) ds_[lane + 64 * c] = pr[c] * (dp[c] - dot);
            __builtin_amdgcn_wave_barrier();
            float acc = 0.f;
            if (d_ok) {
#pragma unroll 8
                for (int j = part; j < L; j += NP) acc = fmaf(ds_[j], A[j * DP + d], acc);
            }
            acc = part_sum<DP2>(acc);
            if (part == 0 && d_ok) dst[(size_t)i * 3 * HD + d] = acc * a.scale;
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (int j = lo + wave; j < hi; j += TA_NW) {
            float dp[2];
            dots(Bm + j * DP, dO, dp);                           // lane = row i: dP[i][j] = dO[i] . V[j]
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = lane + 64 * c;
                if (i < L) { const float pv = pg[(size_t)i * L + j]; pc[i] = pv; ds_[i] = pv * (dp[c] - delta[i]); }
            }
            __builtin_amdgcn_wave_barrier();
            float ak = 0.f, av = 0.f;
            if (d_ok) {
#pragma unroll 8
                for (int i = part; i < L; i += NP) {
                    ak = fmaf(ds_[i], A[i * DP + d], ak);
                    av = fmaf(pc[i], dO[i * DP + d], av);
                }
            }
            ak = part_sum<DP2>(ak); av = part_sum<DP2>(av);
            if (part == 0 && d_ok) { dst[(size_t)j * 3 * HD + HD + d] = ak * a.scale; dst[(size_t)j * 3 * HD + 2 * HD + d] = av; }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

size_t ta_smem(int L, int Dh, bool bwd) {
    const size_t rows = (size_t)L * (Dh + 1), lp = L + 1;
    return 4 * (bwd ? 3 * rows + lp + TA_NW * 2 * lp : 3 * rows + TA_NW * lp);
}

int ta_chunks(int L) { return L >= 64 ? 4 : L >= 16 ? 2 : 1; }

template <int DP2>
int launch_ta(const TaParams& a, bool bwd, hipStream_t st) {
    const size_t smem = ta_smem(a.L, a.Dh, bwd);
    const int nz = ta_chunks(a.L);
    if (bwd) {
        auto k = token_attn_bwd_kernel<DP2>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, dim3(a.H, a.B, 2 * nz), dim3(TA_NT), smem, st, a, nz);
    } else {
        auto k = token_attn_fwd_kernel<DP2>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, dim3(a.H, a.B, nz), dim3(TA_NT), smem, st, a, nz);
    }
    return rs_check_launch();
}

}  // namespace

int rs_token_attn_supported(int L, int Dh) {
    return L >= 1 && L <= TA_MAXL && Dh >= 1 && Dh <= 64 && ta_smem(L, Dh, true) <= 160 * 1024;
}

// d_qkv == nullptr: forward (writes o, p); else backward (reads o, p, d_o)
int rs_launch_token_attn(const float* qkv, float* o, float* p, const float* d_o, float* d_qkv, int B, int L, int H, int Dh, float scale, hipStream_t st) {
    if (!rs_token_attn_supported(L, Dh)) return RS_ERR_UNSUPPORTED;
    const TaParams a = {qkv, o, p, d_o, d_qkv, B, L, H, Dh, scale};
    const bool bwd = d_qkv != nullptr;
    return Dh <= 16 ? launch_ta<16>(a, bwd, st) : Dh <= 32 ? launch_ta<32>(a, bwd, st) : launch_ta<64>(a, bwd, st);
}
