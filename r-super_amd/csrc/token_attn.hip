// Self-attention core of a short token sequence: softmax(q k^T * scale) v per (sample, head), forward and backward, f32.
//
// Replaces the ATen chain of Attention.forward in MedFormer's SemanticMapFusion transformer
// (rsuper_train/model/dim3/trans_layers.py:52-84 under medformer_utils.py:239-273: chunk -> rearrange -> matmul -> scale -> softmax ->
// matmul -> rearrange, eight launches forward and about twelve backward on 81 tokens) by one launch per direction.
// qkv (B, L, 3 * H * Dh): q | k | v thirds, head h at columns [h * Dh, (h + 1) * Dh) of its third -- the layout
// `to_qkv(x).chunk(3, -1)` + `b l (h d) -> b h l d` reads; o (B, L, H * Dh) in the layout `b h l d -> b l (h d)` writes, so neither side
// needs a transposing copy.  One block per (head, sample): Q, K, V (and dO, P, dS in the backward) live in LDS, a wave owns a row of the
// score matrix (lane = column), sums run in a fixed order (deterministic).  The sequence is tiny (81 tokens x 10 heads x 2 samples:
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

template <int DP2, bool BWD>
__global__ __launch_bounds__(TA_NT) void token_attn_kernel(TaParams a) {
    extern __shared__ float sm[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = a.L, Dh = a.Dh, DP = Dh + 1, LP = L + 1, HD = a.H * Dh;
    float* Q = sm; float* K = Q + L * DP; float* V = K + L * DP;
    float* dO = V + L * DP;                                      // backward only
    float* P = BWD ? dO + L * DP : V + L * DP;                   // forward: one row per wave (TA_NW x LP); backward: L x LP
    float* dS = P + L * LP;                                      // backward only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = a.qkv + (size_t)b * L * 3 * HD + h * Dh;
    for (int e = tid; e < L * Dh; e += TA_NT) {
        const int i = e / Dh, d = e - i * Dh;
        const float* r = src + (size_t)i * 3 * HD + d;
        Q[i * DP + d] = r[0]; K[i * DP + d] = r[HD]; V[i * DP + d] = r[2 * HD];
        if (BWD) dO[i * DP + d] = a.d_o[((size_t)b * L + i) * HD + h * Dh + d];
    }
    float* pg = a.p + ((size_t)b * a.H + h) * L * L;
    if (BWD)
        for (int e = tid; e < L * L; e += TA_NT) { const int i = e / L; P[i * LP + (e - i * L)] = pg[e]; }
    __syncthreads();
    constexpr int NP = 64 / DP2;                                 // lanes per d: the j (or i) range is dealt round-robin to NP parts
    const int d = lane % DP2, part = lane / DP2;
    const bool d_ok = d < Dh;

    // the lane's two score columns never change: their K (forward) / V (backward) rows stay in registers, a score costs one broadcast
    // LDS read per multiply-add instead of two
    constexpr bool REG = DP2 <= 32;
    float cr[2][REG ? DP2 : 1];
    if (REG) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < DP2; ++k) {
                const int j = lane + 64 * c;
                cr[c][k] = (j < L && k < Dh) ? (BWD ? V : K)[j * DP + k] : 0.f;
            }
    }
    auto col_dot = [&](const float* row, const float* M, int c) {   // sum_k row[k] * M[lane + 64 c][k]
        const int j = lane + 64 * c;
        float t = 0.f;
        if (REG) {
#pragma unroll
            for (int k = 0; k < DP2; ++k) t = fmaf(k < Dh ? row[k] : 0.f, cr[c][REG ? k : 0], t);
        } else if (j < L) {
            for (int k = 0; k < Dh; ++k) t = fmaf(row[k], M[j * DP + k], t);
        }
        return t;
    };

    if (!BWD) {
        float* prow = P + wave * LP;
        for (int i = wave; i < L; i += TA_NW) {
            float s[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float t = col_dot(Q + i * DP, K, c);
                s[c] = lane + 64 * c < L ? t * a.scale : -INFINITY;
            }
            const float m = wave_max(fmaxf(s[0], s[1]));
            const float e0 = lane < L ? __expf(s[0] - m) : 0.f, e1 = lane + 64 < L ? __expf(s[1] - m) : 0.f;
            const float inv = 1.f / wave_sum(e0 + e1);
            if (lane < L) { prow[lane] = e0 * inv; pg[(size_t)i * L + lane] = e0 * inv; }
            if (lane + 64 < L) { prow[lane + 64] = e1 * inv; pg[(size_t)i * L + lane + 64] = e1 * inv; }
            __builtin_amdgcn_wave_barrier();                     // the row is written and read by this wave only (LDS is in order per wave)
            float acc = 0.f;
            if (d_ok)
                for (int j = part; j < L; j += NP) acc = fmaf(prow[j], V[j * DP + d], acc);
            acc = part_sum<DP2>(acc);
            if (part == 0 && d_ok) a.o[((size_t)b * L + i) * HD + h * Dh + d] = acc;
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }

    float* dst = a.d_qkv + (size_t)b * L * 3 * HD + h * Dh;
    // ---- rows: dP = dO V^T, dS = P o (dP - rowsum(dP o P)); dQ = scale * dS K
    for (int i = wave; i < L; i += TA_NW) {
        float dp[2], pr[2];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            dp[c] = col_dot(dO + i * DP, V, c); pr[c] = j < L ? P[i * LP + j] : 0.f;
            dot = fmaf(dp[c], pr[c], dot);
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (lane + 64 * c < L) dS[i * LP + lane + 64 * c] = pr[c] * (dp[c] - dot);
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        if (d_ok)
            for (int j = part; j < L; j += NP) acc = fmaf(dS[i * LP + j], K[j * DP + d], acc);
        acc = part_sum<DP2>(acc);
        if (part == 0 && d_ok) dst[(size_t)i * 3 * HD + d] = acc * a.scale;
    }
    __syncthreads();
    // ---- columns: dK = scale * dS^T Q, dV = P^T dO
    for (int j = wave; j < L; j += TA_NW) {
        float ak = 0.f, av = 0.f;
        if (d_ok)
            for (int i = part; i < L; i += NP) {
                ak = fmaf(dS[i * LP + j], Q[i * DP + d], ak);
                av = fmaf(P[i * LP + j], dO[i * DP + d], av);
            }
        ak = part_sum<DP2>(ak); av = part_sum<DP2>(av);
        if (part == 0 && d_ok) { dst[(size_t)j * 3 * HD + HD + d] = ak * a.scale; dst[(size_t)j * 3 * HD + 2 * HD + d] = av; }
    }
}

size_t ta_smem(int L, int Dh, bool bwd) {
    const size_t rows = (size_t)L * (Dh + 1), lp = L + 1;
    return 4 * (bwd ? 4 * rows + 2 * (size_t)L * lp : 3 * rows + TA_NW * lp);
}

template <int DP2>
int launch_ta(const TaParams& a, bool bwd, hipStream_t st) {
    const size_t smem = ta_smem(a.L, a.Dh, bwd);
    dim3 grid(a.H, a.B), block(TA_NT);
    if (bwd) {
        auto k = token_attn_kernel<DP2, true>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, a);
    } else {
        auto k = token_attn_kernel<DP2, false>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, a);
    }
    return rs_check_launch();
}

}  // namespace

int rs_token_attn_supported(int L, int Dh) {
    return L >= 1 && L <= TA_MAXL && Dh >= 1 && Dh <= 64 && ta_smem(L, Dh, true) <= 160 * 1024;
}

int rs_launch_token_attn(const float* qkv, float* o, float* p, const float* d_o, float* d_qkv, int B, int L, int H, int Dh, float scale, hipStream_t st) {
    if (!rs_token_attn_supported(L, Dh)) return RS_ERR_UNSUPPORTED;
    const TaParams a = {qkv, o, p, d_o, d_qkv, B, L, H, Dh, scale};
    const bool bwd = d_qkv != nullptr;
    return Dh <= 16 ? launch_ta<16>(a, bwd, st) : Dh <= 32 ? launch_ta<32>(a, bwd, st) : launch_ta<64>(a, bwd, st);
}
