// 1x1x1 convolution / linear layer on channels-last activations as an MFMA GEMM (gfx950): the pointwise members of MedFormer's
// attention stages -- DepthwiseSeparableConv.pointwise (rsuper_train/model/dim3/conv_layers.py:126-157), MBConv expand / project
// (:197-239), the feat_qv / map_qv / out projections of BidirectionAttention (medformer_utils.py:13-99), SemanticMapGeneration's
// 1x1x1 members (:206-232) -- forward and data gradient.  Round 2 issued them as fp32 library GEMMs (rocBLAS / hipBLASLt: 45-70 TF/s,
// 480 launches, 6.2 ms of the 29 ms step).
//
//   forward        y[r][n] = sum_k x[r][k] * W[n][k] (+ b[n])        x: [R][ldx] f32, W: (Cout, Cin) f32 (state_dict layout)
//   data gradient  dx[r][k] = sum_n dy[r][n] * W[n][k]                the same kernel on the transposed weights (pack mode 1)
//
// These products are HBM-bound (K, N <= 1024 against 10^4..10^5 rows), so the kernel moves every byte once and keeps the matrix pipe
// out of the way:
//   * operand roles are swapped -- the WEIGHTS are the MFMA A operand (rows = output channels), the ACTIVATIONS the B operand (columns =
//     voxels): a lane's B fragment is 8 (bf16 compute) / 4 (f32 compute) consecutive channels of ONE voxel row, i.e. a plain 16 / 32-byte
//     global load of the row-major source, converted in registers -- no LDS staging, no transpose, no barrier in the whole kernel;
//   * the accumulator comes out with the voxel in the lane and four consecutive output channels per register quad: 16-byte stores
//     straight from the accumulators (+ bias), no LDS transpose either;
//   * weights are packed once per call into A-fragment order (pw_pack_kernel: [k-step][32-row tile][lane] x 16 B, converted to the
//     compute type) and stream through L2 as coalesced 1 KB wave loads.
// Compute type: bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) in the bf16 mode, exact-f32 MFMA (4 x v_mfma_f32_32x32x2_f32)
// in the f32 parity mode; storage stays fp32 either way (the attention stages' activation dtype).
#include "common.hpp"
#include "kernels.hpp"

namespace {

struct PwParams {
    const float* x; int ldx;       // [R][ldx] f32 (row = voxel)
    const void* wp;                // packed A fragments: [ksteps][ntiles][64 lanes] x 16 B
    const float* bias;             // [N] or nullptr
    const float* res; int ldr;     // [R][ldr] residual added after the bias (MBConv / attention shortcuts), or nullptr
    float* y; int ldy;             // [R][ldy] f32
    int R, K, N, ntiles, ksteps;
};

// W (rows x cols f32, row-major, leading dimension ldw) -> fragments of op(W): mode 0: A[n][k] = W[n][k] (forward, N = rows, K = cols),
// mode 1: A[n][k] = W[k][n] (data gradient, N = cols, K = rows).  Lane l of (kstep, ntile) holds A[ntile*32 + (l & 31)][kstep*2*KP + (l >> 5)*KP .. + KP).
template <typename CT>
__global__ __launch_bounds__(256) void pw_pack_kernel(const float* __restrict__ w, int rows, int cols, int mode, int ntiles, int ksteps, CT* out) {
    constexpr int KP = Elem<CT>::KP;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ksteps * ntiles * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) % ntiles, ks = (idx >> 6) / ntiles;
    const int n = nt * 32 + (lane & 31), k0 = ks * 2 * KP + (lane >> 5) * KP;
    const int N = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
    float f[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = k0 + j;
        f[j] = (n < N && k < K) ? (mode == 0 ? w[(size_t)n * cols + k] : w[(size_t)k * cols + n]) : 0.f;
    }
    *(uint4*)(out + (size_t)idx * KP) = pack16<CT>(f);
}

// The same packing for MANY weights in one launch (one MedFormer step packs 156 fragments sets: 0.8 ms of 5-us launches one by one).
// table: n rows of 8 x int64 {w pointer, byte offset of the fragments in `arena`, rows, cols, mode, ntiles, ksteps, first item}, items = 16-byte
// fragment slots numbered through all entries (row n holds the total); a thread finds its entry by bisection (the table stays in L2).
template <typename CT>
__global__ __launch_bounds__(256) void pw_pack_batch_kernel(const long long* __restrict__ table, int n, long total, char* arena) {
    constexpr int KP = Elem<CT>::KP;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int lo = 0, hi = n;                                       // largest e with first_item[e] <= idx
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (table[(size_t)mid * 8 + 7] <= idx) lo = mid; else hi = mid;
    }
    const long long* d = table + (size_t)lo * 8;
    const float* w = (const float*)d[0];
    const int rows = (int)d[2], cols = (int)d[3], mode = (int)d[4], ntiles = (int)d[5];
    const int li = (int)(idx - d[7]);
    const int lane = li & 63, nt = (li >> 6) % ntiles, ks = (li >> 6) / ntiles;
    const int nn = nt * 32 + (lane & 31), k0 = ks * 2 * KP + (lane >> 5) * KP;
    const int N = mode == 0 ? rows : cols, K = mode == 0 ? cols : rows;
    float f[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = k0 + j;
        f[j] = (nn < N && k < K) ? (mode == 0 ? w[(size_t)nn * cols + k] : w[(size_t)k * cols + nn]) : 0.f;
    }
    *(uint4*)(arena + d[1] + (size_t)li * 16) = pack16<CT>(f);
}

// block = 4 waves.  KSPLIT = false: wave = 32 voxel rows x NF*32 output channels, the block covers 128 rows (grid: row blocks of 128 x column
// blocks of NF*32).  KSPLIT = true (few rows, long reductions: the 12^3 / 6^3 stages -- 27 row blocks of 128 would leave the chip empty and every
// wave with a 64-step dependent chain): the four waves take a quarter of the k-steps each for the SAME 32 rows (grid: row blocks of 32) and
// waves 1-3 hand their partial tiles to wave 0 through LDS, which adds them in wave order (deterministic) and stores.
template <typename CT, int NF, bool KSPLIT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwParams p) {
    constexpr int KP = Elem<CT>::KP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = lane & 31, half = lane >> 5;
    const int row = (KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave) * 32 + v;
    const int nt0 = blockIdx.y * NF;
    const int kper = KSPLIT ? (p.ksteps + 3) / 4 : p.ksteps;                   // k-steps of this wave: [kbeg, kend)
    const int kbeg = KSPLIT ? wave * kper : 0;
    const int kend = KSPLIT ? (kbeg + kper < p.ksteps ? kbeg + kper : p.ksteps) : p.ksteps;
    // activations through a buffer descriptor: rows past R and channels past K read zeros (no branches around the loads)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)((size_t)p.R * p.ldx * 4), 0x00020000);
    const uint32_t xrow = row < p.R ? (uint32_t)row * (uint32_t)p.ldx * 4u : 0xFFFFFFFFu;
    const uint4* wp = (const uint4*)p.wp + (size_t)nt0 * 64 + lane;
    const size_t wstep = (size_t)p.ntiles * 64;

    f32x16_t acc[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;

    auto load_x = [&](int ks, uint4* raw) {                  // KP consecutive f32 channels of this lane's voxel: 1 (f32) / 2 (bf16) vectors
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {
            const int k = ks * 2 * KP + half * KP + q * 4;
            const uint32_t off = (xrow != 0xFFFFFFFFu && k < p.K) ? xrow + (uint32_t)k * 4u : 0xFFFFFFFFu;
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            raw[q] = make_uint4(t[0], t[1], t[2], t[3]);
        }
    };
    auto to_frag = [&](const uint4* raw) {
        float f[KP];
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {
            f[q * 4] = __uint_as_float(raw[q].x); f[q * 4 + 1] = __uint_as_float(raw[q].y);
            f[q * 4 + 2] = __uint_as_float(raw[q].z); f[q * 4 + 3] = __uint_as_float(raw[q].w);
        }
        return pack16<CT>(f);
    };
    // software pipeline: the operands of k-step s + 2 are requested while step s multiplies (memory-bound: keep loads in flight)
    constexpr int D = 2;
    uint4 xraw[D + 1][KP / 4], wq[D + 1][NF];
    auto issue = [&](int ks, int slot) {
        load_x(ks, xraw[slot]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) wq[slot][nf] = (nt0 + nf < p.ntiles) ? wp[(size_t)ks * wstep + nf * 64] : make_uint4(0, 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) if (kbeg + d < kend) issue(kbeg + d, d);
    int ks = kbeg;
    for (; ks + 3 <= kend; ks += 3) {                        // ring of three slots, unrolled so that the slot indices are static
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (ks + u + D < kend) issue(ks + u + D, (u + D) % 3);
            const uint4 xf = to_frag(xraw[u]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma32<CT>(acc[nf], wq[u][nf], xf);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {                            // tail (< 3 steps): step ks + u sits in slot u (static indices)
        if (ks + u < kend) {
            const uint4 xf = to_frag(xraw[u]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma32<CT>(acc[nf], wq[u][nf], xf);
        }
    }

    if constexpr (KSPLIT) {                                  // partial tiles of waves 1-3 -> wave 0 (16-byte lane-contiguous rows: conflict-free)
        __shared__ float4 xch[3][NF][4][64];
        if (wave > 0) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int q = 0; q < 4; ++q) xch[wave - 1][nf][q][lane] = make_float4(acc[nf][4 * q], acc[nf][4 * q + 1], acc[nf][4 * q + 2], acc[nf][4 * q + 3]);
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = xch[w][nf][q][lane];
                    acc[nf][4 * q] += t.x; acc[nf][4 * q + 1] += t.y; acc[nf][4 * q + 2] += t.z; acc[nf][4 * q + 3] += t.w;
                }
    }
    // epilogue: lane = voxel `row`, registers 4q .. 4q+3 of fragment nf = output channels (nt0 + nf)*32 + 8q + 4*half .. + 3
    if (row < p.R) {
        float* yrow = p.y + (size_t)row * p.ldy;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = (nt0 + nf) * 32 + 8 * q + 4 * half;
                if (n < p.N) {                               // N is a multiple of 4 (checked by the launcher)
                    float4 o = make_float4(acc[nf][4 * q], acc[nf][4 * q + 1], acc[nf][4 * q + 2], acc[nf][4 * q + 3]);
                    if (p.bias) { const float4 b = *(const float4*)(p.bias + n); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
                    if (p.res) { const float4 r = *(const float4*)(p.res + (size_t)row * p.ldr + n); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
                    *(float4*)(yrow + n) = o;
                }
            }
        }
    }
}

template <typename CT>
int launch_pw(const PwParams& p, hipStream_t st) {
    const int nf = p.ntiles >= 4 ? 4 : p.ntiles >= 2 ? 2 : 1;
    const unsigned gy = (unsigned)((p.ntiles + nf - 1) / nf);
    // split the reduction over the waves when whole-K blocks of 128 rows would cover less than half of the CUs
    const bool ksplit = p.ksteps >= 8 && (long)((p.R + 127) / 128) * gy < 128;   // measured: 216 whole-K blocks beat 864 split ones (26 vs 37 us), 54 lose (37 vs 17 us)
    dim3 grid((unsigned)((p.R + (ksplit ? 31 : 127)) / (ksplit ? 32 : 128)), gy), block(256);
    if (ksplit) {
        if (nf == 4) hipLaunchKernelGGL((pw_gemm_kernel<CT, 4, true>), grid, block, 0, st, p);
        else if (nf == 2) hipLaunchKernelGGL((pw_gemm_kernel<CT, 2, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((pw_gemm_kernel<CT, 1, true>), grid, block, 0, st, p);
    } else {
        if (nf == 4) hipLaunchKernelGGL((pw_gemm_kernel<CT, 4, false>), grid, block, 0, st, p);
        else if (nf == 2) hipLaunchKernelGGL((pw_gemm_kernel<CT, 2, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((pw_gemm_kernel<CT, 1, false>), grid, block, 0, st, p);
    }
    return rs_check_launch();
}


// ------------------------------------------------------------------------------------------------ weight (+ bias) gradient
//   dW[n][k] = sum_r dy[r][n] * x[r][k],   db[n] = sum_r dy[r][n]            (dy: [R][ldy] f32, x: [R][ldx] f32, dW: (N, K) as the state_dict)
// The reduction runs over the SLOW memory axis of both operands, so a lane's fragment (KP consecutive reduction steps of one channel) is KP
// dword loads, each a coalesced 128-byte row segment across the 32 lanes of a half-wave -- no LDS transpose, no barrier.  Wave = 64 x 64
// output tile (2 x 2 fragments: 4 MFMAs per 4 fragments), block = 4 waves as WNW x (4 / WNW) tiles, blockIdx.z = slab of the rows; every
// slab writes its partial dW (and the column sums of dy, accumulated from the A fragments already in registers) with plain stores and
// pw_wgrad_reduce_kernel adds the slabs in slab order (deterministic; round 2: torch.bmm over slabs + .sum(0), and a ones-row GEMM for db).
struct PwWgParams {
    const float* dy; int ldy;
    const float* x; int ldx;
    float* part;                   // [S][N*K + N]
    int R, N, K, rows_per, bias;
    float* out_w; float* out_b;    // one slab only: the block tiles write dW / db themselves (no reduce launch)
};

template <typename CT, int WNW>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(PwWgParams p) {
    constexpr int KP = Elem<CT>::KP, RS = 2 * KP;                              // rows per MFMA step: 16 (bf16) / 8 (f32)
    constexpr int WKW = 4 / WNW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int wn = wave / WKW, wk = wave % WKW;
    const int n0 = (blockIdx.x * WNW + wn) * 64, k0 = (blockIdx.y * WKW + wk) * 64;
    const int rbeg = blockIdx.z * p.rows_per;
    const int rend = rbeg + p.rows_per < p.R ? rbeg + p.rows_per : p.R;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (uint32_t)((size_t)p.R * p.ldy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)((size_t)p.R * p.ldx * 4), 0x00020000);
    // channels past N / K are clamped to a valid column: they only reach accumulator rows / columns that are never stored
    uint32_t an[2], bk[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int n = n0 + m * 32 + c, k = k0 + m * 32 + c;
        an[m] = (uint32_t)(n < p.N ? n : p.N - 1) * 4u;
        bk[m] = (uint32_t)(k < p.K ? k : p.K - 1) * 4u;
    }
    // rows past the slab read zeros: an offset past num_records that cannot wrap when the column is added (the launcher checks (R + 1) * ld * 4 < 2^32)
    const uint32_t oob_y = 0xFFFFFFFFu - (uint32_t)p.ldy * 4u, oob_x = 0xFFFFFFFFu - (uint32_t)p.ldx * 4u;
    const uint32_t sy = (uint32_t)p.ldy * 4u, sx = (uint32_t)p.ldx * 4u;

    f32x16_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    float bs[2] = {0.f, 0.f};
    const bool do_bias = p.bias && blockIdx.y == 0 && wk == 0;

    float ya[2][2][KP], xb[2][2][KP];                                         // [slot][fragment][reduction step]
    auto issue = [&](int r0, int slot) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int r = r0 + half * KP + j;
            const uint32_t oy = r < rend ? (uint32_t)r * sy : oob_y, ox = r < rend ? (uint32_t)r * sx : oob_x;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ya[slot][m][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, oy + an[m], 0, 0));
                xb[slot][m][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ox + bk[m], 0, 0));
            }
        }
    };
    auto compute = [&](int slot) {
        uint4 a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < KP; ++j) bs[m] += ya[slot][m][j];
            }
            a[m] = pack16<CT>(ya[slot][m]);
            b[m] = pack16<CT>(xb[slot][m]);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) mma32<CT>(acc[m][q], a[m], b[q]);
    };
    if (n0 < p.N && k0 < p.K) {                                                // wave-uniform: tiles wholly outside the matrix do nothing
        int r = rbeg;
        if (r < rend) issue(r, 0);
        for (; r + RS < rend; r += 2 * RS) {                                   // two steps per trip: static slot indices
            issue(r + RS, 1);
            compute(0);
            if (r + 2 * RS < rend) issue(r + 2 * RS, 0);
            compute(1);
        }
        if (r < rend) compute(0);
    }
    // C layout: column (x channel) = lane & 31, row (dy channel) of register reg = (reg & 3) + 8 * (reg >> 2) + 4 * half
    float* part = p.out_w ? p.out_w : p.part + (size_t)blockIdx.z * ((size_t)p.N * p.K + p.N);
    float* bpart = p.out_w ? p.out_b : part + (size_t)p.N * p.K;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = k0 + q * 32 + c;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = n0 + m * 32 + cd_row32(reg, lane);
                if (n < p.N && k < p.K) part[(size_t)n * p.K + k] = acc[m][q][reg];
            }
        }
    if (do_bias) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float t = bs[m] + __shfl_xor(bs[m], 32);
            const int n = n0 + m * 32 + c;
            if (half == 0 && n < p.N) bpart[n] = t;
        }
    }
}

// dW / db = sum over the slabs, in slab order.  One float4 of the (N*K + N)-element slab per thread, four slab loads in flight.
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ part, int S, long E, long NK, float* dw, float* db) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + 4 <= S; i += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(part + (size_t)(i + u) * E + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; i < S; ++i) { const float4 v = *(const float4*)(part + (size_t)i * E + e); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (e < NK) *(float4*)(dw + e) = s;
    else if (db) *(float4*)(db + (e - NK)) = s;
}

template <typename CT>
int launch_pw_wgrad(PwWgParams p, int S, float* dw, float* db, hipStream_t st) {
    p.out_w = S == 1 ? dw : nullptr; p.out_b = S == 1 ? db : nullptr;
    const int wnw = p.N <= 64 ? 1 : p.K <= 64 ? 4 : 2;
    dim3 grid((unsigned)((p.N + wnw * 64 - 1) / (wnw * 64)), (unsigned)((p.K + (4 / wnw) * 64 - 1) / ((4 / wnw) * 64)), (unsigned)S), block(256);
    if (wnw == 1) hipLaunchKernelGGL((pw_wgrad_kernel<CT, 1>), grid, block, 0, st, p);
    else if (wnw == 4) hipLaunchKernelGGL((pw_wgrad_kernel<CT, 4>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((pw_wgrad_kernel<CT, 2>), grid, block, 0, st, p);
    const long NK = (long)p.N * p.K, E = NK + p.N;
    if (S > 1) hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((unsigned)((E / 4 + 255) / 256)), dim3(256), 0, st, (const float*)p.part, S, E, NK, dw, db);
    return rs_check_launch();
}

}  // namespace

size_t rs_pw_packed_bytes(int N, int K, int dtype) {
    const int KS = dtype == RS_F32 ? 8 : 16;
    return (size_t)((K + KS - 1) / KS) * ((N + 31) / 32) * 64 * 16;
}

// mode 0: y = x W^T (+ bias), W (N, K);  mode 1: y = x W, W (K, N) (the data gradient of mode 0 with x := dy)
int rs_launch_pointwise(int dtype, int mode, const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr,
                        float* y, int ldy, int R, int K, int N, void* packed, hipStream_t st) {
    const int KS = dtype == RS_F32 ? 8 : 16;
    PwParams p;
    p.x = x; p.ldx = ldx; p.wp = packed; p.bias = bias; p.res = res; p.ldr = ldr; p.y = y; p.ldy = ldy; p.R = R; p.K = K; p.N = N;
    p.ntiles = (N + 31) / 32; p.ksteps = (K + KS - 1) / KS;
    const int items = p.ksteps * p.ntiles * 64;
    const int rows = mode == 0 ? N : K, cols = mode == 0 ? K : N;
    if (dtype == RS_F32) {
        if (w) hipLaunchKernelGGL(pw_pack_kernel<float>, dim3((items + 255) / 256), dim3(256), 0, st, w, rows, cols, mode, p.ntiles, p.ksteps, (float*)packed);
        return launch_pw<float>(p, st);
    }
    if (w) hipLaunchKernelGGL(pw_pack_kernel<bf16_t>, dim3((items + 255) / 256), dim3(256), 0, st, w, rows, cols, mode, p.ntiles, p.ksteps, (bf16_t*)packed);
    return launch_pw<bf16_t>(p, st);
}

int rs_launch_pointwise_pack_batch(int dtype, const long long* table, int n, long total_items, void* arena, hipStream_t st) {
    const dim3 grid((unsigned)((total_items + 255) / 256)), block(256);
    if (dtype == RS_F32) hipLaunchKernelGGL(pw_pack_batch_kernel<float>, grid, block, 0, st, table, n, total_items, (char*)arena);
    else hipLaunchKernelGGL(pw_pack_batch_kernel<bf16_t>, grid, block, 0, st, table, n, total_items, (char*)arena);
    return rs_check_launch();
}

// slabs of the row axis for the weight gradient: about one resident block per CU (more slabs = more partial traffic for the reduce), at least 64 rows per slab, whole MFMA steps per slab
int rs_pw_wgrad_splits(int R, int N, int K) {
    const int wnw = N <= 64 ? 1 : K <= 64 ? 4 : 2;
    const int tiles = ((N + wnw * 64 - 1) / (wnw * 64)) * ((K + (4 / wnw) * 64 - 1) / ((4 / wnw) * 64));
    static const int target = getenv("RSUPER_PW_WG_TARGET") ? atoi(getenv("RSUPER_PW_WG_TARGET")) : 256;   // blocks per launch (measured: 128 / 256 / 512 / 1024 -> MedFormer step 28.6 / 28.5 / 28.9 / 29.4 ms)
    static const int direct = getenv("RSUPER_PW_WG_DIRECT") ? atoi(getenv("RSUPER_PW_WG_DIRECT")) : 0;   // rows up to which ONE slab is forced (a single slab writes dW itself, no reduce launch): 768 measured no faster
                                                                                                      // than 7 slabs + reduce at 432 rows (MedFormer 27.2 vs 27.0 ms); a naturally single slab (<= 64 rows) always writes directly
    if (R <= direct) return 1;
    int s = target / tiles;
    if (s > (R + 63) / 64) s = (R + 63) / 64;
    if (s < 1) s = 1;
    const int rows_per = (((R + s - 1) / s) + 15) / 16 * 16;                      // as the launcher cuts them: no slab without rows
    return (R + rows_per - 1) / rows_per;
}

int rs_launch_pointwise_wgrad(int dtype, const float* dy, int ldy, const float* x, int ldx, int R, int N, int K, float* part, int S,
                              float* dw, float* db, hipStream_t st) {
    PwWgParams p;
    p.dy = dy; p.ldy = ldy; p.x = x; p.ldx = ldx; p.part = part; p.R = R; p.N = N; p.K = K; p.bias = db != nullptr;
    p.rows_per = (((R + S - 1) / S) + 15) / 16 * 16;
    // (a slab past the last row still writes its -- zero -- partial: any S is valid, rs_pw_wgrad_splits just avoids empty ones)
    return dtype == RS_F32 ? launch_pw_wgrad<float>(p, S, dw, db, st) : launch_pw_wgrad<bf16_t>(p, S, dw, db, st);
}
