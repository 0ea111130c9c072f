// 3x3x3 / STRIDE 2 / pad 1 convolution (forward) and its data gradient (a transposed convolution) as implicit GEMMs on MFMA, gfx950.
//
// The strided member of down_block(pool=False): BasicBlock(in, out, stride=2) -- its conv1 and shortcut convolutions
// (rsuper_train/model/dim3/unet_utils.py:38-39, conv_layers.py:29-38,82-84).  Rounds 1-2 evaluated them at full resolution and kept every
// second voxel (8x the MFMA work, a full-resolution temporary); here both directions do the minimal work by PARITY CLASSES:
//
//   forward   y[o]      = sum_t W[t]   x_hat[2o + t - 1]        per axis: t = 1 reads the even input 2o; t = 0 / 2 read the odd inputs 2(o-1)+1 / 2o+1
//   dgrad     dx[2q+r]  = sum_t W[t]^T dy[q + (r + 1 - t)/2]    per axis: r = 0 takes t = 1 (dy[q]); r = 1 takes t = 0 (dy[q+1]) and t = 2 (dy[q])
//
// so with the input (forward) / the output (dgrad) split into its 8 parity classes (bit per axis), class c needs 1, 2, 4 or 8 of the 27 taps
// -- 27 (class, tap) pairs in all, each a plain stride-1 tap on the HALF-resolution grid with offsets in {-1, 0} (forward) / {0, +1} (dgrad).
// Both kernels tile the half-resolution grid in 4x4x16 bricks (GEMM rows), stage a haloed brick per (class, 32-channel chunk) like
// conv3d_igemm.hip, and run only that class's taps, chosen at compile time (the 8 classes are the cases of a switch around one fully
// unrolled MFMA sequence each, 54 MFMA steps per chunk in total: the code size of one dense 27-tap body):
//   * forward: ONE accumulator over the 8 classes of every chunk; the staging reads x at 2q + class with the InstanceNorm + ReLU prologue
//     (rows a class does not need -- its halo side when the class bit is 0 -- are not fetched); epilogue = the stride-1 forward one
//     (statistics of the output for the next InstanceNorm).  Weights: the ordinary forward fragments (tap index = t).
//   * dgrad: a block owns a 2x4x16 brick of the half grid and ALL 8 classes of its outputs (the 8x8x32 full-resolution voxels 2q + class): one
//     stride-1 staging of [dy1 | dOut] per chunk feeds the 27 (class, tap) pairs -- every A fragment (8 halo positions x 2 k-steps) is read
//     once and used by all the pairs that share its position -- into 8 accumulator sets (2 fragments each per wave); the epilogue walks the
//     classes, scatters to 2q + class, applies the ReLU mask of the forward input there and adds up the InstanceNorm-backward partial sums
//     (one row per brick).  Weights: the ordinary data-gradient fragments (flipped taps: fragment index 26 - t).
// The weight gradient of the strided convolutions still runs on the zero-stuffed full-resolution dy (conv3d_wgrad.hip; 8x the minimal work
// of that third of the block's convolution FLOPs) -- DESIGN.md 8.
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"

#ifndef S2_SKIP
#define S2_SKIP 0
#endif
namespace {

constexpr int TH = 4, TW = 16;
constexpr int HH = TH + 2, HW = TW + 2;
constexpr int PITCH = 80;
template <int MODE> struct Geo {
    static constexpr int TD = MODE == 2 ? 2 : 4;                 // brick depth on the half grid
    static constexpr int MF = TD;                                // fragments per wave (2 M halves x MF = TD * 2 fragments of 2 x 16 voxels)
    static constexpr int HD = TD + 2;
    static constexpr int HROWS = HD * HH * HW;
    static constexpr int HALO_BYTES = HROWS * PITCH;
    static constexpr int NVEC = (HROWS * 4 + 255) / 256;
    static constexpr int NACC = MODE == 2 ? 8 : 1;               // accumulator sets (dgrad: one per output class)
};

// per-axis (halo position a, weight tap b) pairs of a class bit.  MODE 1 forward: bit 0 -> (1, 1); bit 1 -> (0, 0), (1, 2).
//                                                               MODE 2 dgrad  : bit 0 -> (1, 1); bit 1 -> (2, 2), (1, 0)   [fragment 2 - t]
template <int MODE> __host__ __device__ constexpr int ax_a(int bit, int i) { return bit == 0 ? 1 : (MODE == 1 ? (i == 0 ? 0 : 1) : (i == 0 ? 2 : 1)); }
template <int MODE> __host__ __device__ constexpr int ax_b(int bit, int i) { return bit == 0 ? 1 : (MODE == 1 ? (i == 0 ? 0 : 2) : (i == 0 ? 2 : 0)); }

// MODE 1: forward (EPI 0), MODE 2: data gradient (EPI 1).  Block = 4 waves as 2 (M halves: 4 fragments each) x 2 (N halves: NF x 32 columns).
template <typename T, int MODE, int NF>
__global__ __launch_bounds__(256, 1) void igemm_s2_kernel(IgemmParams p, int FD, int FH, int FW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Geo<MODE> G;
    constexpr int TD = G::TD, HD = G::HD, HROWS = G::HROWS, NVEC = G::NVEC, NACC = G::NACC;
    constexpr int KC = Elem<T>::KC, KP = Elem<T>::KP;
    constexpr int MF = G::MF, WN = 2, BN32 = WN * NF, BN = BN32 * 32;
    constexpr int EPF = BN + 4;                                                 // epilogue scratch row pitch (floats)
    constexpr int FP = (MF * 2 * 32 * EPF * 4 <= G::HALO_BYTES + 1024) ? MF * 2 : 4;   // fragments per epilogue pass
    constexpr int SCR_BYTES = FP * 32 * EPF * 4;
    constexpr int LDS_MAIN = SCR_BYTES > G::HALO_BYTES ? SCR_BYTES : G::HALO_BYTES;
    char* halo = smem;
    float* mr_lds = (float*)(smem + LDS_MAIN);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tile_id = t;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int d0 = t * TD, h0 = th * TH, w0 = tw * TW;
    const int n = blockIdx.z;
    const int colgrp = blockIdx.y;
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = nchA + nchB;
    const bool normA = p.a.mr != nullptr, normB = p.b.mr != nullptr;
    if (normA) for (int i = tid; i < 2 * p.a.C; i += 256) mr_lds[i] = p.a.mr[(size_t)n * 2 * p.a.C + i];
    if (normB) for (int i = tid; i < 2 * p.b.C; i += 256) mr_lds[2 * p.a.C + i] = p.b.mr[(size_t)n * 2 * p.b.C + i];

    int hs, wl;
    row_to_hw(lane & 31, hs, wl);
    const int a_base = (((wm * MF / 2) * HH + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
    auto a_const = [](int mf) { return (((mf >> 1) * HH + (mf & 1) * 2) * HW) * PITCH; };

    f32x16_t acc[NACC][MF][NF];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][mf][nf][r] = 0.f;

    const uint4* wp = (const uint4*)p.wp;
    const int ntile0 = colgrp * BN32 + wn * NF;
    const size_t wstep = (size_t)p.ntiles * 64;

    // staging: vector i of this thread = halo row (tid >> 2) + 64 i, 16-byte slot tid & 3; (hd, hh, hw) per vector are fixed
    const int slot = tid & 3;
    // Per-thread constants of the staging (vector i = halo row (tid >> 2) + 64 i): the voxel index of the row for class 0 and, per axis and class
    // bit, one bit per vector saying whether that row is inside the source volume AND needed by the class (a class whose bit is 0 on an axis only
    // uses the centre tap there: its halo rows on that axis are never read) -- an item then costs three ANDs plus an add / select / multiply per
    // vector instead of re-deriving coordinates and bounds (26.7 VALU per MFMA in the first version, profiles/r03_pmc_new_kernels.md).
    const int SD = MODE == 1 ? FD : p.D, SH = MODE == 1 ? FH : p.H, SW = MODE == 1 ? FW : p.W;      // source grid: full resolution (forward) / half (dgrad)
    constexpr int SS = MODE == 1 ? 2 : 1;                                        // source stride of a halo step
    int vbase[NVEC];
    uint32_t okd[2] = {0, 0}, okh[2] = {0, 0}, okw[2] = {0, 0}, vrow = 0;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const int hd = r / (HH * HW), rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int d = SS * (d0 - 1 + hd), h = SS * (h0 - 1 + hh), w = SS * (w0 - 1 + hw);
        vbase[i] = ((n * SD + d) * SH + h) * SW + w;
        vrow |= r < HROWS ? (1u << i) : 0u;
#pragma unroll
        for (int b = 0; b < 2; ++b) {                                            // b = class bit of the axis (dgrad: always 0)
            const bool nd = MODE != 1 || b || (hd >= 1 && hd <= TD), nh = MODE != 1 || b || (hh >= 1 && hh <= TH), nw = MODE != 1 || b || (hw >= 1 && hw <= TW);
            okd[b] |= (nd && d + b >= 0 && d + b < SD) ? (1u << i) : 0u;
            okh[b] |= (nh && h + b >= 0 && h + b < SH) ? (1u << i) : 0u;
            okw[b] |= (nw && w + b >= 0 && w + b < SW) ? (1u << i) : 0u;
        }
    }
    char* lds_st = halo + (tid >> 2) * PITCH + slot * 16;
    const uint32_t nvox_src = (uint32_t)(p.N * SD * SH * SW);
    uint4 pre[NVEC];
    uint32_t okmask = 0;
    auto stage = [&](int ch, int cls) __attribute__((always_inline)) {
        const bool isB = MODE == 2 && ch >= nchA;                               // field selects, not a reference select: `isB ? p.b : p.a` made the
        const void* sx = isB ? p.b.x : p.a.x;                                   // compiler keep a copy of the kernel arguments in scratch memory
        const int sld = isB ? p.b.ld : p.a.ld, sC = isB ? p.b.C : p.a.C;
        const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
        const uint32_t rowb = (uint32_t)sld * (uint32_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sx, 0, nvox_src * rowb, 0x00020000);
        const int cd = MODE == 1 ? (cls >> 2) & 1 : 0, chh = MODE == 1 ? (cls >> 1) & 1 : 0, cw = MODE == 1 ? cls & 1 : 0;
        const int coff = (cd * SH + chh) * SW + cw;                              // voxel offset of the class inside its 2x2x2 cell
        okmask = (c < sC) ? (vrow & okd[cd] & okh[chh] & okw[cw]) : 0u;
        const uint32_t cb = (uint32_t)c * (uint32_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const uint32_t off = ((okmask >> i) & 1u) ? (uint32_t)(vbase[i] + coff) * rowb + cb : 0xFFFFFFFFu;
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            pre[i] = make_uint4(q[0], q[1], q[2], q[3]);
        }
    };
    auto commit = [&](int ch) __attribute__((always_inline)) {
        const bool isB = MODE == 2 && ch >= nchA;
        const int sC = isB ? p.b.C : p.a.C;
        const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
        const bool norm = (isB ? normB : normA) && c < sC;
        float sc_[KP], nb_[KP];
        if (norm) {
            const float* mr = mr_lds + 2 * ((isB ? p.a.C : 0) + c);
#pragma unroll
            for (int j = 0; j < KP; ++j) { sc_[j] = mr[2 * j + 1]; nb_[j] = -mr[2 * j] * mr[2 * j + 1]; }
        }
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            uint4 q = pre[i];
            if (norm && ((okmask >> i) & 1u)) q = norm_relu16<T>(q, sc_, nb_);     // zero padding is applied AFTER the activation
            if ((vrow >> i) & 1u) *(uint4*)(lds_st + i * (64 * PITCH)) = q;
        }
    };
    // Forward: one (chunk, class) item = the class's rows staged, then its taps, fully unrolled (per-axis pair lists from ax_a / ax_b).  The
    // 8 items of a chunk are a static sequence: while item c multiplies, the rows AND the weight fragments (<= 16) of item c + 1 are in flight
    // (register staging `pre`, two fragment sets) -- fetched at their use, every fragment was an L2 round trip the matrix pipe waited for, and
    // every item a full memory latency with one 4-wave block per CU.
    auto load_b = [&](auto clsc, const uint4* wch, uint4 (&bq)[16][NF]) __attribute__((always_inline)) {
        constexpr int CLS = decltype(clsc)::value;
        constexpr int bd = (CLS >> 2) & 1, bh = (CLS >> 1) & 1, bw = CLS & 1;
#pragma unroll
        for (int id = 0; id <= bd; ++id)
#pragma unroll
            for (int ih = 0; ih <= bh; ++ih)
#pragma unroll
                for (int iw = 0; iw <= bw; ++iw) {
                    const int ti = (id * (bh + 1) + ih) * (bw + 1) + iw;
                    const int tapb = (ax_b<MODE>(bd, id) * 3 + ax_b<MODE>(bh, ih)) * 3 + ax_b<MODE>(bw, iw);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf) bq[ti * 2 + ks][nf] = wch[(size_t)(tapb * 2 + ks) * wstep + nf * 64];
                }
    };
    auto mfma_class = [&](auto clsc, uint4 (&bq)[16][NF]) __attribute__((always_inline)) {
        constexpr int CLS = decltype(clsc)::value;
        constexpr int bd = (CLS >> 2) & 1, bh = (CLS >> 1) & 1, bw = CLS & 1;
#pragma unroll
        for (int id = 0; id <= bd; ++id)
#pragma unroll
            for (int ih = 0; ih <= bh; ++ih)
#pragma unroll
                for (int iw = 0; iw <= bw; ++iw) {
                    const int ti = (id * (bh + 1) + ih) * (bw + 1) + iw;
                    const int aoff = ((ax_a<MODE>(bd, id) * HH + ax_a<MODE>(bh, ih)) * HW + ax_a<MODE>(bw, iw)) * PITCH;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        uint4 aq[MF];
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf) aq[mf] = *(const uint4*)(halo + a_base + a_const(mf) + aoff + ks * 32);
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf) mma32<T>(acc[0][mf][nf], aq[mf], bq[ti * 2 + ks][nf]);
                    }
                }
    };
    uint4 fbqA[16][NF], fbqB[16][NF];                                          // fragments of the even / odd classes
    // item CLS of chunk ch: its rows are in `pre`, its fragments in fbq[CLS & 1]; requests item CLS + 1 (or class 0 of the next chunk)
    auto item = [&](auto clsc, int ch, const uint4* wch) __attribute__((always_inline)) {
        constexpr int CLS = decltype(clsc)::value;
        __syncthreads();                                                        // previous item consumed (and mr_lds visible)
        if (!(S2_SKIP & 4)) commit(ch);
        __syncthreads();
        if constexpr (CLS < 7) {
            if (!(S2_SKIP & 4)) stage(ch, CLS + 1);
            if constexpr (CLS & 1) load_b(std::integral_constant<int, (CLS + 1) & 7>{}, wch, fbqA);
            else load_b(std::integral_constant<int, (CLS + 1) & 7>{}, wch, fbqB);
        } else if (ch + 1 < nch) {
            if (!(S2_SKIP & 4)) stage(ch + 1, 0);
            load_b(std::integral_constant<int, 0>{}, wch + (size_t)27 * 2 * wstep, fbqA);
        }
        __builtin_amdgcn_sched_barrier(0);                                      // the requests stay ahead of this item's MFMAs
        if (!(S2_SKIP & 2)) {
            if constexpr (CLS & 1) mfma_class(clsc, fbqB);
            else mfma_class(clsc, fbqA);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // dgrad: all 27 (class, tap) pairs of a chunk from one staged brick, in 16 groups = (halo position (1 or 2 per axis), k-step): the group's A
    // fragments are read once and serve every pair that uses the position -- per axis position 1 <- (class bit 0, fragment 1) and (class bit 1,
    // fragment 0), position 2 <- (class bit 1, fragment 2) -- and the weight fragments of group g + 1 (<= 8) are in flight while group g multiplies.
    auto group_pairs = [&](int g, auto&& fn) {                                  // fn(pair index, class, weight fragment) for every pair of group g
        const int pd = 1 + (g >> 3), ph = 1 + ((g >> 2) & 1), pw = 1 + ((g >> 1) & 1), ks = g & 1;
        int j = 0;
#pragma unroll
        for (int od_ = 0; od_ < (pd == 1 ? 2 : 1); ++od_)
#pragma unroll
            for (int oh_ = 0; oh_ < (ph == 1 ? 2 : 1); ++oh_)
#pragma unroll
                for (int ow_ = 0; ow_ < (pw == 1 ? 2 : 1); ++ow_) {
                    const int cd = pd == 1 ? od_ : 1, chh = ph == 1 ? oh_ : 1, cw = pw == 1 ? ow_ : 1;        // class bits
                    const int fd = pd == 1 ? (cd ? 0 : 1) : 2, fh = ph == 1 ? (chh ? 0 : 1) : 2, fw = pw == 1 ? (cw ? 0 : 1) : 2;
                    fn(j, cd * 4 + chh * 2 + cw, ((fd * 3 + fh) * 3 + fw) * 2 + ks);
                    ++j;
                }
    };
    auto mfma_all = [&](const uint4* wch) __attribute__((always_inline)) {
        uint4 bq[2][8][NF];
        group_pairs(0, [&](int j, int, int frag) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bq[0][j][nf] = wch[(size_t)frag * wstep + nf * 64];
        });
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16)
                group_pairs(g + 1, [&](int j, int, int frag) {
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) bq[(g + 1) & 1][j][nf] = wch[(size_t)frag * wstep + nf * 64];
                });
            __builtin_amdgcn_sched_barrier(0);                                  // keep the requests ahead of this group's MFMAs (the scheduler
                                                                                // otherwise sinks every load to its use: one L2 round trip per pair)
            const int pd = 1 + (g >> 3), ph = 1 + ((g >> 2) & 1), pw = 1 + ((g >> 1) & 1), ks = g & 1;
            const int aoff = ((pd * HH + ph) * HW + pw) * PITCH;
            uint4 aq[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) aq[mf] = *(const uint4*)(halo + a_base + a_const(mf) + aoff + ks * 32);
            group_pairs(g, [&](int j, int cls, int) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) mma32<T>(acc[cls % NACC][mf][nf], aq[mf], bq[g & 1][j][nf]);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int ch = 0; ch < nch; ++ch) {
        const uint4* wch = wp + (size_t)ch * 27 * 2 * wstep + (size_t)ntile0 * 64 + lane;
        if constexpr (MODE == 1) {
            if (ch == 0) {
                if (!(S2_SKIP & 4)) stage(0, 0);
                load_b(std::integral_constant<int, 0>{}, wch, fbqA);
            }
            item(std::integral_constant<int, 0>{}, ch, wch); item(std::integral_constant<int, 1>{}, ch, wch);
            item(std::integral_constant<int, 2>{}, ch, wch); item(std::integral_constant<int, 3>{}, ch, wch);
            item(std::integral_constant<int, 4>{}, ch, wch); item(std::integral_constant<int, 5>{}, ch, wch);
            item(std::integral_constant<int, 6>{}, ch, wch); item(std::integral_constant<int, 7>{}, ch, wch);
        } else {
            if (!(S2_SKIP & 4)) { if (ch == 0) stage(0, 0); }
            __syncthreads();
            if (!(S2_SKIP & 4)) commit(ch);
            __syncthreads();
            if (!(S2_SKIP & 4)) { if (ch + 1 < nch) stage(ch + 1, 0); }         // the next chunk's loads fly during this chunk's MFMAs
            if (!(S2_SKIP & 2)) mfma_all(wch);
        }
    }

    // ------------------------------------------------------------------ epilogue (as conv3d_igemm.hip; MODE 2 walks the 8 classes and scatters)
    __syncthreads();
    constexpr int EPI = MODE == 2 ? 1 : 0;
    constexpr int NFRAG = 2 * MF;
    constexpr int NPASS = NFRAG / FP;
    constexpr int CG = BN / KP;
    constexpr int RPT = 256 / CG;
    constexpr int NV = FP * 32 / RPT;
    float* sc2 = (float*)smem;
    const int col_l = lane & 31, hi = lane >> 5;
    const int cg = tid % CG, pr0 = tid / CG;
    const int col0 = colgrp * BN + cg * KP;
    const bool cok = col0 < p.Cout;
    const ConvSrc& es = p.ea;
    const int ecol0 = col0;
    float emu[KP], ers[KP];
    if (EPI == 1 && cok) {
#pragma unroll
        for (int j = 0; j < KP; ++j) { emu[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2]; ers[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2 + 1]; }
    }
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int LD = MODE == 2 ? FD : p.D, LH = MODE == 2 ? FH : p.H, LW = MODE == 2 ? FW : p.W;      // grid of the stores
#pragma unroll
    for (int c = 0; c < ((S2_SKIP & 1) ? 1 : NACC); ++c) {
        const int od = (c >> 2) & 1, oh = (c >> 1) & 1, ow = c & 1;
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            uint4 xq[NV];                                                       // EPI 1: forward input at the store positions, requested before the scratch pass
            if (EPI == 1) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int pr = pr0 + j * RPT;
                    const int d = d0 + q * (FP / 2) + (pr >> 6), h = h0 + ((pr >> 4) & 3), w = w0 + (pr & 15);
                    const int sd = 2 * d + od, sh = 2 * h + oh, sw = 2 * w + ow;
                    const bool ok = cok && d < p.D && h < p.H && w < p.W && sd < LD && sh < LH && sw < LW;
                    const uint32_t vox = ok ? (uint32_t)(((n * LD + sd) * LH + sh) * LW + sw) : 0u;
                    xq[j] = *(const uint4*)((const T*)es.x + (size_t)(vox * (uint32_t)es.ld + (uint32_t)(cok ? ecol0 : 0)));
                }
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int f = wm * MF + mf;
                if (f / FP == q) {
                    const int rowbase = (((f - q * FP) >> 1) * TH + (f & 1) * 2) * TW;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int hw0 = row_hw_packed((r & 3) + 8 * (r >> 2)), hw1 = row_hw_packed((r & 3) + 8 * (r >> 2) + 4);
                        const int hw = hi ? hw1 : hw0;
                        float* dst = sc2 + (rowbase + (hw >> 4) * TW + (hw & 15)) * EPF + wn * NF * 32 + col_l;
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf) dst[nf * 32] = acc[c][mf][nf][r];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int pr = pr0 + j * RPT;
                const int d = d0 + q * (FP / 2) + (pr >> 6), h = h0 + ((pr >> 4) & 3), w = w0 + (pr & 15);  // half-resolution voxel
                // MODE 1 stores at the half-resolution voxel; MODE 2 at 2q + class of the full-resolution grid
                const int sd = MODE == 2 ? 2 * d + od : d, sh = MODE == 2 ? 2 * h + oh : h, sw = MODE == 2 ? 2 * w + ow : w;
                if (cok && d < p.D && h < p.H && w < p.W && sd < LD && sh < LH && sw < LW) {
                    float v[KP];
                    const float4* sp = (const float4*)(sc2 + pr * EPF + cg * KP);
#pragma unroll
                    for (int k4 = 0; k4 < KP / 4; ++k4) { const float4 t4 = sp[k4]; v[k4 * 4] = t4.x; v[k4 * 4 + 1] = t4.y; v[k4 * 4 + 2] = t4.z; v[k4 * 4 + 3] = t4.w; }
                    const uint32_t vox = (uint32_t)(((n * LD + sd) * LH + sh) * LW + sw);
                    if (EPI == 0) {
                        if (p.res) {
                            float rr[KP];
                            unpack16<T>(*(const uint4*)((const T*)p.res + (size_t)(vox * (uint32_t)p.ldr + (uint32_t)col0)), rr);
#pragma unroll
                            for (int k = 0; k < KP; ++k) v[k] += rr[k];
                        }
#pragma unroll
                        for (int k = 0; k < KP; ++k) { v[k] = Elem<T>::rnd(v[k]); s1[k] += v[k]; s2[k] += v[k] * v[k]; }
                    } else {
                        float xx[KP];
                        unpack16<T>(xq[j], xx);
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            const float xn = (xx[k] - emu[k]) * ers[k];
                            v[k] = Elem<T>::rnd(xn > 0.f ? v[k] : 0.f);
                            s1[k] += v[k]; s2[k] += v[k] * xn;
                        }
                    }
                    *(uint4*)((T*)p.out + (size_t)(vox * (uint32_t)p.ldo + (uint32_t)col0)) = pack16<T>(v);
                }
            }
            __syncthreads();
        }
    }
    if (p.part) {
        float* red = (float*)smem;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            red[(pr0 * BN + cg * KP + k) * 2] = s1[k];
            red[(pr0 * BN + cg * KP + k) * 2 + 1] = s2[k];
        }
        __syncthreads();
        for (int cl = tid; cl < BN; cl += 256) {
            float a = 0.f, b = 0.f;
            for (int m = 0; m < RPT; ++m) { a += red[(m * BN + cl) * 2]; b += red[(m * BN + cl) * 2 + 1]; }
            const int col = colgrp * BN + cl;
            if (col < p.Cout) {
                const size_t row = (size_t)n * gridDim.x + tile_id;
                float* pp = p.part + (row * p.Cout + col) * 2;
                pp[0] = a; pp[1] = b;
            }
        }
    }
}

template <typename T, int MODE>
int launch_s2(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st) {
    typedef Geo<MODE> G;
    const int tiles = ((p.D + G::TD - 1) / G::TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    if (p.bn != 64 || (p.ntiles % 2)) return RS_ERR_ARG;
    constexpr int SCR = 4 * 32 * 68 * 4;                                        // epilogue scratch (4 fragments x 64 columns + pad)
    const size_t smem = (size_t)(SCR > G::HALO_BYTES ? SCR : G::HALO_BYTES) + (size_t)(p.a.C + p.b.C) * 2 * sizeof(float);
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    auto k = igemm_s2_kernel<T, MODE, 1>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(tiles, p.ntiles / 2, p.N), dim3(256), smem, st, p, FD, FH, FW);
    return rs_check_launch();
}

}  // namespace

// p.D/H/W = the HALF-resolution grid (ceil(F / 2) per axis), FD/FH/FW the full-resolution one; mode 1 forward, 2 data gradient; bn 64
int rs_launch_igemm_s2(const IgemmParams& p, int dtype, int mode, int FD, int FH, int FW, hipStream_t st) {
    if (mode != 1 && mode != 2) return RS_ERR_ARG;
    if (dtype == RS_F32) return mode == 1 ? launch_s2<float, 1>(p, FD, FH, FW, st) : launch_s2<float, 2>(p, FD, FH, FW, st);
    if (dtype == RS_BF16) return mode == 1 ? launch_s2<bf16_t, 1>(p, FD, FH, FW, st) : launch_s2<bf16_t, 2>(p, FD, FH, FW, st);
    return RS_ERR_ARG;
}
