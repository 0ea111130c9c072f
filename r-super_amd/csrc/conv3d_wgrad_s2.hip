// Weight gradient of the STRIDED 3x3x3 convolution (stride 2, padding 1) on MFMA (gfx950), channels-last activations.
//
//   dW[co][ci][t] = sum_o dY[o][co] * x_hat[2o + t - 1][ci]        (o on the half-resolution grid, x_hat on the full one)
//
// Replaces the autograd weight gradient of BasicBlock(stride=2)'s conv1 / shortcut conv
// (rsuper_train/model/dim3/conv_layers.py:29-38, :60-94 under down_block(pool=False), unet_utils.py:18-33).
// Rounds 1-3 evaluated it with the stride-1 kernel on a zero-stuffed full-resolution dY (8x the MFMA work plus the stuffing
// pass).  Here a block stages the FULL-resolution halo of a half-resolution dY tile once: (2 TD2 + 1) x (2 TH2 + 1) x 33 voxels
// of x_hat for TD2 x TH2 x 16 voxels of dY.  The w axis of the halo is stored de-interleaved (even positions first, then the
// odd ones), so that the 16 voxels 2q + kw (q = 0..15) of any tap are 16 CONSECUTIVE LDS rows and the transposed fragment read
// (ds_read_b64_tr_b16) of the stride-1 kernel works unchanged at its conflict-free pitch.  Every x voxel is read once
// (x halo factor 1.45), all 27 taps are dense MFMAs on the half grid: the kernel is bound by staging the activations, not
// by the matrix pipes.  Partial dW slabs + the deterministic slab reduction are those of conv3d_wgrad.hip.
#include "common.hpp"
#include "kernels.hpp"
#include "wgrad_frag.hpp"

namespace {

constexpr int TW2 = 16, HW2 = 2 * TW2 + 1;                       // dY tile width, halo width (33 = 17 even + 16 odd positions)

#ifndef S2W_NW
#define S2W_NW 8
#endif
#ifndef S2W_SKIP
#define S2W_SKIP 0                                               // profiling switches (wrong results): 1 no MFMA phase, 2 no norm + ReLU, 4 no LDS commit, 8 no global loads of x, 16 no slab write, 32 no tiles
#endif
template <typename T> struct S2W;
template <> struct S2W<bf16_t> { static constexpr int XP = 64, MT = 2, NW = S2W_NW, TD2 = 2, TH2 = 4, YP = 192; };   // LDS 95.0 + 24.6 KB
template <> struct S2W<float> { static constexpr int XP = 144, MT = 1, NW = 4, TD2 = 1, TH2 = 4, YP = 144; };   // LDS 128.3 + 9.2 KB

// first LDS row (within one (d, h) line of the halo) of the 16 voxels 2q + kw: kw = 0 -> even 0.., 1 -> odd 0.., 2 -> even 1..
__device__ __host__ constexpr int wstart(int kw) { return kw == 1 ? TW2 + 1 : (kw >> 1); }

template <typename T>
__global__ __launch_bounds__(64 * S2W<T>::NW, 1) void wgrad_s2_kernel(WgradParams p, int OD, int OH, int OW) {
    using G = S2W<T>;
    constexpr int NW = G::NW, MT = G::MT, TD2 = G::TD2, TH2 = G::TH2, XP = G::XP, YP = G::YP;
    constexpr int NT = 64 * NW;
    constexpr int KP = Elem<T>::KP;
    constexpr int HD2 = 2 * TD2 + 1, HH2 = 2 * TH2 + 1;
    constexpr int XROWS = HD2 * HH2 * HW2, YROWS = TD2 * TH2 * TW2;
    constexpr int XV = 32 / KP, YV = MT * 32 / KP;
    constexpr int WT = NW / MT, TPW = (27 + WT - 1) / WT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xh = smem;
    char* yt = smem + XROWS * XP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MT, wt = wave / MT;
    const ConvSrc& xs = p.xa;
    const int c0 = blockIdx.x * 32;
    const int Mtot = p.ya.C + p.yb.C;
    const int m0 = blockIdx.y * MT * 32;
    const bool norm = xs.mr != nullptr;
    const int tiles_w = (OW + TW2 - 1) / TW2, tiles_h = (OH + TH2 - 1) / TH2, tiles_d = (OD + TD2 - 1) / TD2;
    const int tiles = (S2W_SKIP & 32) ? 0 : tiles_w * tiles_h * tiles_d * p.N;

    f32x16_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int xoff[TPW];                                               // LDS offset of this wave's taps (wave-uniform); taps past 26 are clamped, never stored
    {
        const int wts = __builtin_amdgcn_readfirstlane(wt);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            int tl = wts + i * WT;
            if (tl > 26) tl = 26;
            const int kd = tl / 9, kh = (tl % 9) / 3, kw = tl % 3;
            xoff[i] = ((kd * HH2 + kh) * HW2 + wstart(kw)) * XP;
        }
    }

    // ---- staging: 16-byte vectors; the row decomposition of a thread's vectors never changes
    constexpr int NXV = (XROWS * XV + NT - 1) / NT, NYV = (YROWS * YV + NT - 1) / NT;
    constexpr int XRS = NT / XV, YRS = NT / YV;
    static_assert(NT % XV == 0 && NT % YV == 0, "staging rows per vector index must be whole");
    uint4 px[NXV], py[NYV];
    uint32_t xmask = 0, ymask = 0;
    const int xs_slot = tid % XV, xs_row = tid / XV, ys_slot = tid % YV, ys_row = tid / YV;
    const bool x_cok = c0 + xs_slot * KP < xs.C;
    const int ym = m0 + ys_slot * KP;
    const T* ysrc = nullptr; int yld = 0;
    if (ym < Mtot) {
        if (ym < p.ya.C) { ysrc = (const T*)p.ya.x + ym; yld = p.ya.ld; }
        else { ysrc = (const T*)p.yb.x + (ym - p.ya.C); yld = p.yb.ld; }
    }
    int xpos[NXV];                                               // hd | hh << 3 | rw << 7 | LDS row << 13 (one register per vector), or -1
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
        const int r = xs_row + i * XRS;
        const int hd = r / (HH2 * HW2), rem = r - hd * (HH2 * HW2);
        const int hh = rem / HW2, rw = rem - hh * HW2;
        const int lrow = (hd * HH2 + hh) * HW2 + ((rw & 1) ? TW2 + 1 + (rw >> 1) : (rw >> 1));
        xpos[i] = r < XROWS ? (hd | (hh << 3) | (rw << 7) | (lrow << 13)) : -1;
    }
    char* x_lds = xh + xs_slot * 16;
    char* y_lds = yt + ys_row * YP + ys_slot * 16;
    float sc_[KP], nb_[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { sc_[j] = 1.f; nb_[j] = 0.f; }
    const uint32_t xrowb = (uint32_t)xs.ld * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, (uint32_t)(p.N * p.D * p.H * p.W) * xrowb, 0x00020000);
    const uint32_t xcol = (uint32_t)(c0 + xs_slot * KP) * (uint32_t)sizeof(T);

    auto issue = [&](int tile) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        const int td = t % tiles_d; t /= tiles_d;
        const int n = t, d0 = td * TD2, h0 = th * TH2, w0 = tw * TW2;
        const int dlo = 2 * d0 - 1, hlo = 2 * h0 - 1, wlo = 2 * w0 - 1;
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int d = dlo + (xpos[i] & 7), h = hlo + ((xpos[i] >> 3) & 15), w = wlo + ((xpos[i] >> 7) & 63);
            const bool ok = xpos[i] >= 0 && x_cok && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const uint32_t off = ok ? (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w) * xrowb + xcol : 0xFFFFFFFFu;   // out of range -> zeros
            if (!(S2W_SKIP & 8)) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
                px[i] = make_uint4(q[0], q[1], q[2], q[3]);
            } else px[i] = make_uint4(off, off, off, off);
            xmask = (xmask & ~(1u << i)) | (ok ? (1u << i) : 0u);
            __builtin_amdgcn_sched_barrier(0);                   // one address at a time (hoisting all NXV offsets overflows the register file)
        }
#pragma unroll
        for (int i = 0; i < NYV; ++i) {
            const int r = ys_row + i * YRS;
            const int dd = r / (TH2 * TW2), hh = (r / TW2) % TH2, ww = r % TW2;
            const bool ok = r < YROWS && ysrc != nullptr && d0 + dd < OD && h0 + hh < OH && w0 + ww < OW;
            const T* src = ok ? ysrc + (size_t)(uint32_t)(((n * OD + d0 + dd) * OH + h0 + hh) * OW + w0 + ww) * (uint32_t)yld : (const T*)p.ya.x;
            py[i] = *(const uint4*)src;
            ymask = (ymask & ~(1u << i)) | (ok ? (1u << i) : 0u);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            uint4 q = px[i];
            if (!(S2W_SKIP & 2) && norm && ((xmask >> i) & 1u)) q = norm_relu16<T>(q, sc_, nb_);
            if ((!(S2W_SKIP & 4) || q.x == 0x12345u) && xpos[i] >= 0) *(uint4*)(x_lds + (xpos[i] >> 13) * XP) = q;
        }
#pragma unroll
        for (int i = 0; i < NYV; ++i)
            if (ys_row + i * YRS < YROWS) *(uint4*)(y_lds + i * (YRS * YP)) = ((ymask >> i) & 1u) ? py[i] : make_uint4(0, 0, 0, 0);
    };

    int cur_n = -1;
    if ((int)blockIdx.z < tiles) issue(blockIdx.z);
    for (int tile = blockIdx.z; tile < tiles; tile += p.splits) {
        const int n = tile / (tiles_w * tiles_h * tiles_d);
        __syncthreads();                                         // previous tile consumed
#pragma unroll
        for (int i = 0; i < NXV; ++i) asm volatile("" : "+v"(xpos[i]));   // opaque per tile: keeps the fields packed (the compiler would hoist 4 unpacked registers per vector)
        if (norm && n != cur_n) {
            cur_n = n;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int c = c0 + xs_slot * KP + j;
                const float mu = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2] : 0.f, rs = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2 + 1] : 1.f;
                sc_[j] = rs; nb_[j] = -mu * rs;
            }
        }
        commit();
        __syncthreads();
        if (tile + p.splits < tiles) issue(tile + p.splits);     // next tile's loads fly under this tile's MFMAs
        if constexpr (S2W_SKIP & 1) {
        } else if constexpr (sizeof(T) == 2) {
            const char* ya_base = yt + wm * 64 + frag_lane_off<1>(YP, lane);
            const char* xl = xh + frag_lane_off<1>(XP, lane);
            auto fetch_a = [&](int row) { return frag_bf16<1>(ya_base + (row * TW2) * YP, YP); };
            auto fetch_b = [&](int row, int i) {
                const int dd = row / TH2, hh = row % TH2;
                return frag_bf16<1>(xl + xoff[i] + ((2 * dd * HH2 + 2 * hh) * HW2) * XP, XP);
            };
            constexpr int NU = TD2 * TH2 * TPW, BD = 3, BR = BD + 1;
            uint4 aq[2], bq[BR];
            aq[0] = fetch_a(0);
#pragma unroll
            for (int u = 0; u < BD; ++u) bq[u] = fetch_b(u / TPW, u % TPW);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int row = u / TPW, i = u % TPW;
                if (u + BD < NU) bq[(u + BD) % BR] = fetch_b((u + BD) / TPW, (u + BD) % TPW);
                if (i == 0 && row + 1 < TD2 * TH2) aq[(row + 1) & 1] = fetch_a(row + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma32<bf16_t>(acc[i], aq[row & 1], bq[u % BR]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int row = 0; row < TD2 * TH2; ++row) {
                const int dd = row / TH2, hh = row % TH2;
                const char* ybase = yt + (row * TW2) * YP + wm * 32 * (int)sizeof(T);
                const char* xbase = xh + ((2 * dd * HH2 + 2 * hh) * HW2) * XP + (lane & 31) * 4;
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {                // 8 MFMAs of K = 2 voxels
                    const int wv = k2 * 2 + (lane >> 5);
                    const float a = *(const float*)(ybase + wv * YP + (lane & 31) * 4);
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const float b = *(const float*)(xbase + wv * XP + xoff[i]);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- partial dW slab of this split: ws[split][tap][m][cin]
    if ((S2W_SKIP & 16) && acc[0][0] != 12345.f) return;
    const int ci = c0 + (lane & 31);
    float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * xs.C;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tap = wt + i * WT;
        if (tap > 26) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + cd_row32(r, lane);
            if (m < Mtot && ci < xs.C) slab[((size_t)tap * Mtot + m) * xs.C + ci] = acc[i][r];
        }
    }
}

template <typename T>
int launch_s2w(const WgradParams& p, hipStream_t st) {
    using G = S2W<T>;
    const int OD = (p.D + 1) / 2, OH = (p.H + 1) / 2, OW = (p.W + 1) / 2;
    const size_t smem = (size_t)(2 * G::TD2 + 1) * (2 * G::TH2 + 1) * HW2 * G::XP + (size_t)G::TD2 * G::TH2 * TW2 * G::YP;
    const int Mtot = p.ya.C + p.yb.C;
    dim3 grid((p.xa.C + 31) / 32, (Mtot + G::MT * 32 - 1) / (G::MT * 32), p.splits), block(64 * G::NW);
    auto k = wgrad_s2_kernel<T>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p, OD, OH, OW);
    return rs_launch_wgrad_reduce(p, st);
}

template <typename T> int tiles_total(int N, int D, int H, int W) {
    using G = S2W<T>;
    const int OD = (D + 1) / 2, OH = (H + 1) / 2, OW = (W + 1) / 2;
    return N * ((OD + G::TD2 - 1) / G::TD2) * ((OH + G::TH2 - 1) / G::TH2) * ((OW + TW2 - 1) / TW2);
}

}  // namespace

// one block per CU (LDS); a split = the tiles tile, tile + splits, ...
int rs_wgrad_s2_splits(int dtype, int Ca, int Mtot, int N, int D, int H, int W) {
    const int tiles = dtype == RS_F32 ? tiles_total<float>(N, D, H, W) : tiles_total<bf16_t>(N, D, H, W);
    const int mt = dtype == RS_F32 ? S2W<float>::MT : S2W<bf16_t>::MT;
    const int blocks = ((Ca + 31) / 32) * ((Mtot + mt * 32 - 1) / (mt * 32));
    int s = 256 / (blocks > 0 ? blocks : 1);
    if (s > tiles) s = tiles;
    return s < 1 ? 1 : s;
}

int rs_launch_wgrad_s2(const WgradParams& p, int dtype, hipStream_t st) {
    if (dtype == RS_F32) return launch_s2w<float>(p, st);
    if (dtype == RS_BF16) return launch_s2w<bf16_t>(p, st);
    return RS_ERR_ARG;
}
