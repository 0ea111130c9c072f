// Weight gradient of the 3x3x3 convolution on MFMA (gfx950), channels-last activations.
//
//   dW[co][ci][tap] = sum_v dY[v][co] * x_hat[v + off(tap)][ci]          (GEMM: M = Cout, N = 27*Cin, K = voxels)
//
// Replaces the autograd weight-gradient of every nn.Conv3d(k=3) on the hot path
// (rsuper_train/model/dim3/conv_layers.py:29-38 under loss.backward(), train_ddp.py:349).
// x_hat = relu((x - mean) * rstd) is recomputed from x while the halo tile is staged (never stored).
// The reduction axis (voxels) is the slow axis of both operands in NDHWC, so MFMA fragments need a
// transposed read: bf16 uses ds_read_b64_tr_b16 (TR=1) or eight 16-bit LDS reads (TR=0, reference path);
// f32 MFMA (32x32x2) holds one k per lane and needs no transpose.
//
// Block = 4 waves, output tile = MT*32 output channels x NTAPS taps x 32 input channels, looping over the
// spatial tiles of its split; each split writes its partial dW slab with coalesced plain stores and a small
// reduce kernel sums the slabs into the f32 dW (deterministic; f32 atomics measured ~14 G/s were the bottleneck).
#include "common.hpp"
#include "kernels.hpp"
#include "wgrad_frag.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
#ifndef WG_BD_
#define WG_BD_ 3
#endif
constexpr int WG_BD = WG_BD_;                                         // B-fragment prefetch distance (MFMA units), classic kernel (>= 2 waves / SIMD)
#ifndef WG_LS
#define WG_LS 2                                                  // MFMA units between two staging loads of the next tile: 2 (round 3) -- with 5 the last loads
#endif                                                           // were issued late in the phase and their latency was exposed behind it (same box: up3.0 301 -> 271-288 us,
                                                                 // 128 -> 128 @24^3 56 -> 52-53, 64 -> 64 @48^3 76 -> 73; 1 measures the same as 2)
constexpr int HH = TH + 2, HW = TW + 2;

#ifdef RS_WG_PROF
__device__ unsigned long long g_wg_prof[16 * 8];                 // block 0: [wave][wait+commit, barrier, issue, mfma, tiles]
#endif

template <typename T> struct WG;
// bf16 rows are only read with ds_read_b64_tr_b16 ([4 rows] x [32 B] per 16-lane group, two groups side by side): the
// four rows must land on disjoint 16-dword bank ranges -> pitch = 16 (mod 64) dwords, or 48 for the 128-byte dY rows.
template <> struct WG<bf16_t> { static constexpr int XP = 64; };     // x_hat row pitch: 32 ch * 2 B, unpadded
template <> struct WG<float> { static constexpr int XP = 144; };     // 32 ch * 4 B + 16

// NW waves per block: wave -> (wm = wave % MT, wt = wave / MT); a wave owns the taps wt, wt + WT, ... (WT = NW / MT).
// NW = 4 everywhere: 6 (9 taps) / 9 (27 taps) waves with three taps each measured 1.2-2.4x slower (register cap, spills).
template <typename T, int MT, int NTAPS, int TR, int NW>
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 2 && MT == 2) ? 2 : 1) void wgrad_kernel(WgradParams p) {   // HIP: 2nd = min waves per SIMD
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = Elem<T>::KP;
    constexpr int XP = WG<T>::XP;
    constexpr int YP = sizeof(T) == 2 ? (MT == 2 ? 192 : 64) : MT * 32 * (int)sizeof(T) + 16;
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;               // halo depth rows
    constexpr int XROWS = HDN * HH * HW;
    constexpr int XV = 32 / KP;                                  // 16-B vectors per x row
    constexpr int YV = MT * 32 / KP;
    constexpr int WT = NW / MT;                                  // tap stride between a wave's taps
    constexpr int TPW = (NTAPS + WT - 1) / WT;                   // taps per wave (max)
    char* xh = smem;
    char* yt = smem + XROWS * XP;
    float* mr_lds = (float*)(yt + 256 * YP);                     // [32][2] for this chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MT, wt = wave / MT;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d * p.N;
    const WgBlockMap bm = wg_block_map(tiles, p.splits);         // (chunk, row group, split) of this block and the split's tiles (wgrad_frag.hpp)
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = bm.bx >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? bm.bx - nchA : bm.bx) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    const int mg = bm.by % mgroups;
    const int kdg = NTAPS == 27 ? 0 : bm.by / mgroups;           // kd handled by this block (9-tap config)
    const int m0 = mg * MT * 32;
    const bool norm = xs.mr != nullptr;

    f32x16_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // LDS offset of each of this wave's taps (wave-uniform -> SGPRs); taps past NTAPS are clamped (their
    // accumulator slot is never stored).
    int xoff[TPW];
    {
        const int wts = __builtin_amdgcn_readfirstlane(wt);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            int tl = wts + i * WT;
            if (tl >= NTAPS) tl = NTAPS - 1;
            const int kd = NTAPS == 27 ? tl / 9 : 0, kh = (tl % 9) / 3, kw = tl % 3;
            xoff[i] = ((kd * HH + kh) * HW + kw) * XP;
        }
    }
    // bf16: per-tap fragment bases with the lane part folded in (one VGPR per tap instead of address math per fetch)
    const char* xb_base[TPW];
    const char* ya_base = yt + wm * 64 + frag_lane_off<TR>(YP, lane);
#pragma unroll
    for (int i = 0; i < TPW; ++i) xb_base[i] = xh + xoff[i] + frag_lane_off<TR>(XP, lane);

    // Issue-early / write-late staging: the 16-byte global loads of the NEXT tile are started right before the MFMA
    // phase of the current one and land in LDS (after norm+ReLU) once the barrier says the tile has been consumed.
    constexpr int NXV = (XROWS * XV + NT - 1) / NT, NYV = (256 * YV + NT - 1) / NT;
    constexpr int XRS = NT / XV, YRS = NT / YV;                  // rows advanced per vector index
    static_assert(NT % XV == 0 && NT % YV == 0, "staging rows per vector index must be whole");
    uint4 px[NXV], py[NYV];
    uint32_t xmask = 0, ymask = 0;                               // bit i: px[i] / py[i] is inside the volume (x: gets norm+ReLU; y: else zero)
    const int xs_slot = tid % XV, xs_row = tid / XV;             // per-thread constants: 16-byte slot and first row
    const int ys_slot = tid % YV, ys_row = tid / YV;
    const bool x_cok = c0 + xs_slot * KP < xs.C;
    const int ym = m0 + ys_slot * KP;                            // first dY channel of this thread's vectors
    const T* ysrc = nullptr; int yld = 0;
    if (ym < Mtot) {
        if (ym < p.ya.C) { ysrc = (const T*)p.ya.x + ym; yld = p.ya.ld; }
        else { ysrc = (const T*)p.yb.x + (ym - p.ya.C); yld = p.yb.ld; }
    }
    const T* xsrc = (const T*)xs.x + c0 + xs_slot * KP;
    char* x_lds = xh + xs_row * XP + xs_slot * 16;
    char* y_lds = yt + ys_row * YP + ys_slot * 16;
    float sc_[KP], nb_[KP];                                      // x_hat = max(x * rstd - mean * rstd, 0) for this thread's channels
#pragma unroll
    for (int j = 0; j < KP; ++j) { sc_[j] = 1.f; nb_[j] = 0.f; }

    auto tile_coords = [&](int tile, int& n, int& d0, int& h0, int& w0) {
        const int per = tiles_w * tiles_h * tiles_d;
        n = tile / per;
        int tw, th, td;
        rs_tile_coords(tile - n * per, tiles_w, tiles_h, tiles_d, tw, th, td);
        d0 = td * TD; h0 = th * TH; w0 = tw * TW;
    };
    // Per-thread constants of the staging vectors: voxel delta of every vector relative to the tile's halo / tile
    // origin (the (hd, hh, hw) decomposition of its row never changes), so an interior tile costs one add per vector.
    int xdelta[NXV], ydelta[NYV];
    uint32_t xpm[NXV], ypm[NYV];                                 // one-hot row positions; bit 31: the vector never holds data
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
        const int r = xs_row + i * XRS;
        const int hd = r / (HH * HW);
        const int rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        xdelta[i] = (hd * p.H + hh) * p.W + hw;
        xpm[i] = (r < XROWS && x_cok) ? ((1u << hd) | (1u << (6 + hh)) | (1u << (12 + hw))) : (1u << 31);
    }
#pragma unroll
    for (int i = 0; i < NYV; ++i) {
        const int r = ys_row + i * YRS;
        ydelta[i] = ((r >> 6) * p.H + ((r >> 4) & 3)) * p.W + (r & 15);
        ypm[i] = (r < 256 && ysrc != nullptr) ? ((1u << (r >> 6)) | (1u << (4 + ((r >> 4) & 3))) | (1u << (8 + (r & 15)))) : (1u << 31);
    }
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    const uint32_t xrowb = (uint32_t)xs.ld * (uint32_t)sizeof(T);
    // wave-uniform resource (a per-lane base would make every buffer load a waterfall loop); the thread's channel slot
    // goes into the offset
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, nvox_total * xrowb, 0x00020000);
    const uint32_t xcol = (uint32_t)(c0 + xs_slot * KP) * (uint32_t)sizeof(T);
    auto ld16 = [&](uint32_t off) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);     // out-of-range offsets return zeros
        return make_uint4(q[0], q[1], q[2], q[3]);
    };
    // Staging loads of one tile, one 16-byte vector per call so that the MFMA loop can interleave them (a burst of all
    // NXV + NYV loads stalls at the CU's ~10 B/clk request rate for ~7.7k cycles per tile -- longer than the MFMA phase --
    // with the matrix pipes idle; tools: RS_WG_PROF).  `IssueTile` holds the wave-uniform part.
    // Bounds are bit tests (no branches -- a branch per load makes the compiler wait for every load at once): xpm / ypm are the
    // per-thread one-hot positions of a vector's row inside the halo / tile, the tile contributes the mask of INVALID positions.
    struct IssueTile { int base, ybase; uint32_t xbad, ybad; };
    auto prepare = [&](int tile) {
        IssueTile t;
        int n, d0, h0, w0;
        tile_coords(tile, n, d0, h0, w0);
        const int dlo = d0 + (NTAPS == 27 ? -1 : kdg - 1);
        auto range = [](int o, int len, int nh) {                // bits hd in [0, nh) with 0 <= o + hd < len
            const int lo = o >= 0 ? 0 : -o;
            int hi_ = len - 1 - o; if (hi_ > nh - 1) hi_ = nh - 1;
            return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
        };
        t.xbad = ~(range(dlo, p.D, HDN) | (range(h0 - 1, p.H, HH) << 6) | (range(w0 - 1, p.W, HW) << 12));
        t.ybad = ~(range(d0, p.D, TD) | (range(h0, p.H, TH) << 4) | (range(w0, p.W, TW) << 8));
        t.base = ((n * p.D + dlo) * p.H + (h0 - 1)) * p.W + (w0 - 1);
        t.ybase = ((n * p.D + d0) * p.H + h0) * p.W + w0;
        return t;
    };
    auto issue_x = [&](const IssueTile& t, int i) {
        const bool ok = (xpm[i] & t.xbad) == 0u;
        px[i] = ld16(ok ? (uint32_t)(t.base + xdelta[i]) * xrowb + xcol : 0xFFFFFFFFu);
        xmask = (xmask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    auto issue_y = [&](const IssueTile& t, int i) {
        const bool ok = (ypm[i] & t.ybad) == 0u;
        const T* src = ok ? ysrc + (size_t)(uint32_t)(t.ybase + ydelta[i]) * (uint32_t)yld : (const T*)p.ya.x;   // address select, no branch
        py[i] = *(const uint4*)src;                              // masked at commit time (a select here would wait for the load)
        ymask = (ymask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    auto issue = [&](int tile) {
        const IssueTile t = prepare(tile);
#pragma unroll
        for (int i = 0; i < NXV; ++i) issue_x(t, i);
#pragma unroll
        for (int i = 0; i < NYV; ++i) issue_y(t, i);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            uint4 q = px[i];
            if (norm && ((xmask >> i) & 1u)) {
                q = norm_relu16<T>(q, sc_, nb_);
            }
            if (xs_row + i * XRS < XROWS) *(uint4*)(x_lds + i * (XRS * XP)) = q;
        }
#pragma unroll
        for (int i = 0; i < NYV; ++i)
            if (ys_row + i * YRS < 256) *(uint4*)(y_lds + i * (YRS * YP)) = ((ymask >> i) & 1u) ? py[i] : make_uint4(0, 0, 0, 0);
    };

    int cur_n = -1;
    // XCD-aware tile order (wg_block_map): the splits on one XCD own a contiguous range of tiles and sweep it together
    const int tile0 = bm.tile0, tile_end = bm.tile_end, tstride = bm.tstride;
    if (tile0 < tile_end) issue(tile0);
#ifdef RS_WG_PROF
    unsigned long long pf[5] = {0, 0, 0, 0, 0};
#endif
    for (int tile = tile0; tile < tile_end; tile += tstride) {
        const int n = tile / (tiles_w * tiles_h * tiles_d);
#ifdef RS_WG_PROF
        const unsigned long long q0 = __builtin_readcyclecounter();
#endif
        __syncthreads();                                         // previous tile consumed
#ifdef RS_WG_PROF
        const unsigned long long q1 = __builtin_readcyclecounter();
#endif
        if (norm && n != cur_n) {                                // per-sample statistics of this thread's KP channels
            cur_n = n;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int c = c0 + xs_slot * KP + j;
                const float mu = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2] : 0.f, rs = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2 + 1] : 1.f;
                sc_[j] = rs; nb_[j] = -mu * rs;
            }
        }
        commit();
#ifdef RS_WG_PROF
        const unsigned long long q2 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef RS_WG_PROF
        const unsigned long long q3 = __builtin_readcyclecounter();
#endif
        const bool has_next = tile + tstride < tile_end;
        if (sizeof(T) != 2 && has_next) issue(tile + tstride);   // f32 parity mode: burst; bf16: interleaved with the MFMAs below
        const IssueTile nt = prepare(has_next ? tile + tstride : tile);    // last tile: harmless re-load of itself
#ifdef RS_WG_PROF
        const unsigned long long q4 = __builtin_readcyclecounter();
        pf[0] += q1 - q0; pf[1] += q2 - q1; pf[2] += q3 - q2; pf[3] += q4 - q3; pf[4] += 1;
#endif
        // ---- MFMA over the 16 (d,h) rows of the tile; k = 16 voxels along w.  Software pipelined with static
        //      indices: the operand fragments of unit (row, tap i)+1 are fetched from LDS while unit (row, i) issues.
        if constexpr (sizeof(T) == 2) {
            auto fetch_a = [&](int row) { return frag_bf16<TR>(ya_base + (row * TW) * YP, YP); };         // static row offsets fold into
            auto fetch_b = [&](int row, int i) {                                                          // the ds_read offset field
                const int dd = row / TH, hh = row % TH;
                return frag_bf16<TR>(xb_base[i] + ((dd * HH + hh) * HW) * XP, XP);
            };
            // few waves per SIMD: LDS latency (~130+ cycles) must be covered by prefetch distance, one MFMA is only 32 cycles
            constexpr int NU = TD * TH * TPW, BD = WG_BD, BR = WG_BD + 1;
            uint4 aq[2], bq[BR];
            aq[0] = fetch_a(0);
#pragma unroll
            for (int u = 0; u < BD; ++u) bq[u] = fetch_b(u / TPW, u % TPW);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int row = u / TPW, i = u % TPW;
#ifndef WG_PRIO
#define WG_PRIO 0
#endif
                if (WG_PRIO == 1) {                              // progress-based priority (conv3d_igemm_kd.hip): the wave that is behind in the tile outranks its partner
                    if (u == 0) __builtin_amdgcn_s_setprio(3);
                    if (u == NU / 4) __builtin_amdgcn_s_setprio(2);
                    if (u == NU / 2) __builtin_amdgcn_s_setprio(1);
                    if (u == (3 * NU) / 4) __builtin_amdgcn_s_setprio(0);
                }
                if (u + BD < NU) bq[(u + BD) % BR] = fetch_b((u + BD) / TPW, (u + BD) % TPW);
                if (i == 0 && row + 1 < TD * TH) aq[(row + 1) & 1] = fetch_a(row + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma32<bf16_t>(acc[i], aq[row & 1], bq[u % BR]);    // invalid taps multiply by a zeroed accumulator slot (never stored)
                // one staging load of the next tile every LS units, from the start of the phase (latency budget = the rest of it)
                constexpr int LS = WG_LS < (NU - 8) / (NXV + NYV) ? WG_LS : ((NU - 8) / (NXV + NYV) > 0 ? (NU - 8) / (NXV + NYV) : 1);
                if (u % LS == 0 && u / LS < NXV) issue_x(nt, u / LS);
                if (u % LS == 0 && u / LS >= NXV && u / LS < NXV + NYV) issue_y(nt, u / LS - NXV);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int row = 0; row < TD * TH; ++row) {
                const int dd = row / TH, hh = row % TH;
                const char* ybase = yt + (row * TW) * YP + wm * 32 * (int)sizeof(T);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {                // 8 MFMAs of K=2 voxels
                    const int wv = k2 * 2 + (lane >> 5);
                    const float a = *(const float*)(ybase + wv * YP + (lane & 31) * 4);
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const float b = *(const float*)(xh + ((dd * HH + hh) * HW + wv) * XP + xoff[i] + (lane & 31) * 4);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    }
                }
            }
        }
#ifdef RS_WG_PROF
        pf[3] += 0; g_wg_prof[127] = 0;
        { const unsigned long long q5 = __builtin_readcyclecounter(); if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_wg_prof[wave * 8 + 6] += q5 - q4; }
#endif
    }
#ifdef RS_WG_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
        for (int i = 0; i < 5; ++i) g_wg_prof[wave * 8 + i] = pf[i];
        g_wg_prof[wave * 8 + 7] = g_wg_prof[wave * 8 + 6]; g_wg_prof[wave * 8 + 6] = 0;
        g_wg_prof[wave * 8 + 5] = __builtin_readcyclecounter();
    }
#endif
    // ---- write this split's partial dW slab: ws[split][tap][m][cin]  (coalesced along cin)
    const int ci = c0 + (lane & 31);
    float* slab = p.ws + (size_t)bm.bz * 27 * Mtot * cin_total;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tl = wt + i * WT;
        if (tl >= NTAPS) continue;
        const int tap = NTAPS == 27 ? tl : kdg * 9 + tl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + cd_row32(r, lane);
            if (m < Mtot && ci < xs.C) slab[((size_t)tap * Mtot + m) * cin_total + cin_base + (lane & 31)] = acc[i][r];
        }
    }
}

// dW[m][cin][tap] = sum_s ws[s][tap][m][cin]; rows [0,Ya) -> dwa, [Ya, Ya+Yb) -> dwb.
// Block = 32 consecutive slab elements x SG split groups (SG = 32 for >= 128 splits, else 8): the loads of one element are spread
// over SG threads and every thread issues ALL its loads before the first add (round 2 walked the slabs in four dependent rounds of
// four loads: ~10 us per launch, pure latency -- the slabs were just written and sit in L2 / Infinity Cache); each row of 32
// threads reads 128 contiguous bytes per slab; the SG partials meet in LDS in a fixed order (deterministic).
template <int SG, int UN>
__global__ __launch_bounds__(32 * SG) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int Mtot, int Ya, int Cin,
                                                               float* dwa, float* dwb) {
    __shared__ float4 part[SG][32];
    const size_t E = (size_t)27 * Mtot * Cin;                    // multiple of 8 (Cin is)
    const int el = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const size_t e = ((size_t)blockIdx.x * 32 + el) * 4;         // four consecutive slab elements (same tap and row, cin .. cin + 3)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < E) {
        for (int s0 = sg; s0 < splits; s0 += SG * UN) {
            float4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int s = s0 + u * SG;
                v[u] = s < splits ? *(const float4*)(ws + (size_t)s * E + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    part[sg][el] = acc;
    __syncthreads();
    if (sg == 0 && e < E) {
        float4 v = part[0][el];
#pragma unroll
        for (int k = 1; k < SG; ++k) { const float4 t = part[k][el]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        const int c = (int)(e % Cin);
        const size_t r = e / Cin;
        const int m = (int)(r % Mtot), tap = (int)(r / Mtot);
        float* dst = (m < Ya ? dwa + ((size_t)m * Cin + c) * 27 : dwb + ((size_t)(m - Ya) * Cin + c) * 27) + tap;
        dst[0] = v.x; dst[27] = v.y; dst[54] = v.z; dst[81] = v.w;
    }
}

// Few slabs (large weight tensors at low resolution).  Block = one output row m x 64 input channels x 27 taps: the slab rows
// ws[s][tap][m][c0 .. c0+63] are read coalesced along c, summed over the splits in registers, transposed through LDS and
// written as ONE contiguous run dW[m][c0 .. c0+63][0 .. 26] (the thread-per-element version scattered 4-byte writes 108 B
// apart: the (Cout, Cin, 27) layout has the tap innermost).
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ ws, int splits, int Mtot, int Ya, int Cin,
                                                                float* dwa, float* dwb) {
    __shared__ float tile[64 * 27 + 64];
    const int cchunks = (Cin + 63) / 64;
    const int m = blockIdx.x / cchunks, c0 = (blockIdx.x % cchunks) * 64;
    const int nc = min(64, Cin - c0);
    const size_t E = (size_t)27 * Mtot * Cin;
    // all loads of a thread's seven (tap, c) items for one group of up to four slabs are issued before the first add
    constexpr int NI = (27 * 64 + 255) / 256;
    float a[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) a[q] = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 4) {
        float v[NI][4];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = threadIdx.x + q * 256;
            const int tap = i >> 6, c = i & 63;
            const bool ok = i < 27 * 64 && c < nc;
            const size_t e = ((size_t)(ok ? tap : 0) * Mtot + m) * Cin + c0 + (ok ? c : 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[q][u] = (ok && s0 + u < splits) ? ws[(size_t)(s0 + u) * E + e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) a[q] += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i < 27 * 64) tile[(i & 63) * 27 + (i >> 6)] = a[q];
    }
    __syncthreads();
    float* dst = m < Ya ? dwa + ((size_t)m * Cin + c0) * 27 : dwb + ((size_t)(m - Ya) * Cin + c0) * 27;
    for (int i = threadIdx.x; i < nc * 27; i += 256) dst[i] = tile[i];
}

// Both reductions above for a LIST of weight gradients in one launch: block -> (entry, block index inside the entry); entries with at least 16
// slabs take the element-parallel form (8 split groups), the others the row form.  Sums are formed in exactly the order of the single-entry
// kernels of the same form (bit-identical results for < 16 and for 16 .. 127 slabs; >= 128 slabs sum in 8 instead of 32 groups).
// stats_finalize_kernel (unet_misc.hip) for 256 threads, bit-identical: the 1024-thread kernel sums in 32 row groups (group g: rows g, g + 32, ... in trips of
// four) and adds the groups in order; here thread (cl, q) carries the four logical groups q, q + 8, q + 16, q + 24 through the same trips and the same final order.
__device__ __forceinline__ void reduce_batch_stats_role(const ReduceBatch::Stats& j, unsigned l, double (*red)[2]) {
    const int cgs = (j.C + 31) / 32;
    const int n = (int)(l / cgs), cg = (int)(l % cgs);
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int c = cg * 32 + cl;
    const int nblk = j.nblk, C = j.C;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g = q + 8 * u;
        double s0 = 0.0, s1 = 0.0;
        if (c < C) {
            const float* base = j.part + ((size_t)n * nblk * C + c) * 2;
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
        red[g * 32 + cl][0] = s0; red[g * 32 + cl][1] = s1;
    }
    __syncthreads();
    if (q == 0 && c < C) {
        double s0 = red[cl][0], s1 = red[cl][1];
        for (int k = 1; k < 32; ++k) { s0 += red[k * 32 + cl][0]; s1 += red[k * 32 + cl][1]; }
        float* o = j.split <= 0 ? j.out + ((size_t)n * C + c) * 2
                   : c < j.split ? j.out + ((size_t)n * j.split + c) * 2
                                 : j.out + (size_t)j.N * j.split * 2 + ((size_t)n * (C - j.split) + (c - j.split)) * 2;
        if (j.mode == 0) {
            const double mean = s0 / j.cnt;
            double var = s1 / j.cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            o[0] = (float)mean;
            o[1] = (float)(1.0 / sqrt(var + (double)j.eps));
        } else {
            o[0] = (float)(s0 / j.cnt); o[1] = (float)(s1 / j.cnt);
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceBatch b) {
    __shared__ float4 part[8][32];
    __shared__ float tile[64 * 27 + 64];
    if (blockIdx.x >= b.blocks) {                                // a statistics job (ReduceBatch::sj)
        __shared__ double sred[1024][2];
        int ji = 0;
        for (int i = 1; i < b.nstats; ++i) ji = blockIdx.x >= b.sj[i].blk_start ? i : ji;
        reduce_batch_stats_role(b.sj[ji], blockIdx.x - b.sj[ji].blk_start, sred);
        return;
    }
    int ei = 0;
    for (int i = 1; i < b.n; ++i) ei = blockIdx.x >= b.e[i].blk_start ? i : ei;      // entries are few: linear scan in scalar registers
    const ReduceBatch::Entry& en = b.e[ei];
    const unsigned blk = blockIdx.x - en.blk_start;
    const float* __restrict__ ws = en.ws;
    const int splits = en.splits, Mtot = en.Mtot, Ya = en.Ya, Cin = en.Cin;
    const size_t E = (size_t)27 * Mtot * Cin;
    if (splits >= 16) {
        constexpr int SG = 8, UN = 8;
        const int el = threadIdx.x & 31, sg = threadIdx.x >> 5;
        const size_t e = ((size_t)blk * 32 + el) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < E) {
            for (int s0 = sg; s0 < splits; s0 += SG * UN) {
                float4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int s = s0 + u * SG;
                    v[u] = s < splits ? *(const float4*)(ws + (size_t)s * E + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        part[sg][el] = acc;
        __syncthreads();
        if (sg == 0 && e < E) {
            float4 v = part[0][el];
#pragma unroll
            for (int k = 1; k < SG; ++k) { const float4 t = part[k][el]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            const int c = (int)(e % Cin);
            const size_t r = e / Cin;
            const int m = (int)(r % Mtot), tap = (int)(r / Mtot);
            float* dst = (m < Ya ? en.dwa + ((size_t)m * Cin + c) * 27 : en.dwb + ((size_t)(m - Ya) * Cin + c) * 27) + tap;
            dst[0] = v.x; dst[27] = v.y; dst[54] = v.z; dst[81] = v.w;
        }
        return;
    }
    const int cchunks = (Cin + 63) / 64;
    const int m = blk / cchunks, c0 = (blk % cchunks) * 64;
    const int nc = min(64, Cin - c0);
    constexpr int NI = (27 * 64 + 255) / 256;
    float a[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) a[q] = 0.f;
    for (int s0 = 0; s0 < splits; s0 += 4) {
        float v[NI][4];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = threadIdx.x + q * 256;
            const int tap = i >> 6, c = i & 63;
            const bool ok = i < 27 * 64 && c < nc;
            const size_t e = ((size_t)(ok ? tap : 0) * Mtot + m) * Cin + c0 + (ok ? c : 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[q][u] = (ok && s0 + u < splits) ? ws[(size_t)(s0 + u) * E + e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) a[q] += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i < 27 * 64) tile[(i & 63) * 27 + (i >> 6)] = a[q];
    }
    __syncthreads();
    float* dst = m < Ya ? en.dwa + ((size_t)m * Cin + c0) * 27 : en.dwb + ((size_t)(m - Ya) * Cin + c0) * 27;
    for (int i = threadIdx.x; i < nc * 27; i += 256) dst[i] = tile[i];
}

static bool g_skip_reduce = false;                               // set by rs_launch_wgrad for the duration of one launch call
static void launch_reduce_now(const WgradParams& p, hipStream_t st);
static void launch_reduce(const WgradParams& p, hipStream_t st) { if (!g_skip_reduce) launch_reduce_now(p, st); }
static void launch_reduce_now(const WgradParams& p, hipStream_t st) {
    const int Mtot = p.ya.C + p.yb.C, Cin = p.xa.C + p.xb.C;
    const size_t elems = (size_t)27 * Mtot * Cin;
    if (p.splits >= 128)
        hipLaunchKernelGGL((wgrad_reduce_kernel<32, 8>), dim3((unsigned)((elems + 127) / 128)), dim3(1024), 0, st, (const float*)p.ws, p.splits, Mtot, p.ya.C, Cin, p.dwa, p.dwb);
    else if (p.splits >= 16)
        hipLaunchKernelGGL((wgrad_reduce_kernel<8, 8>), dim3((unsigned)((elems + 127) / 128)), dim3(256), 0, st, (const float*)p.ws, p.splits, Mtot, p.ya.C, Cin, p.dwa, p.dwb);
    else
        hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3((unsigned)(Mtot * ((Cin + 63) / 64))), dim3(256), 0, st, (const float*)p.ws, p.splits, Mtot, p.ya.C, Cin, p.dwa, p.dwb);
}

template <typename T, int MT, int NTAPS, int TR, int NW>
int launch(const WgradParams& p, hipStream_t st) {
    constexpr int XP = WG<T>::XP;
    constexpr int YP = sizeof(T) == 2 ? (MT == 2 ? 192 : 64) : MT * 32 * (int)sizeof(T) + 16;
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;
    const size_t smem = (size_t)HDN * HH * HW * XP + 256 * YP + 64 * sizeof(float);
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    dim3 grid(nch, mgroups * (NTAPS == 27 ? 1 : 3), p.splits), block(64 * NW);
    auto k = wgrad_kernel<T, MT, NTAPS, TR, NW>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#ifdef RS_WG_PROF
    static unsigned long long t_prev = 0;
#endif
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
#ifdef RS_WG_PROF
    if (getenv("RSUPER_WG_PROF")) {
        unsigned long long h[128];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg_prof), sizeof(h));
        for (int w = 0; w < NW; w += 3) fprintf(stderr, "wg_prof MT%d taps%d NW%d wave %d: tiles %llu | per tile: barrier1 %.0f commit(+wait loads) %.0f barrier2 %.0f issue %.0f mfma %.0f\n", MT, NTAPS, NW, w, h[w * 8 + 4],
                                             (double)h[w * 8] / h[w * 8 + 4], (double)h[w * 8 + 1] / h[w * 8 + 4], (double)h[w * 8 + 2] / h[w * 8 + 4], (double)h[w * 8 + 3] / h[w * 8 + 4], (double)h[w * 8 + 7] / h[w * 8 + 4]);
        (void)t_prev;
    }
#endif
    launch_reduce(p, st);
    return rs_check_launch();
}

}  // namespace

// Configuration per launch (measured on MI355X, tools/bench_conv.py):
//   0: M <= 32           -> 32 rows x 27 taps, 8 waves (four taps per wave), one block per CU
//   1: M > 32, few tiles -> 64 rows x 9 taps (kd split over blocks), 4 waves, two blocks per CU
//   2: M > 32, >= 128 tiles (bf16) -> 64 rows x 27 taps, 8 waves, one block per CU: the x halo and the dY tile are staged
//      once for all 27 taps (2.4x less operand traffic than config 1), seven taps per A fragment
int rs_wgrad_config(int dtype, int Mtot, int tiles_total) {
    if (Mtot <= 32) return 0;
    return (dtype == RS_BF16 && tiles_total >= 128) ? 2 : 1;
}

// 64-row weight gradients with too few tiles for 64-row blocks of the second-generation kernel (the 64 -> 64 layers at 48^3: 864 tiles / 128 splits) run it
// as two 32-row blocks per chunk: half the splits, 13.5 tiles per block -- 64 -> 64 @48^3 72.5 -> 67.7 us (tools/bench_conv.py, same box).  RSUPER_WG2_MT1=0: off (A/B).
bool rs_wgrad2_mt1(int dtype, int Mtot, int tiles_total) {
    static const int on = getenv("RSUPER_WG2_MT1") ? atoi(getenv("RSUPER_WG2_MT1")) : 1;
    static const int maxm = getenv("RSUPER_WG2_MT1_MAXM") ? atoi(getenv("RSUPER_WG2_MT1_MAXM")) : 64;                // experiment knobs (tools/wg_mt1_exp.sh)
    static const int mint = getenv("RSUPER_WG2_MT1_MIN_TILES") ? atoi(getenv("RSUPER_WG2_MT1_MIN_TILES")) : 512;
    return on && dtype == RS_BF16 && Mtot >= 64 && Mtot <= maxm && (Mtot % 32) == 0 && tiles_total >= mint && tiles_total < 2048;
}

int rs_wgrad_splits(int dtype, int Mtot, int nch, int tiles_total) {
    const int cfg = rs_wgrad_config(dtype, Mtot, tiles_total);
    int gy = cfg == 0 ? 1 : (cfg == 1 ? 3 : 1) * ((Mtot + 63) / 64);
    if (rs_wgrad2_mt1(dtype, Mtot, tiles_total)) gy = Mtot / 32;  // 32-row blocks on the second-generation kernel (twice the tiles per block, half the slab bytes)
    const int target = cfg == 1 ? 512 : 256;                     // resident blocks on 256 CUs
    int s = target / (nch * gy > 0 ? nch * gy : 1);
    if (s > tiles_total) s = tiles_total;
    return s < 1 ? 1 : s;
}

int rs_launch_wgrad_reduce_batch(ReduceBatch& b, hipStream_t st) {
    if (b.n <= 0) return RS_OK;
    unsigned blocks = 0;
    for (int i = 0; i < b.n; ++i) {
        ReduceBatch::Entry& e = b.e[i];
        e.blk_start = blocks;
        const size_t elems = (size_t)27 * e.Mtot * e.Cin;
        blocks += e.splits >= 16 ? (unsigned)((elems + 127) / 128) : (unsigned)(e.Mtot * ((e.Cin + 63) / 64));
    }
    b.blocks = blocks;
    for (int i = 0; i < b.nstats; ++i) {
        b.sj[i].blk_start = blocks;
        blocks += (unsigned)(((b.sj[i].C + 31) / 32) * b.sj[i].N);
    }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, b);
    return rs_check_launch();
}

int rs_wgrad2_min_tiles(int t) {
    static int v = getenv("RSUPER_WGRAD2_MIN_TILES") ? atoi(getenv("RSUPER_WGRAD2_MIN_TILES")) : 12;
    if (t >= 0) v = t;
    return v;
}

int rs_launch_wgrad_reduce(const WgradParams& p, hipStream_t st) {
    launch_reduce_now(p, st);
    return rs_check_launch();
}

static int launch_wgrad_impl(const WgradParams& p, int dtype, int use_tr, hipStream_t st);
// reduce = false: only the partial-slab kernel (the caller launches rs_launch_wgrad_reduce itself, e.g. on another stream)
int rs_launch_wgrad(const WgradParams& p, int dtype, int use_tr, hipStream_t st, bool reduce) {
    g_skip_reduce = !reduce;
    const int rc = launch_wgrad_impl(p, dtype, use_tr, st);
    g_skip_reduce = false;
    return rc;
}

static int launch_wgrad_impl(const WgradParams& p, int dtype, int use_tr, hipStream_t st) {
    const int Mtot = p.ya.C + p.yb.C;
    const int tiles_total = p.N * ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const int cfg = rs_wgrad_config(dtype, Mtot, tiles_total);
    if (dtype == RS_F32) return cfg == 0 ? launch<float, 1, 27, 0, 4>(p, st) : launch<float, 2, 9, 0, 4>(p, st);
    {   // small volumes (12^3 / 6^3 levels): one depth slab of a sample whole in LDS (conv3d_wgrad_sv.hip); the caller sized `splits` for it
        const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
        const int sv = use_tr ? rs_wgrad_sv_splits(dtype, Mtot, p.ya.C, nch, p.N, p.D, p.H, p.W) : 0;
        if (sv > 0 && sv == p.splits) {
            const int rc = rs_launch_wgrad_sv(p, st);
            if (rc != RS_OK) return rc;
            launch_reduce(p, st);
            return rs_check_launch();
        }
    }
    // Second-generation kernel (conv3d_wgrad2.hip: operand re-use across taps, double-buffered tiles) where a block sweeps enough tiles to amortise its
    // heavier prologue (descriptor / constants tables, two tiles staged before the first MFMA): measured same-box against the kernel below, >= 12 tiles
    // per block 0.81-0.93x (32 -> 32 @96^3, up4.0, up3.0, up2.0), 6.75 tiles per block 1.02-1.08x (64 -> 64 @48^3, 128 -> 128 @24^3, down1.0)
    if (use_tr && cfg != 1 && (long)tiles_total >= (long)rs_wgrad2_min_tiles(-1) * p.splits && rs_wgrad2_supported(p, dtype)) {
        const int rc = rs_launch_wgrad2(p, st);
        if (rc != RS_OK) return rc;
        launch_reduce(p, st);
        return rs_check_launch();
    }
    if (dtype == RS_BF16) {
        // bf16 launches the second-generation kernel above did not take: few tiles per block, the 9-tap configuration, RSUPER_WGRAD_TR=0
        if (cfg == 0) return use_tr ? launch<bf16_t, 1, 27, 1, 8>(p, st) : launch<bf16_t, 1, 27, 0, 8>(p, st);
        if (cfg == 1) return use_tr ? launch<bf16_t, 2, 9, 1, 4>(p, st) : launch<bf16_t, 2, 9, 0, 4>(p, st);
        return use_tr ? launch<bf16_t, 2, 27, 1, 8>(p, st) : launch<bf16_t, 2, 27, 0, 8>(p, st);
    }
    return RS_ERR_ARG;
}
