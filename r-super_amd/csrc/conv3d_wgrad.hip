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
constexpr int WG_PC_BD = 7;                                      // producer/consumer kernel: one MFMA wave per SIMD, latency covered by distance
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
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    const int mg = blockIdx.y % mgroups;
    const int kdg = NTAPS == 27 ? 0 : blockIdx.y / mgroups;      // kd handled by this block (9-tap config)
    const int m0 = mg * MT * 32;
    const bool norm = xs.mr != nullptr;

    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d * p.N;

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
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        const int td = t % tiles_d; t /= tiles_d;
        n = t; d0 = td * TD; h0 = th * TH; w0 = tw * TW;
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
    // XCD-aware tile order: linear workgroup id b runs on XCD b % 8 (private L2 each), and for a fixed (chunk, row group) the splits z, z + 8, ...
    // share an XCD.  Class z & 7 owns a contiguous range of tiles and its blocks sweep it together, so the halo rows neighbouring tiles share are
    // fetched into that L2 once (the grid-stride order put the 8 w / h neighbours of a tile on 8 different XCDs).  Any placement gives the same sums
    // per slab set; only the assignment of tiles to slabs changes.
    int tile0, tile_end, tstride;
    {
        const int z = blockIdx.z, S = p.splits;
        if (S >= 8 && tiles >= 64) {
            const int cls = z & 7, q = S >> 3, rm = S & 7;
            const int cum0 = cls * q + (cls < rm ? cls : rm), ncl = q + (cls < rm ? 1 : 0);
            tile0 = (int)((long)tiles * cum0 / S) + (z >> 3);
            tile_end = (int)((long)tiles * (cum0 + ncl) / S);
            tstride = ncl;
        } else { tile0 = z; tile_end = tiles; tstride = S; }
    }
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
    float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * cin_total;
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

// =====================================================================================================================
// Double-buffered variant (bf16, 27 taps, 8 waves): the classic kernel above stops the matrix pipes twice per tile -- while
// all waves normalise + write the next tile into the single LDS buffer (~3.4k cycles) and at the barrier after it
// (RS_WG_PROF: barrier 3.0k + commit 3.4k + barrier 0.9k + MFMA 5.2k cycles per tile on 64 -> 64 @48^3).  Here the tile
// after next is in flight in registers, the next tile is normalised and written into the OTHER buffer one dword slice per
// MFMA unit, and one barrier per tile is left.  dY rows of the 64-row configuration use an unpadded 128-byte pitch with the
// two 64-byte halves swapped on rows with bit 1 set, which keeps the four rows of a ds_read_b64_tr_b16 group on disjoint
// bank ranges (the padded 192-byte pitch would not fit twice).
// =====================================================================================================================
#ifdef RS_EXPERIMENTAL                                          // measured slower than the single-buffer kernel: `make EXPERIMENTAL=1` only
template <int MT>
__global__ __launch_bounds__(512, 2) void wgrad_db_kernel(WgradParams p) {
    typedef bf16_t T;
    constexpr int NW = 8, NT = 512, NTAPS = 27, TR = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = 8, XP = 64, YP = MT * 64;
    constexpr int HDN = TD + 2;
    constexpr int XROWS = HDN * HH * HW;
    constexpr int XV = 4, YV = MT * 4;
    constexpr int BUF = XROWS * XP + 256 * YP;                   // bytes per buffer
    constexpr int WT = NW / MT;
    constexpr int TPW = (NTAPS + WT - 1) / WT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MT, wt = wave / MT;
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    const int mg = blockIdx.y % mgroups;
    const int m0 = mg * MT * 32;
    const bool norm = xs.mr != nullptr;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles_per_sample = tiles_w * tiles_h * tiles_d;
    const int tiles = tiles_per_sample * p.N;
    const int nitems = ((int)blockIdx.z < tiles) ? (tiles - 1 - (int)blockIdx.z) / p.splits + 1 : 0;

    f32x16_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // per-tap x_hat fragment offsets (lane part folded in) and the dY fragment offset; buffer 0
    int xb_off[TPW];
    {
        const int wts = __builtin_amdgcn_readfirstlane(wt);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            int tl = wts + i * WT;
            if (tl >= NTAPS) tl = NTAPS - 1;
            const int kd = tl / 9, kh = (tl % 9) / 3, kw = tl % 3;
            xb_off[i] = ((kd * HH + kh) * HW + kw) * XP + frag_lane_off<TR>(XP, lane);
        }
    }
    // dY fragment: rows = voxels; MT == 2: the 64-byte half holding this wave's 32 channels is swapped on rows with bit 1 set.
    // The lane's row inside a 16-voxel group is (g >> 1) * 8 + (q >> 2) (+ 4 for the second read): bit 1 = (q >> 3) & 1.
    const int ya_lane = frag_lane_off<TR>(YP, lane);
    const int ya_half = MT == 2 ? ((wm ^ (((lane & 15) >> 3) & 1)) * 64) : 0;
    const int ya_off = XROWS * XP + ya_lane + ya_half;

    // ------------------------------------------------------------------ staging (all 512 threads)
    constexpr int NXV = (XROWS * XV + NT - 1) / NT, NYV = (256 * YV + NT - 1) / NT;      // 6, 2 or 4
    constexpr int XRS = NT / XV, YRS = NT / YV;
    static_assert(NT % XV == 0 && NT % YV == 0 && (256 * YV) % NT == 0, "staging rows per vector index must be whole");
    uint4 px[NXV], py[NYV];
    uint32_t xmask = 0, ymask = 0;
    const int xs_slot = tid % XV, xs_row = tid / XV;
    const int ys_slot = tid % YV, ys_row = tid / YV;
    const bool x_cok = c0 + xs_slot * KP < xs.C;
    const int ym = m0 + ys_slot * KP;
    const T* ysrc = nullptr; int yld = 0;
    if (ym < Mtot) {
        if (ym < p.ya.C) { ysrc = (const T*)p.ya.x + ym; yld = p.ya.ld; }
        else { ysrc = (const T*)p.yb.x + (ym - p.ya.C); yld = p.yb.ld; }
    }
    const int x_st = xs_row * XP + xs_slot * 16;
    int xdelta[NXV], ydelta[NYV], y_st[NYV];
    uint32_t xpm[NXV], ypm[NYV];
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
        const int r = ys_row + i * YRS;                          // voxel of the tile: (r/64, (r/16)%4, r%16)
        ydelta[i] = ((r >> 6) * p.H + ((r >> 4) & 3)) * p.W + (r & 15);
        ypm[i] = ysrc != nullptr ? ((1u << (r >> 6)) | (1u << (4 + ((r >> 4) & 3))) | (1u << (8 + (r & 15)))) : (1u << 31);
        const int slot = MT == 2 ? (ys_slot ^ (((r >> 1) & 1) * 4)) : ys_slot;          // half swap on rows with bit 1 set
        y_st[i] = XROWS * XP + r * YP + slot * 16;
    }
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    const uint32_t xrowb = (uint32_t)xs.ld * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, nvox_total * xrowb, 0x00020000);
    const uint32_t xcol = (uint32_t)(c0 + xs_slot * KP) * 2u;
    auto ld16 = [&](uint32_t off) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
        return make_uint4(q[0], q[1], q[2], q[3]);
    };
    struct IssueTile { int base, ybase; uint32_t xbad, ybad; };
    auto prepare = [&](int it) {
        IssueTile t;
        int tt = (int)blockIdx.z + (it < nitems ? it : 0) * p.splits;
        const int tw = tt % tiles_w; tt /= tiles_w;
        const int th = tt % tiles_h; tt /= tiles_h;
        const int td = tt % tiles_d; tt /= tiles_d;
        const int n = tt, d0 = td * TD, h0 = th * TH, w0 = tw * TW;
        auto range = [](int o, int len, int nh) {
            const int lo = o >= 0 ? 0 : -o;
            int hi_ = len - 1 - o; if (hi_ > nh - 1) hi_ = nh - 1;
            return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
        };
        t.xbad = it < nitems ? ~(range(d0 - 1, p.D, HDN) | (range(h0 - 1, p.H, HH) << 6) | (range(w0 - 1, p.W, HW) << 12)) : 0xFFFFFFFFu;
        t.ybad = it < nitems ? ~(range(d0, p.D, TD) | (range(h0, p.H, TH) << 4) | (range(w0, p.W, TW) << 8)) : 0xFFFFFFFFu;
        t.base = ((n * p.D + d0 - 1) * p.H + (h0 - 1)) * p.W + (w0 - 1);
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
        const T* src = ok ? ysrc + (size_t)(uint32_t)(t.ybase + ydelta[i]) * (uint32_t)yld : (const T*)p.ya.x;
        py[i] = *(const uint4*)src;
        ymask = (ymask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    float sc_[KP], nb_[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { sc_[j] = 1.f; nb_[j] = 0.f; }
    int cur_n = -1;
    auto load_norm = [&](int it) {                               // per-sample statistics of this thread's 8 channels
        if (!norm) return;
        const int n = ((int)blockIdx.z + (it < nitems ? it : 0) * p.splits) / tiles_per_sample;
        if (n == cur_n) return;
        cur_n = n;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int c = c0 + xs_slot * KP + j;
            const float mu = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2] : 0.f, rs = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2 + 1] : 1.f;
            sc_[j] = rs; nb_[j] = -mu * rs;
        }
    };
    auto norm_dword = [&](int i, int j, uint32_t xm) {
        uint32_t* q = j == 0 ? &px[i].x : j == 1 ? &px[i].y : j == 2 ? &px[i].z : &px[i].w;
        if (!norm) return;
        const uint32_t w = *q;
        f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        const f32x2_t s2 = {sc_[2 * j], sc_[2 * j + 1]}, b2 = {nb_[2 * j], nb_[2 * j + 1]};
        x = __builtin_elementwise_fma(x, s2, b2);
        i16x2_t v = __builtin_bit_cast(i16x2_t, f2bf2(x[0], x[1]));
        const i16x2_t z = {0, 0};
        v = __builtin_elementwise_max(v, z);
        const uint32_t m = (uint32_t)((int32_t)(xm << (31 - i)) >> 31);
        *q = __builtin_bit_cast(uint32_t, v) & m;                // padding stays zero AFTER the activation
    };
    auto store_x = [&](char* buf, int i) {
        if (xs_row + i * XRS < XROWS) *(uint4*)(buf + x_st + i * (XRS * XP)) = px[i];
    };
    auto store_y = [&](char* buf, int i, uint32_t ym_) {
        const uint32_t m = (uint32_t)((int32_t)(ym_ << (31 - i)) >> 31);
        *(uint4*)(buf + y_st[i]) = make_uint4(py[i].x & m, py[i].y & m, py[i].z & m, py[i].w & m);
    };

    // ------------------------------------------------------------------ prologue: tile 0 staged synchronously, tile 1 in flight
    {
        const IssueTile t0 = prepare(0);
#pragma unroll
        for (int i = 0; i < NXV; ++i) issue_x(t0, i);
#pragma unroll
        for (int i = 0; i < NYV; ++i) issue_y(t0, i);
        load_norm(0);
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) norm_dword(i, j, xmask);
            store_x(smem, i);
        }
#pragma unroll
        for (int i = 0; i < NYV; ++i) store_y(smem, i, ymask);
        const IssueTile t1 = prepare(1);
#pragma unroll
        for (int i = 0; i < NXV; ++i) issue_x(t1, i);
#pragma unroll
        for (int i = 0; i < NYV; ++i) issue_y(t1, i);
    }
    __syncthreads();

    for (int it = 0; it < nitems; ++it) {
        const char* buf = smem + (it & 1) * BUF;
        char* nxt = smem + ((it + 1) & 1) * BUF;
        load_norm(it + 1);                                       // statistics of the tile committed during this one
        const IssueTile t2 = prepare(it + 2);
        const uint32_t xm_c = xmask, ym_c = ymask;
        auto fetch_a = [&](int row) { return frag_bf16<TR>(buf + ya_off + (row * TW) * YP, YP); };
        auto fetch_b = [&](int row, int i) {
            const int dd = row / TH, hh = row % TH;
            return frag_bf16<TR>(buf + xb_off[i] + ((dd * HH + hh) * HW) * XP, XP);
        };
        constexpr int NU = TD * TH * TPW, BD = WG_BD, BR = WG_BD + 1;
        // hook schedule (MFMA units): x vector i = four dword slices at units C0 + 5 i .. + 3 (16-byte LDS write with the 4th),
        // its reload for the tile after next one unit later; then the dY vectors (write, reload)
        constexpr int C0 = 2, CY = C0 + 5 * NXV;
        static_assert(CY + 2 * NYV < NU, "staging must fit into one tile");
        uint4 aq[2], bq[BR];
        aq[0] = fetch_a(0);
#pragma unroll
        for (int u = 0; u < BD; ++u) bq[u] = fetch_b(u / TPW, u % TPW);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = u / TPW, i = u % TPW;
            if (u + BD < NU) bq[(u + BD) % BR] = fetch_b((u + BD) / TPW, (u + BD) % TPW);
            if (i == 0 && row + 1 < TD * TH) aq[(row + 1) & 1] = fetch_a(row + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma32<bf16_t>(acc[i], aq[row & 1], bq[u % BR]);      // invalid taps accumulate into a slot that is never stored
            if (u >= C0 && u < CY) {
                const int v = (u - C0) / 5, j = (u - C0) % 5;
                if (j < 4) norm_dword(v, j, xm_c);
                if (j == 3) store_x(nxt, v);
                if (j == 4) issue_x(t2, v);
            }
            if (u >= CY && u < CY + 2 * NYV) {
                const int v = (u - CY) / 2;
                if ((u - CY) % 2 == 0) store_y(nxt, v, ym_c); else issue_y(t2, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                         // tile `it` consumed, tile it + 1 complete in the other buffer
    }
    // ---- write this split's partial dW slab: ws[split][tap][m][cin]
    const int ci = c0 + (lane & 31);
    float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * cin_total;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tl = wt + i * WT;
        if (tl >= NTAPS) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + cd_row32(r, lane);
            if (m < Mtot && ci < xs.C) slab[((size_t)tl * Mtot + m) * cin_total + cin_base + (lane & 31)] = acc[i][r];
        }
    }
}

#endif

// =====================================================================================================================
// Producer / consumer (wave-specialised) variant, bf16.  NCW consumer waves run the VALU-free MFMA loop (two
// ds_read_b64_tr_b16 + one MFMA per unit) on one LDS buffer while four producer waves stage the next tile of the split
// (global loads, norm+ReLU, LDS writes) into the other: the staging VALU no longer serialises with the MFMAs inside a
// wave (the classic kernel has <= 2 waves per SIMD).  One block barrier per tile.  Both operands come from LDS, so
// unlike the igemm there is no L1 weight-fragment stream to bound the consumers.
//   PF2: producers keep two tiles of loads in flight (two register sets); off for the 12-wave configuration (168 VGPRs).
// =====================================================================================================================
template <int MT, int NTAPS, int TR, int NCW, bool PF2>
__global__ __launch_bounds__(64 * (NCW + 4), NCW == 8 ? 3 : 2) void wgrad_pc_kernel(WgradParams p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = 8, XP = 64, YP = MT * 64;                 // unpadded rows: 2-way conflicts only on the (rare) dY fragment reads
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;
    constexpr int XROWS = HDN * HH * HW;
    constexpr int XV = 4, YV = MT * 4;
    constexpr int BUF = XROWS * XP + 256 * YP;                   // bytes per buffer
    constexpr int WT = NCW / MT;
    constexpr int TPW = (NTAPS + WT - 1) / WT;
    constexpr int NT = 256;                                      // producer threads

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= NCW;
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    const int mg = blockIdx.y % mgroups;
    const int kdg = NTAPS == 27 ? 0 : blockIdx.y / mgroups;
    const int m0 = mg * MT * 32;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles_per_sample = tiles_w * tiles_h * tiles_d;
    const int tiles = tiles_per_sample * p.N;
    const int nitems = ((int)blockIdx.z < tiles) ? (tiles - 1 - (int)blockIdx.z) / p.splits + 1 : 0;

    if (producer) {
        // ------------------------------------------------------------------ producer waves
        const int ptid = tid - 64 * NCW;
        const bool norm = xs.mr != nullptr;
        constexpr int NXV = (XROWS * XV + NT - 1) / NT, NYV = (256 * YV + NT - 1) / NT;
        constexpr int XRS = NT / XV, YRS = NT / YV;
        const int xs_slot = ptid % XV, xs_row = ptid / XV;
        const int ys_slot = ptid % YV, ys_row = ptid / YV;
        const bool x_cok = c0 + xs_slot * KP < xs.C;
        const int ym = m0 + ys_slot * KP;
        const T* ysrc = nullptr; int yld = 0;
        if (ym < Mtot) {
            if (ym < p.ya.C) { ysrc = (const T*)p.ya.x + ym; yld = p.ya.ld; }
            else { ysrc = (const T*)p.yb.x + (ym - p.ya.C); yld = p.yb.ld; }
        }
        float sc_[KP], nb_[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) { sc_[j] = 1.f; nb_[j] = 0.f; }
        int xdelta[NXV], ydelta[NYV];
        uint32_t xrows_ok = 0, yrows_ok = 0;
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int r = xs_row + i * XRS;
            const int hd = r / (HH * HW);
            const int rem = r - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            xdelta[i] = (hd * p.H + hh) * p.W + hw;
            xrows_ok |= (r < XROWS && x_cok) ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NYV; ++i) {
            const int r = ys_row + i * YRS;
            ydelta[i] = ((r >> 6) * p.H + ((r >> 4) & 3)) * p.W + (r & 15);
            yrows_ok |= (r < 256 && ysrc != nullptr) ? (1u << i) : 0u;
        }
        const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
        const uint32_t xrowb = (uint32_t)xs.ld * (uint32_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, nvox_total * xrowb, 0x00020000);
        const uint32_t xcol = (uint32_t)(c0 + xs_slot * KP) * (uint32_t)sizeof(T);
        auto ld16 = [&](uint32_t off) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
            return make_uint4(q[0], q[1], q[2], q[3]);
        };
        auto tile_of = [&](int it) { return (int)blockIdx.z + it * p.splits; };
        auto issue = [&](int it, uint4* px, uint4* py, uint32_t& xmask) {
            int t = tile_of(it);
            const int tw = t % tiles_w; t /= tiles_w;
            const int th = t % tiles_h; t /= tiles_h;
            const int td = t % tiles_d; t /= tiles_d;
            const int n = t, d0 = td * TD, h0 = th * TH, w0 = tw * TW;
            const int dlo = d0 + (NTAPS == 27 ? -1 : kdg - 1);
            const bool x_in = dlo >= 0 && dlo + HDN <= p.D && h0 >= 1 && h0 + TH + 1 <= p.H && w0 >= 1 && w0 + TW + 1 <= p.W;
            const bool y_in = d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;
            if (x_in) {
                const int base = ((n * p.D + dlo) * p.H + (h0 - 1)) * p.W + (w0 - 1);
                xmask = xrows_ok;
#pragma unroll
                for (int i = 0; i < NXV; ++i) px[i] = ld16(((xrows_ok >> i) & 1u) ? (uint32_t)(base + xdelta[i]) * xrowb + xcol : 0xFFFFFFFFu);
            } else {
                xmask = 0;
#pragma unroll
                for (int i = 0; i < NXV; ++i) {
                    const int r = xs_row + i * XRS;
                    const int hd = r / (HH * HW);
                    const int rem = r - hd * (HH * HW);
                    const int hh = rem / HW, hw = rem - hh * HW;
                    const int d = dlo + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
                    const bool ok = ((xrows_ok >> i) & 1u) && d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W;
                    const uint32_t vox = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
                    px[i] = ld16(ok ? vox * xrowb + xcol : 0xFFFFFFFFu);
                    xmask |= ok ? (1u << i) : 0u;
                }
            }
            const int ybase = ((n * p.D + d0) * p.H + h0) * p.W + w0;
#pragma unroll
            for (int i = 0; i < NYV; ++i) {
                const int r = ys_row + i * YRS;
                const int d = d0 + (r >> 6), h = h0 + ((r >> 4) & 3), w = w0 + (r & 15);
                const bool ok = ((yrows_ok >> i) & 1u) && (y_in || (d < p.D && h < p.H && w < p.W));
                const T* src = ok ? ysrc + (size_t)(uint32_t)(ybase + ydelta[i]) * (uint32_t)yld : (const T*)p.ya.x;   // branch-free
                const uint4 q = *(const uint4*)src;
                py[i] = ok ? q : make_uint4(0, 0, 0, 0);
            }
        };
        int cur_n = -1;
        auto commit = [&](int it, const uint4* px, const uint4* py, const uint32_t xmask) {
            const int n = tile_of(it) / tiles_per_sample;
            if (norm && n != cur_n) {
                cur_n = n;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const int c = c0 + xs_slot * KP + j;
                    const float mu = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2] : 0.f, rs = c < xs.C ? xs.mr[((size_t)n * xs.C + c) * 2 + 1] : 1.f;
                    sc_[j] = rs; nb_[j] = -mu * rs;
                }
            }
            char* buf = smem + (it & 1) * BUF;
            char* x_lds = buf + xs_row * XP + xs_slot * 16;
            char* y_lds = buf + XROWS * XP + ys_row * YP + ys_slot * 16;
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
                if (ys_row + i * YRS < 256) *(uint4*)(y_lds + i * (YRS * YP)) = py[i];
        };
        constexpr int NXV2 = PF2 ? NXV : 1, NYV2 = PF2 ? NYV : 1;
        uint4 pxA[NXV], pyA[NYV], pxB[NXV2], pyB[NYV2];
        uint32_t xmA = 0, xmB = 0;
#ifdef RS_WG_SKIP_PROD                                          // ablation: barriers only (the consumers alone)
        __syncthreads();
        for (int it = 0; it < nitems; ++it) __syncthreads();
        return;
#endif
        if constexpr (PF2) {
            // even items in set A, odd items in set B; loads are issued two items ahead
            if (nitems > 0) { issue(0, pxA, pyA, xmA); commit(0, pxA, pyA, xmA); }
            if (nitems > 1) issue(1, pxB, pyB, xmB);
            if (nitems > 2) issue(2, pxA, pyA, xmA);
            __syncthreads();
            for (int it = 0; it < nitems; it += 2) {
                if (it + 1 < nitems) {
                    commit(it + 1, pxB, pyB, xmB);
                    if (it + 3 < nitems) issue(it + 3, pxB, pyB, xmB);
                }
                __syncthreads();
                if (it + 1 < nitems) {
                    if (it + 2 < nitems) {
                        commit(it + 2, pxA, pyA, xmA);
                        if (it + 4 < nitems) issue(it + 4, pxA, pyA, xmA);
                    }
                    __syncthreads();
                }
            }
        } else {
            if (nitems > 0) { issue(0, pxA, pyA, xmA); commit(0, pxA, pyA, xmA); }
            if (nitems > 1) issue(1, pxA, pyA, xmA);
            __syncthreads();
            for (int it = 0; it < nitems; ++it) {
                if (it + 1 < nitems) {
                    commit(it + 1, pxA, pyA, xmA);
                    if (it + 2 < nitems) issue(it + 2, pxA, pyA, xmA);
                }
                __syncthreads();
            }
        }
    } else {
        // ------------------------------------------------------------------ consumer waves
        const int wm = wave % MT, wt = wave / MT;
        f32x16_t acc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int xb_off[TPW];                                         // per-tap fragment base (lane part folded in), buffer 0
        {
            const int wts = __builtin_amdgcn_readfirstlane(wt);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                int tl = wts + i * WT;
                if (tl >= NTAPS) tl = NTAPS - 1;
                const int kd = NTAPS == 27 ? tl / 9 : 0, kh = (tl % 9) / 3, kw = tl % 3;
                xb_off[i] = ((kd * HH + kh) * HW + kw) * XP + frag_lane_off<TR>(XP, lane);
            }
        }
        const int ya_off = XROWS * XP + wm * 64 + frag_lane_off<TR>(YP, lane);
        __syncthreads();                                         // item 0 staged
        for (int it = 0; it < nitems; ++it) {
            const char* buf = smem + (it & 1) * BUF;
            auto fetch_a = [&](int row) { return frag_bf16<TR>(buf + ya_off + (row * TW) * YP, YP); };
            auto fetch_b = [&](int row, int i) {
                const int dd = row / TH, hh = row % TH;
                return frag_bf16<TR>(buf + xb_off[i] + ((dd * HH + hh) * HW) * XP, XP);
            };
            constexpr int NU = TD * TH * TPW, BD = WG_PC_BD, BR = WG_PC_BD + 1;
            uint4 aq[2], bq[BR];
            aq[0] = fetch_a(0);
#pragma unroll
            for (int u = 0; u < BD; ++u) bq[u] = fetch_b(u / TPW, u % TPW);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int row = u / TPW, i = u % TPW;
                if (u + BD < NU) bq[(u + BD) % BR] = fetch_b((u + BD) / TPW, (u + BD) % TPW);
                if (i == 0 && row + 1 < TD * TH) aq[(row + 1) & 1] = fetch_a(row + 1);
                __builtin_amdgcn_sched_barrier(0);
#ifndef RS_WG_SKIP_CONS                                         // ablation: operand fetches without the MFMAs would be optimised away -> skip the whole unit
                mma32<bf16_t>(acc[i], aq[row & 1], bq[u % BR]);
#else
                if (u == 0) mma32<bf16_t>(acc[i], aq[row & 1], bq[u % BR]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        // ---- write this split's partial dW slab: ws[split][tap][m][cin]
        const int ci = c0 + (lane & 31);
        float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * cin_total;
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

#ifdef RS_EXPERIMENTAL
template <int MT>
int launch_db(const WgradParams& p, hipStream_t st) {
    const size_t smem = 2 * ((size_t)(TD + 2) * HH * HW * 64 + 256 * MT * 64);
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    dim3 grid(nch, mgroups, p.splits), block(512);
    auto k = wgrad_db_kernel<MT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    launch_reduce(p, st);
    return rs_check_launch();
}
#endif

template <int MT, int NTAPS, int TR, int NCW, bool PF2>
int launch_pc(const WgradParams& p, hipStream_t st) {
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;
    const size_t smem = 2 * ((size_t)HDN * HH * HW * 64 + 256 * MT * 64);
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    dim3 grid(nch, mgroups * (NTAPS == 27 ? 1 : 3), p.splits), block(64 * (NCW + 4));
    auto k = wgrad_pc_kernel<MT, NTAPS, TR, NCW, PF2>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
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

int rs_wgrad_splits(int dtype, int Mtot, int nch, int tiles_total) {
    const int cfg = rs_wgrad_config(dtype, Mtot, tiles_total);
    const int gy = cfg == 0 ? 1 : (cfg == 1 ? 3 : 1) * ((Mtot + 63) / 64);
    const int target = cfg == 1 ? 512 : 256;                     // resident blocks on 256 CUs
    int s = target / (nch * gy > 0 ? nch * gy : 1);
    if (s > tiles_total) s = tiles_total;
    return s < 1 ? 1 : s;
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
    // Second-generation kernel (conv3d_wgrad2.hip: operand re-use across taps, double-buffered tiles) where a block sweeps enough tiles to amortise its
    // heavier prologue (descriptor / constants tables, two tiles staged before the first MFMA): measured same-box against the kernel below, >= 12 tiles
    // per block 0.81-0.93x (32 -> 32 @96^3, up4.0, up3.0, up2.0), 6.75 tiles per block 1.02-1.08x (64 -> 64 @48^3, 128 -> 128 @24^3, down1.0)
    static const int w2min = getenv("RSUPER_WGRAD2_MIN_TILES") ? atoi(getenv("RSUPER_WGRAD2_MIN_TILES")) : 12;
    if (use_tr && cfg != 1 && tiles_total >= w2min * p.splits && rs_wgrad2_supported(p, dtype)) {
        const int rc = rs_launch_wgrad2(p, st);
        if (rc != RS_OK) return rc;
        launch_reduce(p, st);
        return rs_check_launch();
    }
    if (use_tr && cfg != 1 && rs_wgrad_dma_supported(p, dtype)) {      // pre-normalised sources: operands by LDS-DMA (conv3d_wgrad_dma.hip)
        const int rc = rs_launch_wgrad_dma(p, st);
        if (rc != RS_OK) return rc;
        launch_reduce(p, st);
        return rs_check_launch();
    }
    if (dtype == RS_BF16) {
        // config 0 runs the producer/consumer kernel (177 -> 131 us on 32->32 @96^3); on configs 1/2 it measured equal or
        // slower (12 waves hit the 168-VGPR cap) and the classic kernel stays
#ifdef RS_EXPERIMENTAL
        static const int db = getenv("RSUPER_WGRAD_DB") ? atoi(getenv("RSUPER_WGRAD_DB")) : 0;   // double-buffered kernel (opt-in: measured slower than the single-buffer kernel)
        if (use_tr && db && cfg == 0) return launch_db<1>(p, st);
        if (use_tr && db && cfg == 2) return launch_db<2>(p, st);
#endif
        static const int c0 = getenv("RSUPER_WGRAD_CFG0") ? atoi(getenv("RSUPER_WGRAD_CFG0")) : 2;   // 0: producer/consumer kernel, 1 / 2: classic kernel with 4 / 8 waves (155 vs 161 us on 32 -> 32 @96^3)
        if (use_tr && cfg == 0 && c0 == 1) return launch<bf16_t, 1, 27, 1, 4>(p, st);
        if (use_tr && cfg == 0 && c0 == 2) return launch<bf16_t, 1, 27, 1, 8>(p, st);
        if (use_tr) return cfg == 0 ? launch_pc<1, 27, 1, 4, true>(p, st) : cfg == 1 ? launch<bf16_t, 2, 9, 1, 4>(p, st) : launch<bf16_t, 2, 27, 1, 8>(p, st);
        return cfg == 0 ? launch_pc<1, 27, 0, 4, true>(p, st) : cfg == 1 ? launch<bf16_t, 2, 9, 0, 4>(p, st) : launch<bf16_t, 2, 27, 0, 8>(p, st);
    }
    return RS_ERR_ARG;
}
