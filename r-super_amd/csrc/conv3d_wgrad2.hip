// Weight gradient of the 3x3x3 convolution on MFMA, second generation (gfx950, bf16): operand re-use across taps + double-buffered
// tiles whose staging is interleaved into the MFMA loop.
//
//   dW[co][ci][tap] = sum_v dY[v][co] * x_hat[v + off(tap)][ci]          (GEMM: M = Cout, N = 27*Cin, K = voxels)
//
// Replaces the autograd weight-gradient of nn.Conv3d(k=3) (rsuper_train/model/dim3/conv_layers.py:29-38 under loss.backward(), train_ddp.py:349).
// Same contraction, tiling (4x4x16 voxels x 32 input channels per block), fragment reads (ds_read_b64_tr_b16 on 64-byte rows), slab format and
// reduction as conv3d_wgrad.hip; x_hat = relu((x - mean) * rstd) is recomputed from x while a tile is staged (never stored).
//
// What conv3d_wgrad.hip loses per 256-voxel tile (RS_WG_PROF cycle stamps, 64-row x 27-tap blocks): the two waves of a SIMD need 6.0 k / 9.2 k cycles
// for an MFMA phase whose matrix-pipe time is 7.2 k, then the pipes stand still for ~2.6 k cycles of commit (InstanceNorm + ReLU on the staged
// vectors, ds_write_b128) and ~1 k of barrier skew: ~12 k cycles per 6.9 k of MFMA.  Counters: 2.3 LDS instructions per MFMA, LDS pipe 57 % busy.
//   (1) Operand re-use.  A fragment x_hat[(d', h'), kw .. kw+15] is the B operand of EVERY tap (kd, kh, kw) whose output row (d' - kd, h' - kh) lies
//       in the tile: up to nine MFMAs with nine dY rows and accumulators.  A wave owns one kw and seven of the nine (kd, kh) pairs (tap groups 0-2;
//       group 3 takes the two left-over pairs at all three kw: 7 + 7 + 7 + 6 = 27 taps), keeps the dY fragments of three tile planes in registers
//       (each loaded once per tile), walks the halo rows and issues all MFMAs a fragment feeds: 34 + 16 fragment reads per 112 MFMAs = 0.9 LDS
//       instructions per MFMA.  The 8 waves are (tap group) x (32-row group of dY) for 64-row blocks, (tap group) x (depth half of the tile) for
//       32-row blocks, whose two partial sums meet in LDS once at the end of the block.
//   (2) Two tile buffers, ONE barrier per tile, no commit phase: with 0.9 instead of 2.3 LDS instructions per MFMA the wave has issue slots to spare,
//       so the staging of tile t + 1 (norm + ReLU in registers, ds_write_b128 into the other buffer) and the global loads of tile t + 2 are hooks in
//       the MFMA loop of tile t -- one 16-byte vector per hook (the double-buffered variant of round 2 did this with 2.3 LDS instructions per MFMA in
//       the same loop and lost; RSUPER_WGRAD_DB, removed).
#include "common.hpp"
#include "kernels.hpp"
#include "wgrad_frag.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

#ifdef WG2_PROF
__device__ unsigned long long g_wg2_prof[8 * 4];                 // block 0: [wave][loop cycles, barrier cycles, tiles, -]
#endif

constexpr int TD = 4, TH = 4, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int XROWS = HD * HH * HW;                              // 648 halo rows of 64 B (32 channels)
constexpr int XBYTES = XROWS * 64;
constexpr int YPLANE = 256 * 64;                                 // one 32-row group of dY: 256 voxels x 64 B
constexpr int NT = 512, NW = 8;

// dY needs no arithmetic on its way into LDS: `buffer_load_dwordx4 ... lds` copies 1 KB per wave instruction from per-lane global offsets into a
// lane-linear LDS image (lane l -> M0 + 16 l), out-of-range offsets arrive as zeros; semantics pinned by tools/ubench/lds_dma_probe.hip.  The
// compiler does not see the instruction: completion is waited for by an explicit counted vmcnt before the tile barrier.
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_byte), "s"(rs) : "memory", "m0");
}

// ---- static schedule of one wave's tile: the halo fragments it fetches, in order.  Pair index pidx = kd * 3 + kh; groups 0-2 own pairs 0..6 at kw = G,
//      group 3 owns pairs 7, 8 = (kd 2, kh 1 / 2) at every kw.  NDL = tile planes handled by the wave (4, or 2 for the depth halves of 32-row blocks).
constexpr bool pair_in_group(int G, int pidx) { return G < 3 ? pidx < 7 : pidx >= 7; }
constexpr bool unit_valid(int NDL, int dp, int hp, int pidx) {
    const int d = dp - pidx / 3, h = hp - pidx % 3;
    return d >= 0 && d < NDL && h >= 0 && h < TH;
}
constexpr int frag_uses(int G, int NDL, int dp, int hp) {
    int n = 0;
    for (int pidx = 0; pidx < 9; ++pidx) n += (pair_in_group(G, pidx) && unit_valid(NDL, dp, hp, pidx)) ? 1 : 0;
    return n;
}
struct Sched { int n; int dp[64], hp[64], kw[64]; };
constexpr Sched make_sched(int G, int NDL) {
    Sched s{};
    for (int dp = 0; dp < NDL + 2; ++dp)
        for (int hp = 0; hp < HH; ++hp) {
            if (frag_uses(G, NDL, dp, hp) == 0) continue;
            for (int kw = (G < 3 ? G : 0); kw <= (G < 3 ? G : 2); ++kw) { s.dp[s.n] = dp; s.hp[s.n] = hp; s.kw[s.n] = kw; ++s.n; }
        }
    return s;
}

// MT: 32-row groups of dY per block.  wave -> (wsel = wave & 1, g = wave >> 1): MT 2: wsel = row group, all 16 (d, h) rows of the tile;
// MT 1: wsel = depth half (planes 2 wsel, 2 wsel + 1), the halves are summed through LDS at the end.
template <int MT, int BD, bool NORM>
__global__ __launch_bounds__(NT, 2) void wgrad2_kernel(WgradParams p) {
    constexpr int NDL = MT == 2 ? 4 : 2;
    constexpr int BUF = XBYTES + MT * YPLANE;
    constexpr int NXV = (XROWS * 4 + NT - 1) / NT;               // 16-byte x vectors per thread (6; the last one on 32 threads only)
    constexpr int YK = 16 * MT / NW;                             // dY DMA pieces per wave (2 / 4): piece = one (d, h) line of a 32-row group, 16 voxels x 64 B
    constexpr int NV = NXV;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsel = wave & 1, g = wave >> 1;
    const int dsel = MT == 1 ? wsel * 2 : 0;                     // first tile plane of this wave
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles_per_sample = tiles_w * tiles_h * tiles_d;
    const WgBlockMap bm = wg_block_map(tiles_per_sample * p.N, p.splits);   // (chunk, row group, split) of this block and the split's tiles (wgrad_frag.hpp)
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = bm.bx >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? bm.bx - nchA : bm.bx) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int m0 = bm.by * MT * 32;
    constexpr bool norm = NORM;                                  // sources carry (mean, rstd): InstanceNorm + ReLU while staging
    const int tiles = tiles_per_sample * p.N;

    // fragment bases (lane part folded in), relative to a tile buffer
    const int x_off = dsel * (HH * HW * 64) + frag_lane_off<1>(64, lane);
    const int ya_off = XBYTES + (MT == 2 ? wsel * YPLANE : 0) + dsel * (TH * TW * 64) + frag_lane_off<1>(64, lane);

    // ---- staging side (all 512 threads): thread -> 16-byte slot tid & 3 of rows tid / 4 + 128 i
    const int s_slot = tid & 3, s_row = tid >> 2;
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    const uint32_t xrowb = (uint32_t)xs.ld * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, nvox_total * xrowb, 0x00020000);
    uint32_t xoffb[NXV], xpm[NXV];                               // byte offset of the row relative to the halo origin; one-hot (hd | hh << 6 | hw << 12), bit 31: no data
    const bool x_cok = c0 + s_slot * 8 < xs.C;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
        const int r = s_row + 128 * i;
        const int hd = r / (HH * HW), rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        xoffb[i] = (uint32_t)((hd * p.H + hh) * p.W + hw) * xrowb + (uint32_t)(c0 + s_slot * 8) * 2u;
        xpm[i] = (r < XROWS && x_cok) ? ((1u << hd) | (1u << (6 + hh)) | (1u << (12 + hw))) : (1u << 31);
    }
    const int x_st = s_row * 64 + s_slot * 16;                   // LDS byte of vector 0; vector i at + 8192 i
    // dY: plane q = rows m0 + 32 q .. + 31 of [ya | yb] (a 32-row group never straddles the two sources: checked by the launcher).  DMA piece y of a tile
    // = plane y >> 4, line (d, h) = ((y & 15) >> 2, y & 3); lane -> voxel w = lane >> 2, 16-byte slot lane & 3.
    const int prow = lane >> 2, pslot = lane & 3;
    __amdgpu_buffer_rsrc_t yrs[MT];
    uint32_t yrowb[MT], yoffb[MT];
    bool yok[MT];
#pragma unroll
    for (int q = 0; q < MT; ++q) {
        const int mq = m0 + 32 * q;
        const bool inA = mq < p.ya.C;
        const ConvSrc& ys = inA ? p.ya : p.yb;
        const int ch = (inA ? mq : mq - p.ya.C) + pslot * 8;
        yok[q] = mq < Mtot && ch < ys.C;
        yrowb[q] = (uint32_t)ys.ld * 2u;
        yoffb[q] = (uint32_t)prow * yrowb[q] + (uint32_t)ch * 2u;
        yrs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)ys.x, 0, nvox_total * yrowb[q], 0x00020000);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic region
    const uint32_t ywbit = 1u << (8 + prow);                     // this lane's w position in the tile's validity mask

    // XCD-aware tile order (wg_block_map): the splits on one XCD own a contiguous range of tiles and sweep it together
    const int tile0 = bm.tile0, tile_end = bm.tile_end, tstride = bm.tstride;

    // Tile descriptors of this block's tiles, built ONCE (the per-tile index arithmetic -- five integer divisions and the range masks -- cost ~1.2 k
    // cycles at the top of every tile with all matrix pipes idle: 108 of 613 us on up4.0): entry k = k-th tile of the block, entries past the
    // last tile describe "nothing to load" (every vector out of range: zeros).
    struct IssueTile { uint32_t baseb, ybase; uint32_t xbad, ybad; int n; };
    uint4* dtab = (uint4*)(smem + 2 * BUF + p.N * 256);          // [ntile + 3] x (baseb, ybase, xbad, ybad); ybad uses 24 bits (4 d, 4 h, 16 w), the sample index rides in its top byte
    const int ntile = tile0 < tile_end ? (tile_end - 1 - tile0) / tstride + 1 : 0;
    for (int k = tid; k < ntile + 3; k += NT) {
        const bool live = k < ntile;
        const int q = live ? tile0 + k * tstride : 0;
        const int n = q / tiles_per_sample;
        int tw, th, td;
        rs_tile_coords(q - n * tiles_per_sample, tiles_w, tiles_h, tiles_d, tw, th, td);
        const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
        auto range = [](int o, int len, int nh) {                // bits i in [0, nh) with 0 <= o + i < len
            const int lo = o >= 0 ? 0 : -o;
            int hi_ = len - 1 - o; if (hi_ > nh - 1) hi_ = nh - 1;
            return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
        };
        const uint32_t xbad = live ? ~(range(d0 - 1, p.D, HD) | (range(h0 - 1, p.H, HH) << 6) | (range(w0 - 1, p.W, HW) << 12)) : 0xFFFFFFFFu;
        const uint32_t ybad = live ? (~(range(d0, p.D, TD) | (range(h0, p.H, TH) << 4) | (range(w0, p.W, TW) << 8)) & 0xFFFFFFu) : 0xFFFFFFu;
        dtab[k] = make_uint4((uint32_t)(((n * p.D + d0 - 1) * p.H + (h0 - 1)) * p.W + (w0 - 1)) * xrowb,
                             (uint32_t)(((n * p.D + d0) * p.H + h0) * p.W + w0), xbad, ybad | ((uint32_t)n << 24));
    }
    auto fetch_desc = [&](int k) {                               // wave-uniform: one broadcast LDS read + readfirstlanes
        const uint4 v = dtab[k];
        IssueTile t;
        t.baseb = __builtin_amdgcn_readfirstlane(v.x); t.ybase = __builtin_amdgcn_readfirstlane(v.y);
        t.xbad = __builtin_amdgcn_readfirstlane(v.z);
        const uint32_t w = __builtin_amdgcn_readfirstlane(v.w);
        t.ybad = w & 0xFFFFFFu; t.n = (int)(w >> 24);
        return t;
    };
    uint4 px[NXV];
    auto ld16 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t off) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);      // out-of-range offsets return zeros
        return make_uint4(q[0], q[1], q[2], q[3]);
    };
    auto issue_v = [&](const IssueTile& t, int i) {              // i static
        const bool ok = (xpm[i] & t.xbad) == 0u;
        px[i] = ld16(xrs, ok ? t.baseb + xoffb[i] : 0xFFFFFFF0u);
    };
    auto dma_y = [&](const IssueTile& t, int k, uint32_t buf) {  // k static: this wave's k-th dY piece of tile t -> tile buffer at LDS byte offset buf
        const int y = wave + NW * k;
        const int q = MT == 1 ? 0 : (y >> 4), line = y & 15, d = line >> 2, h = line & 3;
        const uint32_t lbit = (1u << d) | (1u << (4 + h)) | ywbit;
        const bool ok = yok[q] && (lbit & t.ybad) == 0u;
        const uint32_t voff = ok ? (t.ybase + (uint32_t)((d * p.H + h) * p.W)) * yrowb[q] + yoffb[q] : 0xFFFFFFF0u;
        dma16(MT == 1 ? yrs[0] : (q ? yrs[MT - 1] : yrs[0]), voff, lds0 + buf + XBYTES + (uint32_t)y * 1024u);
    };
    // x_hat = max(x * rstd - mean * rstd, 0): constants of this thread's 8 channels as (sc0, sc1, nb0, nb1) per pair, re-read from a per-sample LDS table
    // when the sample changes (the dY vectors travel by DMA, which frees the registers these need)
    float4* ntab = (float4*)(smem + 2 * BUF);                    // [N][4 slots][4 pairs]
    if (norm) {
        for (int e = tid; e < p.N * 16; e += NT) {
            const int n = e >> 4, c = c0 + (e & 15) * 2;
            float4 v = make_float4(1.f, 1.f, 0.f, 0.f);
            if (c < xs.C) { const float* m = xs.mr + ((size_t)n * xs.C + c) * 2; v = make_float4(m[1], m[3], -m[0] * m[1], -m[2] * m[3]); }
            ntab[e] = v;
        }
    }
    const float4* nrow = ntab + s_slot * 4;                      // + 16 n
    float4 ncst[4];
    int cur_n = -1;
    auto load_norm = [&](int n) {
        if (!norm || n == cur_n) return;
        cur_n = n;
#pragma unroll
        for (int j = 0; j < 4; ++j) ncst[j] = nrow[16 * n + j];
    };
    auto commit_v = [&](char* buf, int i, const IssueTile& t) {  // i static: normalise and write x vector i of the tile `t` held in px
        {
            uint4 q = px[i];
            if (norm) {
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                uint32_t o[4];
                const uint32_t m = (xpm[i] & t.xbad) == 0u ? 0xFFFFFFFFu : 0u;        // padding stays zero AFTER the activation
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 c = ncst[j];
#ifndef WG2_PKFMA
                    // two v_fma_f32, not one v_pk_fma_f32: beside MFMAs the packed f32 forms cost ~22 cycles more per instruction (MI355X_MICROARCH.md, filler prices)
                    float x0, x1;
                    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(__uint_as_float(w[j] << 16)), "v"(c.x), "v"(c.z));
                    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(__uint_as_float(w[j] & 0xffff0000u)), "v"(c.y), "v"(c.w));
                    i16x2_t v = __builtin_bit_cast(i16x2_t, f2bf2(x0, x1));
#else
                    f32x2_t x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
                    const f32x2_t s2 = {c.x, c.y}, b2 = {c.z, c.w};
                    x = __builtin_elementwise_fma(x, s2, b2);
                    i16x2_t v = __builtin_bit_cast(i16x2_t, f2bf2(x[0], x[1]));
#endif
                    const i16x2_t z = {0, 0};
                    v = __builtin_elementwise_max(v, z);
                    o[j] = __builtin_bit_cast(uint32_t, v) & m;
                }
                q = make_uint4(o[0], o[1], o[2], o[3]);
            }
            if (s_row + 128 * i < XROWS) *(uint4*)(buf + x_st + i * 8192) = q;
        }
    };

    auto run = [&](auto G_) {
        constexpr int G = std::remove_reference_t<decltype(G_)>::value;
        constexpr Sched S = make_sched(G, NDL);
        constexpr int KDMIN = G < 3 ? 0 : 2;                     // first halo plane that uses dY plane 0
        constexpr int AP = NDL < 3 ? NDL : 3;                    // dY planes held in registers (ring)
        // The two waves of a SIMD are the tap groups g and g + 2 (waves w, w + 4).  The SIMD arbitrates by age: the older wave runs nearly unimpeded and the
        // younger one gets what is left (cycle stamps, 32 -> 32 @96^3, staging off: older wave 2.0 k cycles for its 56 MFMAs, then 1.8 k at the barrier;
        // younger 3.7 k).  So the two programs are made complementary instead of identical: the older wave multiplies first and stages in the second
        // half of its steps, the younger one stages in the first half (while the matrix pipe belongs to its partner) and multiplies after.
#ifndef WG2_HOOK_MODE
#define WG2_HOOK_MODE 1
#endif
        constexpr int HMODE = WG2_HOOK_MODE;                     // 0: hooks spread evenly; 1: complementary halves; 2: the other way round
        constexpr bool late = HMODE == 0 ? false : ((G < 2) == (HMODE == 1));
        auto hook_step = [&](int i) constexpr { return HMODE == 0 ? (i * S.n) / NV : (late ? S.n / 2 : 0) + (i * (S.n / 2)) / NV; };
        f32x16_t acc[7];
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        // prologue: tile 0 staged synchronously into buffer 0, tile 1 in flight in registers
        __syncthreads();                                         // descriptor + constants tables
        IssueTile t1 = fetch_desc(1);
        {
            const IssueTile t0 = fetch_desc(0);
#pragma unroll
            for (int k = 0; k < YK; ++k) dma_y(t0, k, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t0, i);
            load_norm(t0.n);
#pragma unroll
            for (int i = 0; i < NV; ++i) commit_v(smem, i, t0);
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t1, i);
        }
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NV) : "memory");      // the dY pieces of tile 0 (the x loads of tile 1 stay in flight)
        __syncthreads();

#ifdef WG2_PROF
        unsigned long long pf_loop = 0, pf_bar = 0;
#endif
        int it = 0;
        for (int tile = tile0; tile < tile_end; tile += tstride, ++it) {
#ifdef WG2_PROF
            const unsigned long long q0 = __builtin_readcyclecounter();
#endif
            const char* buf = smem + (it & 1) * BUF;
            char* nxt = smem + ((it + 1) & 1) * BUF;
            const IssueTile t2 = fetch_desc(it + 2);             // t1: the tile committed during this one; t2: the tile whose loads are issued during it
            load_norm(t1.n);
#ifndef WG2_SKIP_STAGE
#pragma unroll
            for (int k = 0; k < YK; ++k) dma_y(t1, k, (uint32_t)((it + 1) & 1) * BUF);      // first VMEM operations of the tile: NV x loads follow
#endif
            auto fetch_a = [&](int row) { return frag_bf16<1>(buf + ya_off + (row * TW) * 64, 64); };
            auto fetch_b = [&](int s) { return frag_bf16<1>(buf + x_off + ((S.dp[s] * HH + S.hp[s]) * HW + S.kw[s]) * 64, 64); };
            constexpr int BR = BD + 1;
            uint4 a[AP * TH], bq[BR];
#pragma unroll
            for (int h = 0; h < TH; ++h) a[h] = fetch_a(h);
#pragma unroll
            for (int s = 0; s < BD; ++s) bq[s] = fetch_b(s);
#pragma unroll
            for (int s = 0; s < S.n; ++s) {
#ifndef WG2_PRIO
#define WG2_PRIO 1
#endif
                // progress-based priority (conv3d_igemm_kd.hip): the wave of a SIMD that is behind in the tile outranks its partner
                if (WG2_PRIO == 1) {
                    if (s == 0) __builtin_amdgcn_s_setprio(3);
                    if (s == S.n / 4) __builtin_amdgcn_s_setprio(2);
                    if (s == S.n / 2) __builtin_amdgcn_s_setprio(1);
                    if (s == (3 * S.n) / 4) __builtin_amdgcn_s_setprio(0);
                }
                if (s + BD < S.n) bq[(s + BD) % BR] = fetch_b(s + BD);
                // dY plane d is used on halo planes d + KDMIN .. d + 2.  Fully resident (AP = NDL): request it at the first step of halo plane d + KDMIN - 1.
                // Ring of three (AP = 3 < NDL): the slot of plane d + 1 is the one of plane d - 2, whose last use in groups 0-2 is the pair (kd 2, kh 0) at
                // hp <= 3 -> request at the first step with hp >= 4 of halo plane d (group 3 only ever has one plane live: first step of the halo plane).
                {
                    constexpr bool ring = AP < NDL;
                    const int pl = S.dp[s] - KDMIN + 1;          // plane to request during halo plane dp
                    const bool first_of_plane = s == 0 || S.dp[s] != S.dp[s - 1];
                    const bool first_hp4 = S.hp[s] >= 4 && (s == 0 || S.dp[s] != S.dp[s - 1] || S.hp[s - 1] < 4);
                    if (pl >= 1 && pl < NDL && ((ring && G < 3) ? first_hp4 : first_of_plane)) {
#pragma unroll
                        for (int h = 0; h < TH; ++h) a[(pl % AP) * TH + h] = fetch_a(pl * TH + h);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pidx = 0; pidx < 9; ++pidx) {
                    if (!pair_in_group(G, pidx) || !unit_valid(NDL, S.dp[s], S.hp[s], pidx)) continue;
                    const int d = S.dp[s] - pidx / 3, h = S.hp[s] - pidx % 3;
                    const int slot = G < 3 ? pidx : (pidx - 7) * 3 + S.kw[s];
#ifndef WG2_SKIP_MMA                                             // ablation switches (tools/wg2_ablate.sh)
                    mma32<bf16_t>(acc[slot], a[(d % AP) * TH + h], bq[s % BR]);
#else
                    if (s == 0) mma32<bf16_t>(acc[slot], a[(d % AP) * TH + h], bq[s % BR]);
#endif
                }
#ifndef WG2_SKIP_STAGE
                // staging hooks: vector i of tile t + 1 is normalised + written into the other buffer, then its register takes the load of tile t + 2
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (s == hook_step(i)) { commit_v(nxt, i, t1); issue_v(t2, i); }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef WG2_PROF
            const unsigned long long q1 = __builtin_readcyclecounter();
#endif
#ifndef WG2_SKIP_STAGE
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NV) : "memory");  // this tile's dY pieces have landed; the NV x loads issued after them stay in flight
#endif
            __syncthreads();
#ifdef WG2_PROF
            const unsigned long long q2 = __builtin_readcyclecounter();
            pf_loop += q1 - q0; pf_bar += q2 - q1;
#endif                                     // tile t consumed, tile t + 1 complete in the other buffer
            t1 = t2;
        }

#ifdef WG2_PROF
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) { g_wg2_prof[wave * 4] = pf_loop; g_wg2_prof[wave * 4 + 1] = pf_bar; g_wg2_prof[wave * 4 + 2] = (unsigned long long)it; }
#endif
        // ---- 32-row blocks: the two depth halves meet in LDS (the tile buffers are free now); wave wsel = 0 of each pair holds the sum
        if constexpr (MT == 1) {
            float* red = (float*)smem + (size_t)g * 7 * 16 * 64 + lane;
            if (wsel == 1) {
#pragma unroll
                for (int i = 0; i < 7; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(i * 16 + r) * 64] = acc[i][r];
            }
            __syncthreads();
            if (wsel == 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += red[(i * 16 + r) * 64];
            }
        }
        // ---- this split's partial dW slab: ws[split][tap][m][cin]  (coalesced along cin)
        if (MT == 2 || wsel == 0) {
            const int ci = c0 + (lane & 31);
            float* slab = p.ws + (size_t)bm.bz * 27 * Mtot * cin_total;
            const int mrow0 = m0 + (MT == 2 ? wsel * 32 : 0);
#pragma unroll
            for (int i = 0; i < (G < 3 ? 7 : 6); ++i) {
                const int tap = G < 3 ? i * 3 + G : 21 + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow0 + cd_row32(r, lane);
                    if (m < Mtot && ci < xs.C) slab[((size_t)tap * Mtot + m) * cin_total + cin_base + (lane & 31)] = acc[i][r];
                }
            }
        }
    };
    switch (g) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 3>{}); break;
    }
}

template <int MT, int BD, bool NORM>
int launch2(const WgradParams& p, hipStream_t st) {
    const int tiles = p.N * ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    // two tile buffers + the normalisation constants of every sample + the tile descriptors of one block
    const int smem = 2 * (XBYTES + MT * YPLANE) + p.N * 256 + ((tiles + p.splits - 1) / p.splits + 4) * 16;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    static_assert(MT == 2 || 4 * 7 * 16 * 64 * 4 <= 2 * (XBYTES + MT * YPLANE), "depth-half reduction scratch must fit the tile buffers");
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    dim3 grid(nch, (Mtot + MT * 32 - 1) / (MT * 32), p.splits), block(NT);
    auto k = wgrad2_kernel<MT, BD, NORM>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
#ifdef WG2_PROF
    if (getenv("RSUPER_WG2_PROF")) {
        unsigned long long h[32];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wg2_prof), sizeof(h));
        fprintf(stderr, "wg2_prof MT%d norm%d:", MT, (int)NORM);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d(g%d) loop %.0f bar %.0f |", w, w >> 1, (double)h[w * 4] / h[w * 4 + 2], (double)h[w * 4 + 1] / h[w * 4 + 2]);
        fprintf(stderr, " tiles %llu\n", h[2]);
    }
#endif
    return RS_OK;
}

}  // namespace

// bf16; 32-row groups of dY inside one source each (buffer resources are wave-uniform)
bool rs_wgrad2_supported(const WgradParams& p, int dtype) {
    static const int off = getenv("RSUPER_WGRAD2") ? atoi(getenv("RSUPER_WGRAD2")) == 0 : 0;
    if (off || dtype != RS_BF16 || p.N > 32) return false;      // per-sample constants table: 256 B of LDS per sample
    if (p.yb.C > 0 && (p.ya.C % 32)) return false;
    if ((p.xa.mr != nullptr) != (p.xb.C > 0 ? p.xb.mr != nullptr : p.xa.mr != nullptr)) return false;   // both sources normalised or both raw, as everywhere
    return true;
}

int rs_launch_wgrad2(const WgradParams& p, hipStream_t st) {
    const int Mtot = p.ya.C + p.yb.C;
#ifndef WG2_BD
#define WG2_BD 2
#endif
    const int tiles = p.N * ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const bool mt1 = Mtot <= 32 || rs_wgrad2_mt1(RS_BF16, Mtot, tiles);
    if (p.xa.mr) return mt1 ? launch2<1, WG2_BD, true>(p, st) : launch2<2, WG2_BD, true>(p, st);
    return mt1 ? launch2<1, WG2_BD, false>(p, st) : launch2<2, WG2_BD, false>(p, st);
}
