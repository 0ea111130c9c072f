// Weight gradient of the 3x3x3 convolution for PRE-NORMALISED inputs, operands fed by LDS-DMA (gfx950, bf16).
//
//   dW[co][ci][tap] = sum_v dY[v][co] * x_hat[v + off(tap)][ci]          (GEMM: M = Cout, N = 27*Cin, K = voxels)
//
// Same contraction, tiling (4x4x16 voxels, 32 input channels per block), fragment reads (ds_read_b64_tr_b16 on 64-byte rows) and slab
// format as conv3d_wgrad.hip.  Two things differ.
//
// (1) How a tile gets into LDS.  conv3d_wgrad.hip loads 16-byte vectors into registers, applies InstanceNorm + ReLU and writes them with
// ds_write_b128 between two block barriers: the matrix pipes stand still for that commit (RS_WG_PROF, 32 -> 32 @96^3: 3.4-5.0 k cycles of MFMA
// phase, 2.4 k of commit, 1.6 k of barrier wait per tile = 45 % of the MFMA rate), and a producer / consumer split does not help because a wave
// that shares its SIMD with an MFMA-issuing wave gets about one issue slot per MFMA (its ~300 staging instructions per tile then outlast the
// tile; measured: consumers alone 92 us, producers alone 100 us, together 158 us).  Here the source is x_hat itself (bf16, normalised and
// activated), so a tile is a plain copy: `buffer_load_dwordx4 ... lds` moves 1 KB per wave instruction from per-lane global offsets straight into
// a lane-linear LDS image; out-of-range offsets arrive as zeros (= the zero padding of the activated tensor); no VGPRs, no VALU, no ds_write.
// Two tile buffers, the DMA pieces of tile t + 1 are spread over the MFMA loop of tile t, one block barrier per tile (`s_waitcnt vmcnt(0)` +
// `s_barrier`: my pieces have landed, everybody is done reading the other buffer).  The semantics relied upon (lane l -> M0 + 16 l, M0 above
// 64 KB, zeros for out-of-range lanes, visibility after vmcnt(0) + barrier) are pinned by tools/ubench/lds_dma_probe.hip.
//
// (2) Operand re-use in the MFMA loop.  With one x_hat fragment fetched per MFMA the loop needs 2.3 LDS instructions per MFMA and the LDS pipe is
// 57 % busy at 71 % MFMA utilisation (rocprofv3 SQ counters of the first version of this kernel on up4.0; a single wave per SIMD is then
// issue-bound at 44 %).  A fragment x_hat[(d', h'), kw .. kw+15] is the B operand of EVERY tap (kd, kh, kw) whose output row (d' - kd, h' - kh)
// lies in the tile -- up to nine MFMAs with nine different dY rows and accumulators.  So a wave owns one kw and seven of the nine (kd, kh) pairs
// (tap groups 0-2; group 3 takes the two left-over pairs for all three kw: 7 + 7 + 7 + 6 = 27 taps), keeps the dY fragments of its rows resident
// in registers (loaded once per tile, plane by plane), walks the halo rows and issues all MFMAs a fragment feeds: 34 + 16 fragment reads for
// 112 MFMAs = 0.9 LDS instructions per MFMA.  The 8 waves of a block are (tap group) x (32-row group of dY) for 64-row blocks, or (tap group) x
// (depth half of the tile) for 32-row blocks, whose two partial sums meet in LDS once at the end of the block.
//
// Replaces the autograd weight-gradient of nn.Conv3d(k=3) (rsuper_train/model/dim3/conv_layers.py:29-38 under loss.backward(), train_ddp.py:349).
#include "common.hpp"
#include "kernels.hpp"
#include "wgrad_frag.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TD = 4, TH = 4, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int XROWS = HD * HH * HW;                              // 648 halo rows of 64 B (32 channels)
constexpr int XPIECES = (XROWS + 15) / 16;                       // 41 DMA pieces of 16 rows (the last one half padding)
constexpr int XBYTES = XPIECES * 1024;
constexpr int YPLANE = 256 * 64;                                 // one 32-row group of dY: 256 voxels x 64 B
constexpr int NW = 8;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t lds_byte) {
    // M0 = wave-uniform LDS byte address of lane 0's 16 bytes; written in the statement that uses it (the compiler does not model it)
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
template <int MT, int BD>
__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(WgradParams p) {
    constexpr int NDL = MT == 2 ? 4 : 2, NROWS = NDL * TH;
    constexpr int BUF = XBYTES + MT * YPLANE;
    constexpr int XK = (XPIECES + NW - 1) / NW;                  // x pieces per wave (6; 5 for the last wave)
    constexpr int YK = 16 * MT / NW;                             // dY pieces per wave (4 / 2)
    constexpr int NP = XK + YK;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsel = wave & 1, g = wave >> 1;
    const int dsel = MT == 1 ? wsel * 2 : 0;                     // first tile plane of this wave
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int m0 = blockIdx.y * MT * 32;

    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d * p.N;

    // fragment bases (lane part folded in), relative to a tile buffer
    const int x_off = dsel * (HH * HW * 64) + frag_lane_off<1>(64, lane);
    const int ya_off = XBYTES + (MT == 2 ? wsel * YPLANE : 0) + dsel * (TH * TW * 64) + frag_lane_off<1>(64, lane);

    // ---- DMA side: per-lane constants of this wave's pieces.  A piece = 16 consecutive 64-byte LDS rows; lane -> (row = lane >> 2, 16-byte slot = lane & 3).
    const int prow = lane >> 2, pslot = lane & 3;
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    const uint32_t xrowb = (uint32_t)xs.ld * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xs.x, 0, nvox_total * xrowb, 0x00020000);
    uint32_t xoffb[XK], xpm[XK];                                 // byte offset of the lane's halo row relative to the halo origin; one-hot (hd | hh << 6 | hw << 12), bit 31: no data
    const bool x_cok = c0 + pslot * 8 < xs.C;
#pragma unroll
    for (int k = 0; k < XK; ++k) {
        const int r = (wave + NW * k) * 16 + prow;
        const int hd = r / (HH * HW), rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        xoffb[k] = (uint32_t)((hd * p.H + hh) * p.W + hw) * xrowb + (uint32_t)(c0 + pslot * 8) * 2u;
        xpm[k] = (r < XROWS && x_cok) ? ((1u << hd) | (1u << (6 + hh)) | (1u << (12 + hw))) : (1u << 31);
    }
    // dY planes: plane q = rows m0 + 32 q .. + 31 of [ya | yb] (a 32-row group never straddles the two sources: checked by the launcher)
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

    struct IssueTile { int base, ybase, d0, h0, w0; uint32_t xbad; };
    auto prepare = [&](int tile) {
        IssueTile t;
        int q = tile;
        const int tw = q % tiles_w; q /= tiles_w;
        const int th = q % tiles_h; q /= tiles_h;
        const int td = q % tiles_d; q /= tiles_d;
        const int n = q;
        t.d0 = td * TD; t.h0 = th * TH; t.w0 = tw * TW;
        auto range = [](int o, int len, int nh) {                // bits i in [0, nh) with 0 <= o + i < len
            const int lo = o >= 0 ? 0 : -o;
            int hi_ = len - 1 - o; if (hi_ > nh - 1) hi_ = nh - 1;
            return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
        };
        t.xbad = ~(range(t.d0 - 1, p.D, HD) | (range(t.h0 - 1, p.H, HH) << 6) | (range(t.w0 - 1, p.W, HW) << 12));
        t.base = ((n * p.D + t.d0 - 1) * p.H + (t.h0 - 1)) * p.W + (t.w0 - 1);
        t.ybase = ((n * p.D + t.d0) * p.H + t.h0) * p.W + t.w0;
        return t;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic region
    auto issue_x = [&](const IssueTile& t, int k, uint32_t buf) {
        const int piece = wave + NW * k;
        if (piece >= XPIECES) return;                            // wave-uniform
        const bool ok = (xpm[k] & t.xbad) == 0u;
        const uint32_t voff = ok ? (uint32_t)t.base * xrowb + xoffb[k] : 0xFFFFFFF0u;       // out of range -> the DMA writes zeros
        dma16(xrs, voff, lds0 + buf + (uint32_t)piece * 1024u);
    };
    auto issue_y = [&](const IssueTile& t, int k, uint32_t buf) {
        const int y = wave + NW * k;                             // plane q, line (d, h) of the tile: 16 voxels along w x 64 B
        const int q = MT == 1 ? 0 : (y >> 4), line = y & 15, d = line >> 2, h = line & 3;
        const bool ok = yok[q] && t.d0 + d < p.D && t.h0 + h < p.H && t.w0 + prow < p.W;
        const uint32_t voff = ok ? (uint32_t)(t.ybase + (d * p.H + h) * p.W) * yrowb[q] + yoffb[q] : 0xFFFFFFF0u;
        dma16(MT == 1 ? yrs[0] : (q ? yrs[MT - 1] : yrs[0]), voff, lds0 + buf + XBYTES + (uint32_t)y * 1024u);
    };
    auto issue_piece = [&](const IssueTile& t, int j, uint32_t buf) {      // j static
        if (j < XK) issue_x(t, j, buf); else issue_y(t, j - XK, buf);
    };

    // XCD-aware tile order (see conv3d_wgrad.hip): class z & 7 owns a contiguous range of tiles and its blocks sweep it together
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

    auto run = [&](auto G_) {
        constexpr int G = std::remove_reference_t<decltype(G_)>::value;
        constexpr Sched S = make_sched(G, NDL);
        constexpr int KDMIN = G < 3 ? 0 : 2;                     // first halo plane that uses dY plane 0
        f32x16_t acc[7];
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        if (tile0 < tile_end) {
            const IssueTile t = prepare(tile0);
#pragma unroll
            for (int j = 0; j < NP; ++j) issue_piece(t, j, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

        int it = 0;
        for (int tile = tile0; tile < tile_end; tile += tstride, ++it) {
            const uint32_t cur = (uint32_t)(it & 1) * BUF, nxt = (uint32_t)((it + 1) & 1) * BUF;
            const bool has_next = tile + tstride < tile_end;
            const IssueTile nt = prepare(has_next ? tile + tstride : tile);
            const char* buf = smem + cur;
            auto fetch_a = [&](int row) { return frag_bf16<1>(buf + ya_off + (row * TW) * 64, 64); };
            auto fetch_b = [&](int s) { return frag_bf16<1>(buf + x_off + ((S.dp[s] * HH + S.hp[s]) * HW + S.kw[s]) * 64, 64); };
            constexpr int BR = BD + 1;
            uint4 a[NROWS], bq[BR];
#pragma unroll
            for (int h = 0; h < TH; ++h) a[h] = fetch_a(h);
#pragma unroll
            for (int s = 0; s < BD; ++s) bq[s] = fetch_b(s);
#pragma unroll
            for (int s = 0; s < S.n; ++s) {
                if (s + BD < S.n) bq[(s + BD) % BR] = fetch_b(s + BD);
                // dY plane d is first used on halo plane d + KDMIN: request it one halo plane earlier
                if ((s == 0 || S.dp[s] != S.dp[s - 1]) && S.dp[s] - KDMIN + 1 >= 1 && S.dp[s] - KDMIN + 1 < NDL) {
#pragma unroll
                    for (int h = 0; h < TH; ++h) a[(S.dp[s] - KDMIN + 1) * TH + h] = fetch_a((S.dp[s] - KDMIN + 1) * TH + h);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pidx = 0; pidx < 9; ++pidx) {
                    if (!pair_in_group(G, pidx) || !unit_valid(NDL, S.dp[s], S.hp[s], pidx)) continue;
                    const int d = S.dp[s] - pidx / 3, h = S.hp[s] - pidx % 3;
                    const int slot = G < 3 ? pidx : (pidx - 7) * 3 + S.kw[s];
#ifndef WGD_SKIP_MMA                                             // ablation switches (tools/wgd_ablate.sh): the DMA stream alone / the MFMA loop alone
                    mma32<bf16_t>(acc[slot], a[d * TH + h], bq[s % BR]);
#else
                    if (s == 0) mma32<bf16_t>(acc[slot], a[d * TH + h], bq[s % BR]);
#endif
                }
#ifndef WGD_SKIP_DMA
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (s == (j * S.n) / NP && has_next) issue_piece(nt, j, nxt);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // tile + 1 has landed (my pieces); everybody is done with `cur`
        }

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
            float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * cin_total;
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

template <int MT, int BD>
int launch_dma(const WgradParams& p, hipStream_t st) {
    constexpr int smem = 2 * (XBYTES + MT * YPLANE);
    static_assert(MT == 2 || 4 * 7 * 16 * 64 * 4 <= smem, "depth-half reduction scratch must fit the tile buffers");
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    dim3 grid(nch, (Mtot + MT * 32 - 1) / (MT * 32), p.splits), block(512);
    auto k = wgrad_dma_kernel<MT, BD>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    return RS_OK;
}

}  // namespace

// bf16, both x sources pre-normalised (no statistics), 32-row groups of dY inside one source each
bool rs_wgrad_dma_supported(const WgradParams& p, int dtype) {
    static const int off = getenv("RSUPER_WGRAD_DMA") ? atoi(getenv("RSUPER_WGRAD_DMA")) == 0 : 0;
    if (off || dtype != RS_BF16 || p.xa.mr || (p.xb.C > 0 && p.xb.mr)) return false;
    if (p.yb.C > 0 && (p.ya.C % 32)) return false;
    return true;
}

int rs_launch_wgrad_dma(const WgradParams& p, hipStream_t st) {
    const int Mtot = p.ya.C + p.yb.C;
#ifndef WGD_BD
#define WGD_BD 3
#endif
    return Mtot <= 32 ? launch_dma<1, WGD_BD>(p, st) : launch_dma<2, WGD_BD>(p, st);
}
