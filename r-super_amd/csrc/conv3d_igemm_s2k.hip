// 3x3x3 / STRIDE 2 / pad 1 convolution, forward, as an implicit GEMM on MFMA (gfx950, bf16, channels-last): the persistent kernel for the strided
// [conv1 | shortcut] GEMM of down_block(pool=False) -- BasicBlock(in, out, stride=2), rsuper_train/model/dim3/unet_utils.py:38-39,
// conv_layers.py:29-38,82-84.  Same operation, arguments, fused InstanceNorm + ReLU prologue and statistics epilogue as the forward mode of
// conv3d_igemm_s2.hip (which stays the f32 / data-gradient kernel and the A/B reference); what differs is how the work is laid out:
//
//   * the parity-class kernel stages one (class, 32-channel chunk) brick at a time -- 3.4 taps of MFMA work per staged brick on average, a 2.5x halo, the
//     input read once per 64-column group, one 4-wave block per CU (454 registers) with staging, barrier and MFMA phases in series: 262 TF on down1.0.
//   * here a stride-2 output tile is 4 x TH x 16 voxels of the half grid and an ITEM is (depth tap td, 16 input channels): the four input planes
//     2 (d0 + od) + td - 1 it needs, (2 TH + 1) x 33 rows each, 48-byte pitch, rows of one parity in w stored together -- the fragment of tap (th, tw)
//     is then 2 x 16 CONSECUTIVE LDS rows, the conflict-free ds_read_b128 pattern of the stride-1 kernels.  57 KB (TH 4), two buffers, ONE barrier per item;
//   * block = 8 matrix waves (two per SIMD) = (h pair, 32-column fragment): 128 columns x TH 4 or 256 columns x TH 2 per block -- the input is staged once
//     for all columns of [conv1 | shortcut]; a wave holds the four depth planes of its h pair (4 accumulators) and multiplies each weight fragment four
//     times: 1 LDS read + 0.25 weight loads per MFMA (a strided tap shares no operand with its neighbours: no depth re-use as in conv3d_igemm_kd.hip);
//   * everything else is conv3d_igemm_kd.hip's machinery: persistent blocks over the tiles of a sample in XCD-aware order, tile descriptors in LDS,
//     the eight waves stage the next item from hooks inside the MFMA loop (norm + ReLU in registers, the loads of the item after that in flight),
//     progress-based wave priority, wave-private epilogue through LDS scratch with the statistics accumulated in registers (one partial row per
//     (block, h pair)).
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TD = 4, TW = 16, HWC = 2 * TW + 1;    // tile depth / width on the half grid; staged input columns of a line
constexpr int PITCH = 48;                           // 32 B of data (16 bf16 channels) + 16 B: odd multiple of 16 -> conflict-free ds_read_b128
constexpr int NT = 512, NW = 8;
constexpr int SCR_ROW = 36;                         // floats per epilogue scratch row
constexpr int SCR_BYTES = 32 * SCR_ROW * 4;         // per wave
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int NCF> struct Geo {
    static constexpr int NMG = 8 / NCF;             // h pairs of a tile (waves = NMG x NCF)
    static constexpr int TH = 2 * NMG;
    static constexpr int HL = 2 * TH + 1;           // staged input lines of a plane
    static constexpr int HROWS = TD * HL * HWC;     // 1188 (NCF 4) / 660 (NCF 8)
    static constexpr int HB = ((HROWS * PITCH + 1023) / 1024) * 1024;
    static constexpr int NV = (HROWS * 2 + NT - 1) / NT;   // 16-byte staging vectors per thread and item
};

__device__ __forceinline__ void row_to_hw_nt(int i, int& hs, int& w) {     // row_to_hw (common.hpp) without branches (conv3d_igemm_kd.hip)
    hs = (int)((0xF00F0FF0u >> i) & 1u);
    const unsigned long long t = i < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
    w = (int)((t >> ((i & 15) * 4)) & 15ull);
}

struct Item { int c, td; uint32_t wofs; };
struct Tile { uint32_t base, bad0, bad1, org; };

// p.D/H/W = the half-resolution grid, FD/FH/FW the full-resolution one.  One normalised source (p.a), no residual.
template <int NCF>
__global__ __launch_bounds__(NT, 2) void igemm_s2k_kernel(IgemmParams p, int FD, int FH, int FW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Geo<NCF> G;
    constexpr int NMG = G::NMG, TH = G::TH, HL = G::HL, HROWS = G::HROWS, HB = G::HB, NV = G::NV;
    auto U = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto UP = [](const void* q) {
        const uint64_t a = (uint64_t)q;
        return (const void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    };
    const int Ca = U(p.a.C), lda = U(p.a.ld);
    const void* const xa = UP(p.a.x);
    const float* const mra = (const float*)UP(p.a.mr);
    const int pD = p.D, pH = p.H, pW = p.W, pN = p.N, Cout = p.Cout, ldo = p.ldo, ntiles = p.ntiles;
    const void* const wpk = p.wp; void* const outp = p.out; float* const partp = p.part;
    char* bufs = smem;                                                  // 2 x HB
    float4* ntab = (float4*)(smem + 2 * HB);                            // [Ca / 2] (sc0, sc1, nb0, nb1)
    char* scr_base = smem + 2 * HB + Ca * 8;
    uint4* dtab = (uint4*)(scr_base + NW * SCR_BYTES);                  // tile descriptors of this block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hp = wave % NMG, cf = wave / NMG;
    const int n = blockIdx.z;
    const int tiles_w = (pW + TW - 1) / TW, tiles_h = (pH + TH - 1) / TH, tiles_d = (pD + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int gx = (int)gridDim.x;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / gx + 1 : 0;
    const int nk = (Ca + 15) / 16, nit = 3 * nk;                        // items of a tile: depth tap outer, 16-channel slice inner (same cache lines back to back)
    const int nitems = my_tiles * nit;
    const uint32_t tapstride = (uint32_t)ntiles * 2048u;                // bytes between consecutive taps of one (chunk, k-step) in the packed weights
    const uint32_t nvox_src = (uint32_t)(pN * FD * FH * FW), nvox_out = (uint32_t)(pN * pD * pH * pW);
    const uint32_t rowb = (uint32_t)lda * 2u, nrec = nvox_src * rowb;
    const uint32_t plane = (uint32_t)(FH * FW);

    // ---- per-block tables
    for (int i = tid; i < Ca / 2; i += NT) {
        const float* m = mra + ((size_t)n * Ca + 2 * i) * 2;
        ntab[i] = make_float4(m[1], m[3], -m[0] * m[1], -m[2] * m[3]);
    }
    {
        // entry k = k-th tile of this block, XCD-aware order (linear workgroup id b runs on XCD b % 8; every XCD gets a contiguous run of tiles):
        // (voxel of the staged origin (2 d0 - 1, 2 h0 - 1, 2 w0 - 1), ~valid lines << 4 | ~valid column 32 << 13 | bit 31, ~valid columns 0..31, d0 | h0 << 10 | w0 << 20);
        // the planes' validity depends on the depth tap and is derived per item.  Entries past the last tile describe "nothing to load".
        const bool xcd_remap = (gx & 7) == 0 && tiles >= 64;
        for (int k = tid; k < my_tiles + 3; k += NT) {
            const bool live = k < my_tiles;
            int t = live ? (int)blockIdx.x + k * gx : 0;
            if (xcd_remap) {
                const int q = tiles >> 3, r = tiles & 7, xcd = t & 7, kk = t >> 3;
                t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
            }
            const int tw = t % tiles_w; t /= tiles_w;
            const int th = t % tiles_h; t /= tiles_h;
            const int d0 = t * TD, h0 = th * TH, w0 = tw * TW;
            uint32_t badl = 0, badw = 0, badw32 = 0;
            for (int i = 0; i < HL; ++i) badl |= ((unsigned)(2 * h0 - 1 + i) >= (unsigned)FH) ? (1u << i) : 0u;
            for (int i = 0; i < 32; ++i) badw |= ((unsigned)(2 * w0 - 1 + i) >= (unsigned)FW) ? (1u << i) : 0u;
            badw32 = ((unsigned)(2 * w0 + 31) >= (unsigned)FW) ? 1u : 0u;
            dtab[k] = make_uint4((uint32_t)(((n * FD + 2 * d0 - 1) * FH + 2 * h0 - 1) * FW + 2 * w0 - 1),
                                 live ? ((badl << 4) | (badw32 << 13) | 0x80000000u) : 0x80003FFFu, live ? badw : 0xFFFFFFFFu,
                                 (uint32_t)(d0 | (h0 << 10) | (w0 << 20)));
        }
    }
    auto fetch_tile = [&](int k) {                                      // wave-uniform: one broadcast LDS read + readfirstlanes
        const uint4 v = dtab[k];
        Tile t;
        t.base = __builtin_amdgcn_readfirstlane(v.x); t.bad0 = __builtin_amdgcn_readfirstlane(v.y);
        t.bad1 = __builtin_amdgcn_readfirstlane(v.z); t.org = __builtin_amdgcn_readfirstlane(v.w);
        return t;
    };
    auto item_of = [&](int j) {                                         // j-th item of a tile
        Item it;
        it.td = j >= 2 * nk ? 2 : (j >= nk ? 1 : 0);
        const int jj = j - it.td * nk;
        it.c = jj * 16;
        it.wofs = (uint32_t)(((jj >> 1) * 54 + (jj & 1)) * ntiles) * 1024u + (uint32_t)(it.td * 9) * tapstride;
        return it;
    };
    auto planes_bad = [&](const Tile& t, int td) {                      // bits 0..3: input plane 2 (d0 + od) + td - 1 outside the volume
        const int d0 = (int)(t.org & 1023u);
        uint32_t b = 0;
#pragma unroll
        for (int od = 0; od < TD; ++od) b |= ((unsigned)(2 * (d0 + od) + td - 1) >= (unsigned)FD) ? (1u << od) : 0u;
        return b;
    };

    // ---- staging through registers: thread -> 16-byte slot tid & 1 of staged rows perm(tid >> 1) + 256 i (inside every run of 8 rows the order is
    //      0,2,4,6,1,3,5,7: conflict-free ds_write_b128 at the 48-byte pitch, conv3d_igemm_kd.hip).  Staged row r = (plane od, line l, column c);
    //      columns 0..16 are the even positions 2 c of a line (taps tw 0 / 2), columns 17..32 the odd positions 2 (c - 17) + 1 (tap tw 1).
    const int s_slot = tid & 1, rk = tid >> 1;
    const int row0 = (rk & ~7) | ((rk & 3) << 1) | ((rk >> 2) & 1);
    int xvo[NV];
    uint32_t pm0[NV], pm1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int r = row0 + 256 * i;
        const int od = r / (HL * HWC), rem = r - od * (HL * HWC);
        const int l = rem / HWC, c = rem - l * HWC;
        const int rw = c < 17 ? 2 * c : 2 * (c - 17) + 1;
        xvo[i] = (od * 2 * FH + l) * FW + rw;
        pm0[i] = r < HROWS ? ((1u << od) | (1u << (4 + l)) | (rw == 32 ? (1u << 13) : 0u)) : 0x80000000u;
        pm1[i] = (r < HROWS && rw < 32) ? (1u << rw) : 0u;
    }
    const int x_st = row0 * PITCH + s_slot * 16;                         // LDS byte of vector 0; vector i at + 12288 i
    uint4 px[NV];
    uint32_t pvm = 0;                                                    // validity bits of the vectors held in px
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xa, 0, nrec, 0x00020000);
    auto issue_v = [&](const Tile& t, uint32_t tbad0, const Item& it, int i) {   // i static: load vector i of (tile, item) into px[i]
        const bool ok = ((pm0[i] & tbad0) | (pm1[i] & t.bad1)) == 0u && it.c + s_slot * 8 < Ca;
        const uint32_t off = ok ? __umul24(t.base + (uint32_t)it.td * plane + (uint32_t)xvo[i], rowb) + (uint32_t)(s_slot * 16) : 0xFFFFFFF0u;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, it.c * 2, 0);
        px[i] = make_uint4(q[0], q[1], q[2], q[3]);
        pvm = ok ? (pvm | (1u << i)) : (pvm & ~(1u << i));
    };
    float4 ncst[4];                                                      // constants of the item held in px (this thread's 8 channels)
    auto load_norm = [&](const Item& it) {
        const float4* row = ntab + (it.c >> 1) + s_slot * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) ncst[j] = row[j];
    };
    auto commit_w = [&](int i, int j) {                                  // word j of vector i normalised in place: 7 vector-ALU operations
        uint32_t* w = j == 0 ? &px[i].x : j == 1 ? &px[i].y : j == 2 ? &px[i].z : &px[i].w;
        const uint32_t m = ((pvm >> i) & 1u) ? 0xFFFFFFFFu : 0u;         // padding stays zero AFTER the activation
        const float4 c = ncst[j];
        float x0, x1;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(__uint_as_float(*w << 16)), "v"(c.x), "v"(c.z));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(__uint_as_float(*w & 0xffff0000u)), "v"(c.y), "v"(c.w));
        i16x2_t v = __builtin_bit_cast(i16x2_t, f2bf2(x0, x1));
        const i16x2_t z = {0, 0};
        v = __builtin_elementwise_max(v, z);
        *w = __builtin_bit_cast(uint32_t, v) & m;
    };
    auto commit_st = [&](char* buf, int i) {
        if (row0 + 256 * i < HROWS) *(uint4*)(buf + x_st + i * (256 * PITCH)) = px[i];
    };

    auto run = [&](auto LATE_) {                                         // LATE: the second wave of its SIMD runs its staging hooks one step later
    constexpr bool late = std::remove_reference_t<decltype(LATE_)>::value;
    constexpr int RB = 3;                                                // weight ring, in (th, tw) groups: 3 divides the 9 groups of an item
    const int ntile0 = blockIdx.y * NCF + cf;
    int hs, wl;
    row_to_hw_nt(lane & 31, hs, wl);
    const int a_lane = ((2 * (2 * hp + hs)) * HWC + wl) * PITCH + (lane >> 5) * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t lane16 = (uint32_t)lane * 16u + (uint32_t)ntile0 * 1024u;
    uint32_t ts_item = tapstride;                                        // re-laundered every item (keeps 9 tap offsets out of scalar registers)
    auto load_b = [&](uint32_t wofs, int g, uint4& dst) {                // g = th * 3 + tw (static)
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, wofs + (uint32_t)g * ts_item, 0);
        dst = make_uint4(q[0], q[1], q[2], q[3]);
    };
    // epilogue geometry of this wave: lane -> 16-byte column group cg of rows er0, er0 + 16 of a fragment
    const int cg = lane & 3, er0 = lane >> 2;
    int rhs[2], rw_[2];
    row_to_hw_nt(er0, rhs[0], rw_[0]);
    row_to_hw_nt(er0 + 16, rhs[1], rw_[1]);
    float* scr = (float*)(scr_base + wave * SCR_BYTES);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nvox_out * (uint32_t)ldo * 2u, 0x00020000);

    f32x16_t acc[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    f32x2_t rs1[4], rs2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { rs1[q] = f32x2_t{0.f, 0.f}; rs2[q] = f32x2_t{0.f, 0.f}; }
    const bool wave_live = ntile0 * 32 < Cout;

    // ---- prologue: item 0 staged synchronously into buffer 0, item 1 in flight in registers
    __syncthreads();                                                     // tables
    int k1 = 0, j1 = 0;                                                  // (tile, item) of the NEXT item (the one held in px)
    Tile t1 = fetch_tile(0);
    Item i1 = item_of(0);
    uint32_t b1 = t1.bad0 | planes_bad(t1, i1.td);
#pragma unroll
    for (int i = 0; i < NV; ++i) issue_v(t1, b1, i1, i);
    load_norm(i1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) commit_w(i, j);
        commit_st(bufs, i);
    }
    if (++j1 == nit) { j1 = 0; ++k1; t1 = fetch_tile(k1); }
    i1 = item_of(j1);
    b1 = t1.bad0 | planes_bad(t1, i1.td);
#pragma unroll
    for (int i = 0; i < NV; ++i) issue_v(t1, b1, i1, i);
    uint4 bq[RB];
    uint32_t wofs_cur = item_of(0).wofs;
#pragma unroll
    for (int g = 0; g < RB - 1; ++g) load_b(wofs_cur, g, bq[g]);
    __syncthreads();

    int kc = 0, jc = 0;                                                  // (tile, item) of the current item
    Tile tc = fetch_tile(0);
    for (int it = 0; it < nitems; ++it) {
        const char* buf = bufs + (it & 1) * HB;
        char* nxt = bufs + ((it + 1) & 1) * HB;
        const bool last = jc == nit - 1;                                 // this item completes a tile
        asm volatile("" : "+s"(ts_item));
        // item it + 1 sits in px (tile t1, item i1), item it + 2 is requested by the hooks
        int k2 = k1, j2 = j1 + 1;
        if (j2 == nit) { j2 = 0; ++k2; }
        Tile t2 = t1;
        if (j2 == 0) t2 = fetch_tile(k2);
        const Item i2 = item_of(j2);
        const uint32_t b2 = t2.bad0 | planes_bad(t2, i2.td);
        const uint32_t wofs_next = i1.wofs;                              // weights of the next item: the ring runs across the barrier
        load_norm(i1);
        auto fetch_a = [&](int g, int od) {                              // static: group g = th * 3 + tw, output plane od
            const int th = g / 3, tw = g % 3;
            return *(const uint4*)(buf + a_lane + ((od * HL + th) * HWC + (tw == 1 ? 17 : tw == 2 ? 1 : 0)) * PITCH);
        };
        constexpr int AD = 5, AR = AD + 1;                               // fragment ring / prefetch distance
        constexpr int NS = 9 * TD;                                       // MFMA steps of an item
        uint4 aq[AR];
        if (!wave_live) {                                                // a dead wave only stages its share and meets the others at the barrier
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) commit_w(i, j);
                commit_st(nxt, i);
                issue_v(t2, b2, i2, i);
            }
        } else {
#pragma unroll
            for (int s = 0; s < AD; ++s) aq[s % AR] = fetch_a(s / TD, s % TD);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                // progress-based priority: the wave of a SIMD that is behind in the item outranks its partner (conv3d_igemm_kd.hip)
                if (g == 0) __builtin_amdgcn_s_setprio(3);
                if (g == 2) __builtin_amdgcn_s_setprio(2);
                if (g == 4) __builtin_amdgcn_s_setprio(1);
                if (g == 6) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int od = 0; od < TD; ++od) {
                    const int s = g * TD + od;
                    if (s + AD < NS) aq[(s + AD) % AR] = fetch_a((s + AD) / TD, (s + AD) % TD);
                    if (od == 0) {                                       // weights of group g + RB - 1 (possibly of the next item)
                        const int gn = g + RB - 1;
                        if (gn < 9) load_b(wofs_cur, gn, bq[gn % RB]);
                        else load_b(wofs_next, gn - 9, bq[gn % RB]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma32<bf16_t>(acc[od], aq[s % AR], bq[g % RB]);
                    // hooks: 5 pieces per staging vector (4 words + store / next load) spread over the steps; the two waves of a SIMD one step apart
#pragma unroll
                    for (int i = 0; i < NV; ++i)
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            const int at0 = ((i * 5 + j) * NS) / (NV * 5);
                            const int at = at0 + 1 < NS ? at0 + 1 : NS - 1;
                            if (s == (late ? at : at0)) {
                                if (j < 4) commit_w(i, j);
                                else { commit_st(nxt, i); issue_v(t2, b2, i2, i); }
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (last) {
                // -------------------------------------------------------------- wave-private epilogue of this tile (conv3d_igemm_kd.hip, forward)
                const int hi = lane >> 5, col_l = lane & 31;
                const int d0 = tc.org & 1023, h0 = (tc.org >> 10) & 1023, w0 = (int)(tc.org >> 20);
                const int col0 = ntile0 * 32 + cg * 8;                   // first output column of this lane's vectors
#pragma unroll
                for (int d = 0; d < TD; ++d) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * SCR_ROW + col_l] = acc[d][r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = er0 + 16 * ps;
                        const float4* sp = (const float4*)(scr + row * SCR_ROW + cg * 8);
                        const float4 ta = sp[0], tb = sp[1];
                        const f32x2_t v2[4] = {{ta.x, ta.y}, {ta.z, ta.w}, {tb.x, tb.y}, {tb.z, tb.w}};
                        const int h = h0 + 2 * hp + rhs[ps], w = w0 + rw_[ps];
                        const uint32_t vx = (uint32_t)(((n * pD + d0 + d) * pH + h) * pW + w);
                        const bool ok = h < pH && w < pW && d0 + d < pD && col0 < Cout;
                        uint32_t ow[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a0 = ok ? v2[q][0] : 0.f, a1 = ok ? v2[q][1] : 0.f;
                            const uint32_t wv = f2bf2(a0, a1);
                            ow[q] = wv;
                            const f32x2_t r = {__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u)};
                            rs1[q] = rs1[q] + r;
                            rs2[q] = __builtin_elementwise_fma(r, r, rs2[q]);
                        }
                        const u32x4_t pk = {ow[0], ow[1], ow[2], ow[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ors, ok ? (vx * (uint32_t)ldo + (uint32_t)col0) * 2u : 0xFFFFFFF0u, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                                                 // item it consumed, item it + 1 complete in the other buffer
        if (++jc == nit) { jc = 0; ++kc; tc = fetch_tile(kc); }
        wofs_cur = wofs_next;
        t1 = t2; i1 = i2;
        k1 = k2; j1 = j2;
    }
    // statistics: ONE partial row per (block, h pair); this wave's columns
    if (partp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
                rs1[q][0] += __shfl_xor(rs1[q][0], o, 64); rs1[q][1] += __shfl_xor(rs1[q][1], o, 64);
                rs2[q][0] += __shfl_xor(rs2[q][0], o, 64); rs2[q][1] += __shfl_xor(rs2[q][1], o, 64);
            }
        }
        if (lane < 4) {
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(partp, 0, 0x7FFFFFFF, 0x00020000);
            const int col0 = ntile0 * 32 + cg * 8;
            const uint32_t poff = col0 < Cout ? (uint32_t)(((((size_t)n * gx + blockIdx.x) * NMG + hp) * Cout + col0) * 8) : 0xFFFFFFF0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4_t pv;
                pv[0] = __float_as_uint(rs1[q][0]); pv[1] = __float_as_uint(rs2[q][0]); pv[2] = __float_as_uint(rs1[q][1]); pv[3] = __float_as_uint(rs2[q][1]);
                __builtin_amdgcn_raw_buffer_store_b128(pv, prs, poff == 0xFFFFFFF0u ? poff : poff + q * 16, 0, 0);
            }
        }
    }
    };
    if (wave >= 4) run(std::true_type{});
    else run(std::false_type{});
}

// column fragments of a block: 4 (128 columns, tile 4 x 4 x 16) unless the GEMM is wider AND the half grid is small -- 256-column blocks stage the
// input once for twice the columns, on tiles of half the size (more blocks for the chip)
int s2k_ncf(int n_cols) { return n_cols > 128 ? 8 : 4; }
int s2k_tiles(int ncf, int D, int H, int W) {
    const int th = 2 * (8 / ncf);
    return ((D + TD - 1) / TD) * ((H + th - 1) / th) * ((W + TW - 1) / TW);
}
int s2k_grid_x(int tiles, int gy, int N) {                               // ~one persistent block per CU
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}

template <int NCF>
int launch_s2k(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st) {
    typedef Geo<NCF> G;
    const int tiles = s2k_tiles(NCF, p.D, p.H, p.W);
    const int gy = (p.ntiles + NCF - 1) / NCF;
    const int gx = s2k_grid_x(tiles, gy, p.N);
    const size_t smem = 2 * (size_t)G::HB + (size_t)p.a.C * 8 + NW * (size_t)SCR_BYTES + ((tiles + gx - 1) / gx + 4) * 16;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    auto k = igemm_s2k_kernel<NCF>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(gx, gy, p.N), dim3(NT), smem, st, p, FD, FH, FW);
    return rs_check_launch();
}

}  // namespace

// forward, bf16, one normalised source whose channels are a multiple of 16, even half-grid-independent full-resolution sizes are NOT required
bool rs_igemm_s2k_supported(const IgemmParams& p, int dtype, int FD, int FH, int FW) {
    if (dtype != RS_BF16 || !p.a.mr || p.b.C != 0 || p.res) return false;
    if ((p.a.C % 16) || p.a.C < 16 || p.a.C > 512) return false;
    if (p.D > 1020 || p.H > 1020 || p.W > 4000) return false;
    if ((unsigned long long)p.N * FD * FH * FW >= (1ull << 24)) return false;   // __umul24 of the voxel index
    const int ncf = s2k_ncf(p.Cout);
    const int tiles = s2k_tiles(ncf, p.D, p.H, p.W);
    const int gy = (p.ntiles + ncf - 1) / ncf;
    const int gx = s2k_grid_x(tiles, gy, p.N);
    const size_t hb = ncf == 4 ? Geo<4>::HB : Geo<8>::HB;
    return 2 * hb + (size_t)p.a.C * 8 + NW * (size_t)SCR_BYTES + ((tiles + gx - 1) / gx + 4) * 16 <= 160 * 1024;
}

int rs_igemm_s2k_part_rows(int ntiles, int n_cols, int N, int D, int H, int W) {
    const int ncf = s2k_ncf(n_cols);
    const int gy = (ntiles + ncf - 1) / ncf;
    return s2k_grid_x(s2k_tiles(ncf, D, H, W), gy, N) * (8 / ncf);
}

int rs_launch_igemm_s2k(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st) {
    return s2k_ncf(p.Cout) == 4 ? launch_s2k<4>(p, FD, FH, FW, st) : launch_s2k<8>(p, FD, FH, FW, st);
}
