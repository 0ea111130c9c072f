// 3x3x3 / stride 1 / pad 1 convolution as an implicit GEMM on MFMA (gfx950, bf16, channels-last): the depth-reuse ("kd") kernel for the
// wide full-resolution layers (64 .. 128 output columns at 96^3 / 48^3: up4.0, up3.0, the 64 -> 64 blocks; forward and data gradient).
//
// Same operation, arguments, fused prologue and epilogues as conv3d_igemm.hip (rsuper_train/model/dim3/conv_layers.py:29-51 ConvNormAct
// inside BasicBlock :86-94): x_hat = relu((x - mean) * rstd) while the halo is staged, two sources instead of a concat (unet_utils.py:71),
// forward epilogue (+ residual, statistics of the output), data-gradient epilogue (ReLU mask, InstanceNorm-backward sums).
//
// What is different: the operand traffic per MFMA.  tools/ubench/mfma_lds.hip: one ds_read_b128 per MFMA caps the matrix
// pipe at 1.45 PF, one per two MFMAs at 1.88 PF -- and the tilings of conv3d_igemm.hip read 1.0 (64 columns) .. 1.5 (32 columns) fragments
// per MFMA, their consumers alone run at 1.5 PF.  Here
//   * block = 8 matrix waves (two per SIMD), output tile 4 x 8 x 16 voxels (512 = 16 M fragments of (d, h pair) x 16 w) x up to 128 columns;
//     wave = (h pair hp, column half): ALL FOUR depth planes of its h pair x NF column fragments (4 x NF accumulators);
//   * an activation fragment (halo plane d', rows 2 hp + kh, shift kw) is the operand of the three taps kd with 0 <= d' - kd < 4: per (kh, kw)
//     group 6 fragment reads + 3 NF weight fragments feed 12 NF MFMAs -> 0.5 / NF LDS reads + 0.25 weight loads per MFMA;
//   * K is staged 16 channels (one MFMA k-step) at a time: halo 6 x 10 x 18 rows x 48-byte pitch = 51.8 KB, two buffers, ONE barrier per item;
//     the eight waves stage the next item themselves from hooks in the MFMA loop (norm + ReLU in registers, ds_write_b128), the loads of the
//     item after that in flight in registers -- the scheme of conv3d_wgrad2.hip;
//   * persistent blocks over the tiles of a sample (XCD-aware order), per-block tile descriptors in LDS (no index arithmetic in the loop),
//     wave-private epilogue through a 4.6 KB LDS scratch, statistics accumulated in registers over the block's tiles (one partial row per
//     (block, h pair)).
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <type_traits>
#include <stdio.h>

namespace {

constexpr int TD = 4, TH = 8, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HROWS = HD * HH * HW;                 // 1080 halo rows
constexpr int PITCH = 48;                           // 32 B of data (16 bf16 channels) + 16 B: odd multiple of 16 -> conflict-free ds_read_b128
constexpr int HALO_BYTES = HROWS * PITCH;           // 51840
constexpr int NT = 512, NW = 8;
constexpr int NV = 5;                               // 16-byte staging vectors per thread and item (2160 over 512 threads)
constexpr int SCR_ROW = 36;                         // floats per epilogue scratch row
constexpr int SCR_BYTES = 32 * SCR_ROW * 4;         // per wave
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// row_to_hw (common.hpp) without branches: the if-chain becomes a private-memory lookup table (scratch) otherwise
__device__ __forceinline__ void row_to_hw_nt(int i, int& hs, int& w) {
    hs = (int)((0xF00F0FF0u >> i) & 1u);
    const unsigned long long t = i < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
    w = (int)((t >> ((i & 15) * 4)) & 15ull);
}

#ifdef KD_PROF
__device__ unsigned long long g_kd_prof[8 * 4 + 2];                    // block (0, 0, 0): [wave][item-loop cycles, barrier cycles, epilogue cycles, items]
#endif

// one 16-channel slice of [a | b]: first channel, channels / row bytes / base / bytes of its source, table index of its constants, byte offset of its packed weights (tap 0)
struct Item { int c, C; uint32_t rowb, nrec; uint64_t base; int tabc; uint32_t wofs; };

// NF0 / NF1: 32-column fragments of the waves 0-3 / 4-7 (block = (NF0 + NF1) x 32 columns).  EPI: 0 forward, 1 data gradient, 2 forward + residual.
// NORM: both sources carry (mean, rstd).
template <int NF0, int NF1, int EPI, bool NORM>
__global__ __launch_bounds__(NT, 2) void igemm_kd_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN32 = NF0 + NF1;
    // every field is copied into a local once: a select between two fields of the kernel-argument struct can become a load from a computed address,
    // which moves the struct to scratch and turns everything read from it into per-lane values (waterfall loops around every buffer instruction)
    // (readfirstlane: the value becomes the result of an intrinsic instead of a load, so `cond ? Cb : Ca` cannot be rewritten into a table lookup)
    auto U = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto UP = [](const void* q) {
        const uint64_t a = (uint64_t)q;
        return (const void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    };
    const int Ca = U(p.a.C), Cb = U(p.b.C), lda = U(p.a.ld), ldb = U(p.b.ld);
    const void* const xa = UP(p.a.x); const void* const xb = UP(p.b.x);
    const float* const mra = (const float*)UP(p.a.mr); const float* const mrb = (const float*)UP(p.b.mr);
    const int eCa = U(p.ea.C), eCb = U(p.eb.C), elda = U(p.ea.ld), eldb = U(p.eb.ld);
    const void* const exa = UP(p.ea.x); const void* const exb = UP(p.eb.x);
    const float* const emra = (const float*)UP(p.ea.mr); const float* const emrb = (const float*)UP(p.eb.mr);
    const int pD = p.D, pH = p.H, pW = p.W, pN = p.N, Cout = p.Cout, ldo = p.ldo, ldr = p.ldr, ntiles = p.ntiles;
    const void* const wpk = p.wp; void* const outp = p.out; const void* const resp = p.res; float* const partp = p.part;
    const int ctot = Ca + Cb;
    char* bufs = smem;                                                  // 2 x HALO_BYTES
    float4* ntab = (float4*)(smem + 2 * HALO_BYTES);                    // [(Ca + Cb) / 2] (sc0, sc1, nb0, nb1)
    float* emr = (float*)(smem + 2 * HALO_BYTES + ctot * 8);            // [BN32 * 32][mean, rstd] of the epilogue source (EPI 1)
    char* scr_base = smem + 2 * HALO_BYTES + ctot * 8 + BN32 * 256;
    uint4* dtab = (uint4*)(scr_base + NW * SCR_BYTES);                  // tile descriptors of this block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hp = wave & 3, nhalf = wave >> 2;
    const int n = blockIdx.z;
    const int tiles_w = (pW + TW - 1) / TW, tiles_h = (pH + TH - 1) / TH, tiles_d = (pD + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int gx = (int)gridDim.x;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / gx + 1 : 0;
    const int nkA = (Ca + 15) / 16, nkB = (Cb + 15) / 16, nk = nkA + nkB;
    const int nchA = (Ca + 31) / 32;
    const int nitems = my_tiles * nk;
    const uint32_t tapstride = (uint32_t)ntiles * 2048u;              // bytes between consecutive taps of one (chunk, k-step) in the packed weights
    const uint32_t nvox_total = (uint32_t)(pN * pD * pH * pW);
    const uint32_t rowbA = (uint32_t)lda * 2u, rowbB = (uint32_t)ldb * 2u;
    const uint32_t nrecA = nvox_total * rowbA, nrecB = Cb ? nvox_total * rowbB : 0u;

    // ---- per-block tables
    if (NORM) {
        for (int i = tid; i < ctot / 2; i += NT) {
            const int c = 2 * i;
            const float* m = c < Ca ? mra + ((size_t)n * Ca + c) * 2 : mrb + ((size_t)n * Cb + c - Ca) * 2;
            ntab[i] = make_float4(m[1], m[3], -m[0] * m[1], -m[2] * m[3]);
        }
    }
    if (EPI == 1) {
        for (int i = tid; i < BN32 * 64; i += NT) {
            const int col = blockIdx.y * BN32 * 32 + (i >> 1);
            float v = (i & 1) ? 1.f : 0.f;
            if (col < Cout) v = col < eCa ? emra[((size_t)n * eCa + col) * 2 + (i & 1)] : emrb[((size_t)n * eCb + col - eCa) * 2 + (i & 1)];
            emr[i] = v;
        }
    }
    {
        // entry k = k-th tile of this block in the XCD-aware order of conv3d_igemm.hip (linear workgroup id b runs on XCD b % 8; every XCD gets a
        // contiguous run of tiles): (halo origin voxel, ~valid (hd | hh << 6) | bit 31, ~valid hw, d0 | h0 << 10 | w0 << 20); entries past the last tile
        // describe "nothing to load"
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
            auto range = [](int o, int len, int nh) {                   // bits i in [0, nh) with 0 <= o + i < len
                const int lo = o >= 0 ? 0 : -o;
                int hi_ = len - 1 - o; if (hi_ > nh - 1) hi_ = nh - 1;
                return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
            };
            const uint32_t ok0 = range(d0 - 1, pD, HD) | (range(h0 - 1, pH, HH) << 6), ok1 = range(w0 - 1, pW, HW);
            dtab[k] = make_uint4((uint32_t)(((n * pD + d0 - 1) * pH + h0 - 1) * pW + w0 - 1), live ? ((~ok0 & 0xFFFFu) | 0x80000000u) : 0x8000FFFFu,
                                 live ? (~ok1 & 0x3FFFFu) : 0x3FFFFu, (uint32_t)(d0 | (h0 << 10) | (w0 << 20)));
        }
    }
    struct Tile { uint32_t base, bad0, bad1, org; };
    auto fetch_tile = [&](int k) {                                      // wave-uniform: one broadcast LDS read + readfirstlanes
        const uint4 v = dtab[k];
        Tile t;
        t.base = __builtin_amdgcn_readfirstlane(v.x); t.bad0 = __builtin_amdgcn_readfirstlane(v.y);
        t.bad1 = __builtin_amdgcn_readfirstlane(v.z); t.org = __builtin_amdgcn_readfirstlane(v.w);
        return t;
    };
    auto item_of = [&](int j) {                                         // j-th 16-channel slice of [a | b]
        // selections between the two sources are mask arithmetic, not `cond ? b : a`: inside these by-reference lambdas a select of two captured variables
        // becomes a select of two ADDRESSES, the variables move to scratch and every use pays a flat load + vmcnt(0)
        Item it;
        const uint32_t mb = j >= nkA ? 0xFFFFFFFFu : 0u;
        const uint64_t mb64 = j >= nkA ? ~0ull : 0ull;
        const int jj = j - (int)((uint32_t)nkA & mb);
        it.c = jj * 16;
        it.C = (int)((uint32_t)Ca ^ (((uint32_t)Ca ^ (uint32_t)Cb) & mb));
        it.rowb = rowbA ^ ((rowbA ^ rowbB) & mb);
        it.nrec = nrecA ^ ((nrecA ^ nrecB) & mb);
        it.base = (uint64_t)xa ^ (((uint64_t)xa ^ (uint64_t)xb) & mb64);
        it.tabc = (int)((uint32_t)Ca & mb) + it.c;
        const int ch = (int)((uint32_t)nchA & mb) + (jj >> 1);
        it.wofs = (uint32_t)((ch * 54 + (jj & 1)) * ntiles) * 1024u;
        return it;
    };

    // ---- staging side: thread -> 16-byte slot tid & 1 of halo rows perm(tid >> 1) + 256 i.  Inside every run of 8 rows the order is 0,2,4,6,1,3,5,7: the
    //      eight lanes of a ds_write_b128 group then cover rows r, r+2, r+4, r+6 = dword offsets 12 r + {0, 24, 48, 72} + 0..7 -> 32 distinct banks
    const int s_slot = tid & 1, rk = tid >> 1;
    const int row0 = (rk & ~7) | ((rk & 3) << 1) | ((rk >> 2) & 1);
    int xvo[NV];
    uint32_t pm0[NV], pm1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int r = row0 + 256 * i;
        const int hd = r / (HH * HW), rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        xvo[i] = (hd * pH + hh) * pW + hw;
        pm0[i] = r < HROWS ? ((1u << hd) | (1u << (6 + hh))) : 0x80000000u;
        pm1[i] = r < HROWS ? (1u << hw) : 0u;
    }
    const int x_st = row0 * PITCH + s_slot * 16;                         // LDS byte of vector 0; vector i at + 12288 i
    uint4 px[NV];
    uint32_t pvm = 0;                                                    // validity bits of the vectors held in px
    auto issue_v = [&](const Tile& t, const Item& it, int i) {           // i static: load vector i of (tile, item) into px[i]
        const bool ok = ((pm0[i] & t.bad0) | (pm1[i] & t.bad1)) == 0u && it.c + s_slot * 8 < it.C;
        const uint32_t off = ok ? __umul24(t.base + (uint32_t)xvo[i], it.rowb) + (uint32_t)(s_slot * 16) : 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)it.base, 0, it.nrec, 0x00020000);
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, it.c * 2, 0);
        px[i] = make_uint4(q[0], q[1], q[2], q[3]);
        pvm = ok ? (pvm | (1u << i)) : (pvm & ~(1u << i));
    };
    float4 ncst[4];                                                      // constants of the item held in px (this thread's 8 channels)
    auto load_norm = [&](const Item& it) {
        if (!NORM) return;
        const float4* row = ntab + (it.tabc >> 1) + s_slot * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) ncst[j] = row[j];
    };
    // staging pieces (i, j static): word j of vector i (held in px) is normalised in place -- 7 vector-ALU operations, small enough to ride between two MFMAs
    auto commit_w = [&](int i, int j) {
        if (!NORM) return;
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
    auto commit_st = [&](char* buf, int i) {                             // ... and the finished vector written into `buf`
        if (row0 + 256 * i < HROWS) *(uint4*)(buf + x_st + i * (256 * PITCH)) = px[i];
    };
    auto commit_v = [&](char* buf, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) commit_w(i, j);
        commit_st(buf, i);
    };

    auto run = [&](auto NF_, auto LATE_) {
        constexpr int NF = std::remove_reference_t<decltype(NF_)>::value;
        constexpr bool late = std::remove_reference_t<decltype(LATE_)>::value;
        constexpr int RB = NF == 1 ? 3 : 2;                              // weight ring, in (kh, kw) groups: 3 divides the 9 groups of an item, 2 needs a move per item
        const int ntile0 = blockIdx.y * BN32 + (nhalf ? NF0 : 0);
        // A fragments: lane -> (row of the h pair, w) by row_to_hw, 16-byte half lane >> 5
        int hs, wl;
        row_to_hw_nt(lane & 31, hs, wl);
        const int a_lane = ((2 * hp + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
        // B fragments: buffer loads with a per-lane VGPR offset and wave-uniform SGPR offsets
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, 0x7FFFFFFF, 0x00020000);
        const uint32_t lane16 = (uint32_t)lane * 16u + (uint32_t)ntile0 * 1024u;
        auto load_b = [&](uint32_t wofs, int g, int kd, uint4* dst) {    // g = kh * 3 + kw (static), kd static
            const uint32_t so = wofs + (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(kd * 9 + g) * tapstride));
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + nf * 1024, so, 0);
                dst[nf] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        };
        // epilogue geometry of this wave: lane -> 16-byte column group cg of rows er0, er0 + 16 of a fragment
        const int cg = lane & 3, er0 = lane >> 2;
        int rhs[2], rw[2];
        row_to_hw_nt(er0, rhs[0], rw[0]);
        row_to_hw_nt(er0 + 16, rhs[1], rw[1]);
        float* scr = (float*)(scr_base + wave * SCR_BYTES);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nvox_total * (uint32_t)ldo * 2u, 0x00020000);

        f32x16_t acc[TD][NF];
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[d][nf][r] = 0.f;
        float s1[NF][8], s2[NF][8];                                      // running statistics over all tiles of this block
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

        // ---- prologue: item 0 staged synchronously into buffer 0, item 1 in flight in registers
        __syncthreads();                                                 // tables
        int k1 = 0, j1 = 0;                                              // (tile, slice) of the item held in px
        Tile t1 = fetch_tile(0);
        Item i1 = item_of(0);
        {
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t1, i1, i);
            load_norm(i1);
#pragma unroll
            for (int i = 0; i < NV; ++i) commit_v(bufs, i);
            if (++j1 == nk) { j1 = 0; ++k1; t1 = fetch_tile(k1); }
            i1 = item_of(j1);
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t1, i1, i);
        }
        uint4 bq[RB][3][NF];
        Item icur = item_of(0);
#pragma unroll
        for (int g = 0; g < RB - 1; ++g)
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) load_b(icur.wofs, g, kd, bq[g][kd]);
        __syncthreads();

        int kc = 0, jc = 0;                                              // (tile, slice) of the current item
        Tile tc = fetch_tile(0);
        // The SIMD arbitrates its two waves by age: without help the older wave (0-3) runs its 108 MFMAs in 5.6 k ticks and waits 3 k at the barrier while the
        // younger one needs 8.5 k, the last third of it alone on the SIMD with nothing to cover its operand waits (tools/kd_prof.sh).  Static priority for the
        // younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4) lets both finish together.
#ifndef KD_PRIO
#define KD_PRIO 1
#endif
        if (KD_PRIO == 1 && !late) __builtin_amdgcn_s_setprio(1);
        if (KD_PRIO == 2 && !late) __builtin_amdgcn_s_setprio(3);
        if (KD_PRIO == 3 && late) __builtin_amdgcn_s_setprio(1);
#ifdef KD_PROF
        unsigned long long pf_loop = 0, pf_bar = 0, pf_epi = 0;
        const unsigned long long pf_c0 = __builtin_readcyclecounter(), pf_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int it = 0; it < nitems; ++it) {
#ifdef KD_PROF
            const unsigned long long q0 = __builtin_readcyclecounter();
#endif
            const char* buf = bufs + (it & 1) * HALO_BYTES;
            char* nxt = bufs + ((it + 1) & 1) * HALO_BYTES;
            // item it + 1 sits in px (tile t1, slice i1); item it + 2 is requested by the hooks
            int k2 = k1, j2 = j1 + 1;
            if (j2 == nk) { j2 = 0; ++k2; }
            const Tile t2 = j2 == 0 ? fetch_tile(k2) : t1;
            const Item i2 = item_of(j2);
            load_norm(i1);
            const Item inext = i1;                                       // weights of the next item: the ring runs across the barrier
            const uint32_t a_base = (uint32_t)(a_lane);
            auto fetch_a = [&](int g, int dp) {                          // static: group g = kh * 3 + kw, halo plane dp
                const int kh = g / 3, kw = g % 3;
                return *(const uint4*)(buf + a_base + ((dp * HH + kh) * HW + kw) * PITCH);
            };
            constexpr int AR = 4, AD = 3;                                // fragment ring / prefetch distance
            uint4 aq[AR];
#pragma unroll
            for (int s = 0; s < AD; ++s) aq[s % AR] = fetch_a(s / HD, s % HD);
            // staging hooks: the older wave of a SIMD (waves 0-3) multiplies first and stages in the second half of the item, the younger one the other way
            // round (conv3d_wgrad2.hip: complementary halves)
#pragma unroll
            for (int g = 0; g < 9; ++g) {
#pragma unroll
                for (int dp = 0; dp < HD; ++dp) {
                    const int s = g * HD + dp;
#ifndef KD_SKIP_A
                    if (s + AD < 9 * HD) aq[(s + AD) % AR] = fetch_a((s + AD) / HD, (s + AD) % HD);
#endif
#ifndef KD_SKIP_B
                    if (dp < 3) {                                        // weights of group g + RB - 1 (possibly of the next item), tap kd = dp
                        const int gn = g + RB - 1;
                        if (gn < 9) load_b(icur.wofs, gn, dp, bq[gn % RB][dp]);
                        else load_b(inext.wofs, gn - 9, dp, bq[gn % RB][dp]);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const int d = dp - kd;
                        if (d < 0 || d >= TD) continue;
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf) {
#ifndef KD_SKIP_MMA
                            mma32<bf16_t>(acc[d][nf], aq[s % AR], bq[g % RB][kd][nf]);
#else
                            if (s == 0) mma32<bf16_t>(acc[d][nf], aq[s % AR], bq[g % RB][kd][nf]);
                            else asm volatile("" : : "v"(aq[s % AR]), "v"(bq[g % RB][kd][nf]));
#endif
                        }
                    }
                    // hooks: vector i of item it + 1 is normalised + written into the other buffer, then its register takes the load of item it + 2
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int hs_early = (i * 27) / NV, hs_late = 27 + (i * 27) / NV;
#ifndef KD_SKIP_STAGE                                                    // ablation switches (tools/kd_ablate.sh): where the time of an item goes
#ifndef KD_HOOKS
#define KD_HOOKS 1
#endif
                        if (KD_HOOKS == 0) {                             // whole vectors, complementary halves
                            if (s == (late ? hs_late : hs_early)) { commit_v(nxt, i); issue_v(t2, i2, i); }
                        } else {                                         // 25 pieces (4 words + store / next load per vector) spread over the 54 steps (1) or over this wave's half (2)
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                const int pc = i * 5 + j;
                                const int at = KD_HOOKS == 1 ? (pc * 54) / 25 + (late ? 1 : 0) : (late ? 27 : 0) + (pc * 27) / 25;
                                if (s == (at < 54 ? at : 53)) {
                                    if (j < 4) commit_w(i, j);
                                    else { commit_st(nxt, i); issue_v(t2, i2, i); }
                                }
                            }
                        }
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (RB == 2) {                                               // group 9 = group 0 of the next item was loaded into slot 1
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) bq[0][kd][nf] = bq[1][kd][nf];
            }

#ifdef KD_PROF
            const unsigned long long q1 = __builtin_readcyclecounter();
#endif
#ifdef KD_SKIP_EPI
            if (jc == nk - 1 && it == nitems - 1) {
#else
            if (jc == nk - 1) {
#endif
                // -------------------------------------------------------------- wave-private epilogue of this tile
                const int d0 = tc.org & 1023, h0 = (tc.org >> 10) & 1023, w0 = (int)(tc.org >> 20);
                const int hi = lane >> 5, col_l = lane & 31;
                uint32_t vox[TD][2];
                bool okv[TD][2];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int h = h0 + 2 * hp + rhs[ps], w = w0 + rw[ps];
#pragma unroll
                    for (int d = 0; d < TD; ++d) {
                        okv[d][ps] = d0 + d < pD && h < pH && w < pW;
                        vox[d][ps] = (uint32_t)(((n * pD + d0 + d) * pH + h) * pW + w);
                    }
                }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const int col0 = (ntile0 + nf) * 32 + cg * 8;        // first output column of this lane's vectors
                    const bool cok = col0 < Cout;
                    const bool useb = EPI == 1 && col0 >= eCa;
                    const uint64_t um64 = useb ? ~0ull : 0ull;
                    const uint32_t um = useb ? 0xFFFFFFFFu : 0u;
                    const bf16_t* es_x = (const bf16_t*)((uint64_t)exa ^ (((uint64_t)exa ^ (uint64_t)exb) & um64));
                    const uint32_t es_ld = (uint32_t)elda ^ (((uint32_t)elda ^ (uint32_t)eldb) & um);
                    const int ecol0 = col0 - (int)((uint32_t)eCa & um);
                    float emu[8], ers[8];
                    if (EPI == 1) {
                        const float4* e4 = (const float4*)(emr + (((nhalf ? NF0 : 0) + nf) * 32 + cg * 8) * 2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float4 v = e4[j]; emu[2 * j] = v.x; ers[2 * j] = v.y; emu[2 * j + 1] = v.z; ers[2 * j + 1] = v.w; }
                    }
                    uint4 ev[2][2];
                    auto epi_load = [&](int d, uint4* dst) {
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const bool ok = cok && okv[d][ps];
                            // branch-free (address select): divergent branches around VMEM make the compiler fall back to vmcnt(0)
                            const bf16_t* ptr = EPI == 1 ? (ok ? es_x + (size_t)(vox[d][ps] * es_ld + (uint32_t)ecol0) : (const bf16_t*)exa)
                                                         : (const bf16_t*)resp + (ok ? (size_t)(vox[d][ps] * (uint32_t)ldr + (uint32_t)col0) : (size_t)0);
                            dst[ps] = *(const uint4*)ptr;
                        }
                    };
                    if (EPI != 0) epi_load(0, ev[0]);
#pragma unroll
                    for (int d = 0; d < TD; ++d) {
                        if (EPI != 0 && d + 1 < TD) epi_load(d + 1, ev[(d + 1) & 1]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * SCR_ROW + col_l] = acc[d][nf][r];
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int row = er0 + 16 * ps;
                            float v[8];
                            const float4* sp = (const float4*)(scr + row * SCR_ROW + cg * 8);
                            { const float4 t4 = sp[0]; v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w; }
                            { const float4 t4 = sp[1]; v[4] = t4.x; v[5] = t4.y; v[6] = t4.z; v[7] = t4.w; }
                            const bool ok = cok && okv[d][ps];
                            if (EPI != 1) {
                                if (EPI == 2) {
                                    float rr[8];
                                    unpack16<bf16_t>(ev[d & 1][ps], rr);
#pragma unroll
                                    for (int q = 0; q < 8; ++q) v[q] += rr[q];
                                }
#pragma unroll
                                for (int q = 0; q < 8; ++q) { v[q] = ok ? Elem<bf16_t>::rnd(v[q]) : 0.f; s1[nf][q] += v[q]; s2[nf][q] += v[q] * v[q]; }
                            } else {
                                float xx[8];
                                unpack16<bf16_t>(ev[d & 1][ps], xx);
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float xn = (xx[q] - emu[q]) * ers[q];
                                    v[q] = Elem<bf16_t>::rnd((ok && xn > 0.f) ? v[q] : 0.f);
                                    s1[nf][q] += v[q]; s2[nf][q] += v[q] * xn;
                                }
                            }
                            const uint4 pk = pack16<bf16_t>(v);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pk), ors,
                                                                   ok ? (vox[d][ps] * (uint32_t)ldo + (uint32_t)col0) * 2u : 0xFFFFFFF0u, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);           // keep the scheduler from interleaving all passes (register pressure)
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[d][nf][r] = 0.f;
                    }
                }
            }
#ifdef KD_PROF
            const unsigned long long q2 = __builtin_readcyclecounter();
#endif
            __syncthreads();                                             // item it consumed, item it + 1 complete in the other buffer
#ifdef KD_PROF
            const unsigned long long q3 = __builtin_readcyclecounter();
            pf_loop += q1 - q0; pf_epi += q2 - q1; pf_bar += q3 - q2;
#endif
            // advance the item cursors
            if (++jc == nk) { jc = 0; ++kc; tc = fetch_tile(kc); }
            icur = inext;
            k1 = k2; j1 = j2; t1 = t2; i1 = i2;
        }

#ifdef KD_PROF
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
            g_kd_prof[wave * 4] = pf_loop; g_kd_prof[wave * 4 + 1] = pf_bar; g_kd_prof[wave * 4 + 2] = pf_epi; g_kd_prof[wave * 4 + 3] = (unsigned long long)nitems;
            if (wave == 0) { g_kd_prof[32] = __builtin_readcyclecounter() - pf_c0; g_kd_prof[33] = __builtin_amdgcn_s_memrealtime() - pf_r0; }
        }
#endif
        // statistics: ONE partial row per (block, h pair); this wave's columns
        if (partp) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int o = 4; o < 64; o <<= 1) { s1[nf][q] += __shfl_xor(s1[nf][q], o, 64); s2[nf][q] += __shfl_xor(s2[nf][q], o, 64); }
                }
                const int col0 = (ntile0 + nf) * 32 + cg * 8;
                const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(partp, 0, 0x7FFFFFFF, 0x00020000);
                const uint32_t poff = (lane < 4 && col0 < Cout) ? (uint32_t)(((((size_t)n * gx + blockIdx.x) * 4 + hp) * Cout + col0) * 8) : 0xFFFFFFF0u;
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    u32x4_t pv;
                    pv[0] = __float_as_uint(s1[nf][q]); pv[1] = __float_as_uint(s2[nf][q]); pv[2] = __float_as_uint(s1[nf][q + 1]); pv[3] = __float_as_uint(s2[nf][q + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(pv, prs, poff == 0xFFFFFFF0u ? poff : poff + q * 8, 0, 0);
                }
            }
        }
    };
    if (nhalf == 0) run(std::integral_constant<int, NF0>{}, std::true_type{});
    else run(std::integral_constant<int, NF1>{}, std::false_type{});
}

int kd_grid_x(int tiles, int gy, int N) {                                // ~one persistent block per CU
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}
int kd_tiles(int D, int H, int W) { return ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }

template <int NF0, int NF1>
int launch_kd(const IgemmParams& p, int epi, hipStream_t st) {
    const int tiles = kd_tiles(p.D, p.H, p.W);
    const int gy = p.ntiles / (NF0 + NF1);
    const int gx = kd_grid_x(tiles, gy, p.N);
    const size_t smem = 2 * (size_t)HALO_BYTES + (size_t)(p.a.C + p.b.C) * 8 + (NF0 + NF1) * 256 + NW * (size_t)SCR_BYTES + ((tiles + gx - 1) / gx + 4) * 16;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    dim3 grid(gx, gy, p.N), block(NT);
    const bool norm = p.a.mr != nullptr;
#define KD_LAUNCH(E, NRM)                                                                                              \
    {                                                                                                                  \
        auto k = igemm_kd_kernel<NF0, NF1, E, NRM>;                                                                    \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);              \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                                               \
    }
    if (epi == 1) { if (norm) KD_LAUNCH(1, true) else KD_LAUNCH(1, false) }
    else if (p.res) { if (norm) KD_LAUNCH(2, true) else KD_LAUNCH(2, false) }
    else { if (norm) KD_LAUNCH(0, true) else KD_LAUNCH(0, false) }
#undef KD_LAUNCH
#ifdef KD_PROF
    if (getenv("RSUPER_KD_PROF")) {
        unsigned long long h[34];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_kd_prof), sizeof(h));
        fprintf(stderr, "kd_prof epi%d norm%d (cycles per item; readcyclecounter ticks):", epi, (int)norm);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d loop %.0f epi %.0f bar %.0f |", w, (double)h[w * 4] / h[w * 4 + 3], (double)h[w * 4 + 2] / h[w * 4 + 3], (double)h[w * 4 + 1] / h[w * 4 + 3]);
        fprintf(stderr, " items %llu; %llu ticks in %.1f us (100 MHz counter) = %.0f MHz\n", h[3], h[32], h[33] / 100.0, h[32] / (h[33] / 100.0));
    }
#endif
    return rs_check_launch();
}

}  // namespace

// bf16, 64 / 96 / 128-column blocks; both sources normalised or both raw; 24-bit voxel arithmetic in the staging addresses
bool rs_igemm_kd_supported(const IgemmParams& p, int dtype) {
    if (dtype != RS_BF16 || (p.bn != 64 && p.bn != 96 && p.bn != 128) || p.ntiles % (p.bn / 32)) return false;
    if (p.b.C > 0 && (p.a.mr != nullptr) != (p.b.mr != nullptr)) return false;
    if ((long)p.N * p.D * p.H * p.W >= (1l << 24) || p.D > 1023 || p.H > 1023 || p.W > 1023) return false;
    if (p.a.ld * 2 >= (1 << 24) || p.b.ld * 2 >= (1 << 24)) return false;
    return true;
}
int rs_igemm_kd_part_rows(int bn, int N, int D, int H, int W, int n_cols) {
    const int gy = (n_cols + bn - 1) / bn;
    return kd_grid_x(kd_tiles(D, H, W), gy, N) * 4;
}
int rs_launch_igemm_kd(const IgemmParams& p, int epi, hipStream_t st) {
    switch (p.bn) {
        case 64: return launch_kd<1, 1>(p, epi, st);
        case 96: return launch_kd<2, 1>(p, epi, st);
        case 128: return launch_kd<2, 2>(p, epi, st);
    }
    return RS_ERR_ARG;
}
