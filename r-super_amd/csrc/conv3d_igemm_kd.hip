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
//   * K is staged 16 channels (one MFMA k-step) at a time: halo 6 x 10 x 18 rows x 48-byte pitch = 51.8 KB, two buffers, ONE barrier per item.
//     Normalised sources (forward): the eight waves stage the next item themselves from hooks in the MFMA loop (norm + ReLU in registers, one
//     word = 7 vector-ALU operations per hook, ds_write_b128), the loads of the item after that in flight in registers (conv3d_wgrad2.hip's scheme).
//     Raw sources (data gradient: dY): LDS-DMA, 7 wave instructions per wave and item, no registers, no arithmetic -- which is what leaves room for
//     two column fragments per wave (96 / 128-column blocks);
//   * persistent blocks over the tiles of a sample (XCD-aware order), per-block tile descriptors in LDS (no index arithmetic in the loop),
//     wave-private epilogue through a 4.6 KB LDS scratch whose operands (residual / forward input of the ReLU mask) are requested from inside the
//     tile's last MFMA loop, statistics reduced per tile into per-wave LDS accumulators (one partial row per (block, h pair)).
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

namespace {

constexpr int TD = 4, TH = 8, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HROWS = HD * HH * HW;                 // 1080 halo rows
constexpr int PITCH = 48;                           // 32 B of data (16 bf16 channels) + 16 B: odd multiple of 16 -> conflict-free ds_read_b128
constexpr int HB = 51 * 1024;                       // buffer stride (>= 1080 x 48): the LDS-DMA path fills a buffer in 51 pieces of 1 KB
constexpr int NP = 7;                               // DMA pieces per wave and item (51 over 8 waves)
constexpr int NT = 512, NW = 8;
constexpr int NV = 5;                               // 16-byte staging vectors per thread and item, register path (2160 over 512 threads)
constexpr int SCR_ROW = 36;                         // floats per epilogue scratch row
constexpr int SCR_BYTES = 32 * SCR_ROW * 4;         // per wave
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// raw sources need no arithmetic on their way into LDS: `buffer_load_dwordx4 ... lds` copies 1 KB per wave instruction from per-lane global offsets into a
// lane-linear LDS image (lane l -> M0 + 16 l), out-of-range offsets arrive as zeros (semantics pinned by tools/ubench/lds_dma_probe.hip, as in
// conv3d_wgrad2.hip).  The compiler does not see the instruction: completion is waited for by an explicit counted vmcnt before the item barrier.
__device__ __forceinline__ void kd_dma16(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff, uint32_t lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_byte), "s"(rs), "s"(soff) : "memory", "m0");
}

// row_to_hw (common.hpp) without branches: the if-chain becomes a private-memory lookup table (scratch) otherwise
__device__ __forceinline__ void row_to_hw_nt(int i, int& hs, int& w) {
    hs = (int)((0xF00F0FF0u >> i) & 1u);
    const unsigned long long t = i < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
    w = (int)((t >> ((i & 15) * 4)) & 15ull);
}

#ifdef KD_PROF
__device__ unsigned long long g_kd_prof[8 * 4 + 2];                // block (0, 0, 0): [wave][item-loop, barrier, epilogue ticks, items], kernel ticks, 100 MHz ticks
#endif

// one 16-channel slice of [a | b]: first channel, channels / row bytes / base / bytes of its source, table index of its constants, byte offset of its packed weights (tap 0)
struct Item { int c, C; uint32_t rowb, nrec; uint64_t base; int tabc; uint32_t wofs; };
struct Tile { uint32_t base, bad0, bad1, org; };

// NF0 / NF1: 32-column fragments of the waves 0-3 / 4-7 (block = (NF0 + NF1) x 32 columns).  EPI: 0 forward, 1 data gradient, 2 forward + residual.
// NORM: both sources carry (mean, rstd) and are staged through registers; raw sources are staged by LDS-DMA.
template <int NF0, int NF1, int EPI, bool NORM>
__global__ __launch_bounds__(NT, 2) void igemm_kd_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN32 = NF0 + NF1;
#ifndef KD_DMA
#define KD_DMA 1
#endif
    // raw sources: LDS-DMA for the wide blocks (no registers to spare there); 64-column blocks stage through registers like the normalised ones -- with the
    // DMA pieces in the VMEM queue every wait for a weight fragment issued after them also waits for the pieces (in-order vmcnt): item loop 4.4 k -> 6.8 k ticks
    constexpr bool DMA = !NORM && (KD_DMA == 2 || (KD_DMA == 1 && BN32 > 2));
    // Every field is copied into a local once, and selections between the two sources are mask arithmetic further down: inside by-reference lambdas a
    // `cond ? b : a` of two such variables becomes a select of two ADDRESSES -- the variables then live in scratch, every use is a flat load + vmcnt(0) and
    // everything derived from it a per-lane value (waterfall loops around the buffer instructions).  Measured: 938 us vs 559 us on up4.0 forward.
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
    const uint32_t osplit = (uint32_t)p.out_split, opart = (uint32_t)p.out_part;       // (locals, like every other kernel-argument field: see the note on captures below)
    const void* const wpk = p.wp; void* const outp = p.out; const void* const resp = p.res; float* const partp = p.part;
    const int ctot = Ca + Cb;
    char* bufs = smem;                                                  // 2 x HB
    float4* ntab = (float4*)(smem + 2 * HB);                            // [(Ca + Cb) / 2] (sc0, sc1, nb0, nb1)   (NORM)
    float* emr = (float*)(smem + 2 * HB + (NORM ? ctot * 8 : 0));       // [BN32 * 32][mean, rstd] of the epilogue source (EPI 1)
    float* sacc = emr + BN32 * 64;                                      // [wave][2][32][sum, sum2]: statistics of this block's tiles
    char* scr_base = (char*)(sacc + NW * 128);
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
    const uint32_t tapstride = (uint32_t)ntiles * 2048u;                // bytes between consecutive taps of one (chunk, k-step) in the packed weights
    const uint32_t nvox_total = (uint32_t)(pN * pD * pH * pW);
    const uint32_t rowbA = (uint32_t)lda * 2u, rowbB = (uint32_t)ldb * 2u;
    const uint32_t nrecA = nvox_total * rowbA, nrecB = Cb ? nvox_total * rowbB : 0u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic region

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
    for (int i = tid; i < NW * 128; i += NT) sacc[i] = 0.f;
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
            int tw, th, td;
            rs_tile_coords(t, tiles_w, tiles_h, tiles_d, tw, th, td);
            const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
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
    auto fetch_tile = [&](int k) {                                      // wave-uniform: one broadcast LDS read + readfirstlanes
        const uint4 v = dtab[k];
        Tile t;
        t.base = __builtin_amdgcn_readfirstlane(v.x); t.bad0 = __builtin_amdgcn_readfirstlane(v.y);
        t.bad1 = __builtin_amdgcn_readfirstlane(v.z); t.org = __builtin_amdgcn_readfirstlane(v.w);
        return t;
    };
    auto item_of = [&](int j) {                                         // j-th 16-channel slice of [a | b] (mask arithmetic: see the top)
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

    // ---- staging through registers (NORM): thread -> 16-byte slot tid & 1 of halo rows perm(tid >> 1) + 256 i.  Inside every run of 8 rows the order is
    //      0,2,4,6,1,3,5,7: the eight lanes of a ds_write_b128 group then cover rows r, r+2, r+4, r+6 = dword offsets 12 r + {0, 24, 48, 72} + 0..7 -> 32 banks
    const int s_slot = tid & 1, rk = tid >> 1;
    const int row0 = (rk & ~7) | ((rk & 3) << 1) | ((rk >> 2) & 1);
    int xvo[NV];
    uint32_t pm0[NV], pm1[NV];
    if (!DMA) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = row0 + 256 * i;
            const int hd = r / (HH * HW), rem = r - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            xvo[i] = (hd * pH + hh) * pW + hw;
            pm0[i] = r < HROWS ? ((1u << hd) | (1u << (6 + hh))) : 0x80000000u;
            pm1[i] = r < HROWS ? (1u << hw) : 0u;
        }
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

    // ---- staging by LDS-DMA (raw sources): piece q = wave + 8 k of a buffer is its bytes [1024 q, 1024 q + 1024); lane l moves 16-byte unit u = 64 q + l
    //      = (halo row u / 3, slot u % 3; slot 2 is the pad of the 48-byte pitch and rows >= 1080 do not exist: out-of-range offset, zeros)
    uint32_t dvo[NP];                                                    // voxel offset of the unit's halo row | slot << 24 | invalid << 25
    auto dma_geom = [&](int k, int& hd, int& hh, int& hw, int& slot, bool& valid) {
        const int u = 64 * (wave + 8 * k) + lane;
        const int r = u / 3;
        slot = u - 3 * r;
        hd = r / (HH * HW);
        const int rem = r - hd * (HH * HW);
        hh = rem / HW; hw = rem - hh * HW;
        valid = r < HROWS && slot < 2 && wave + 8 * k < 51;
    };
    if (DMA) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            int hd, hh, hw, slot; bool valid;
            dma_geom(k, hd, hh, hw, slot, valid);
            dvo[k] = valid ? ((uint32_t)((hd * pH + hh) * pW + hw) | ((uint32_t)slot << 24)) : (1u << 25);
        }
    }
    auto dma_item = [&](const Tile& t, const Item& it, uint32_t buf_off) {   // stage (tile, item) into the buffer at byte offset buf_off
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)it.base, 0, it.nrec, 0x00020000);
        uint32_t lbase = lds0 + buf_off + (uint32_t)wave * 1024u;
        asm volatile("" : "+s"(lbase));                                  // (keeps the 7 piece addresses of both buffers from being hoisted into scalar registers)
        const bool boundary = ((t.bad0 & 0xFFFFu) | t.bad1) != 0u;       // wave-uniform: a tile that touches a face of the volume (or a dead entry)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (wave + 8 * k >= 51) continue;                            // wave-uniform
            const uint32_t q = dvo[k];
            const uint32_t slot = (q >> 24) & 1u;
            bool ok = (q >> 25) == 0u && it.c + (int)slot * 8 < it.C;
            if (boundary) {                                              // no VMEM inside the branch; interior tiles (the bulk) skip the coordinate arithmetic
                int hd, hh, hw, sl; bool valid;
                dma_geom(k, hd, hh, hw, sl, valid);
                ok = ok && (((t.bad0 >> hd) | (t.bad0 >> (6 + hh)) | (t.bad1 >> hw)) & 1u) == 0u;
            }
            const uint32_t voff = ok ? __umul24(t.base + (q & 0xFFFFFFu), it.rowb) + slot * 16u : 0xFFFFFFF0u;
            kd_dma16(rs, voff, (uint32_t)(it.c * 2), lbase + (uint32_t)k * 8192u);
        }
    };

    auto run = [&](auto NF_, auto LATE_) {
        constexpr int NF = std::remove_reference_t<decltype(NF_)>::value;
        constexpr bool late = std::remove_reference_t<decltype(LATE_)>::value;
#ifndef KD_RB
#define KD_RB 3
#endif
        constexpr int RB = NF == 1 ? KD_RB : 2;                                        // weight ring, in (kh, kw) groups: 3 divides the 9 groups of an item, 2 needs a move per item
        constexpr int EVF = NF == 1 ? 2 : 1;                             // fragments whose epilogue operands are requested ahead (2 vectors each)
        const int ntile0 = blockIdx.y * BN32 + (nhalf ? NF0 : 0);
        // A fragments: lane -> (row of the h pair, w) by row_to_hw, 16-byte half lane >> 5
        int hs, wl;
        row_to_hw_nt(lane & 31, hs, wl);
        const int a_lane = ((2 * hp + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
        // B fragments: buffer loads with a per-lane VGPR offset and wave-uniform SGPR offsets
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, 0x7FFFFFFF, 0x00020000);
        const uint32_t lane16 = (uint32_t)lane * 16u + (uint32_t)ntile0 * 1024u;
        uint32_t ts_item = tapstride;                                    // re-laundered every item: the 27 tap offsets are loop invariants the compiler would
                                                                         // otherwise keep in 27 scalar registers across the whole loop (scalar spills)
        auto load_b = [&](uint32_t wofs, int g, int kd, uint4* dst) {    // g = kh * 3 + kw (static), kd static
            const uint32_t so = wofs + (uint32_t)(kd * 9 + g) * ts_item;
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
        float* sw = sacc + wave * 128;
        // p.out_split > 0: the columns from out_split on live in a second tensor of the same row stride, out_part elements behind the first (the two halves of a
        // fused [conv1 | shortcut] output whose halves are narrower than a cache line: consumers then read whole lines)
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (nvox_total * (uint32_t)ldo + (osplit ? opart : 0u)) * 2u, 0x00020000);
        // epilogue operand (residual / forward input of the ReLU mask) of column fragment nf: source, row pitch, this lane's first column -- recomputed
        // where needed (a few operations) instead of living in registers across the MFMA loop
        auto ep_desc = [&](int nf, const bf16_t*& x, uint32_t& ld, int& c0, bool& cok) {
            const int col0 = (ntile0 + nf) * 32 + cg * 8;
            cok = col0 < Cout;
            if (EPI == 1) {
                const bool useb = col0 >= eCa;
                const uint64_t um64 = useb ? ~0ull : 0ull;
                const uint32_t um = useb ? 0xFFFFFFFFu : 0u;
                x = (const bf16_t*)((uint64_t)exa ^ (((uint64_t)exa ^ (uint64_t)exb) & um64));
                ld = (uint32_t)elda ^ (((uint32_t)elda ^ (uint32_t)eldb) & um);
                c0 = col0 - (int)((uint32_t)eCa & um);
            } else {
                x = (const bf16_t*)resp; ld = (uint32_t)ldr; c0 = col0;
            }
        };
        const bf16_t* const dummy = (const bf16_t*)(EPI == 1 ? exa : (resp ? resp : xa));   // a mapped address for the requests of items that end no tile

        f32x16_t acc[TD][NF];
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[d][nf][r] = 0.f;
        constexpr bool RSTAT = NF == 1;                                  // statistics over the block's tiles in registers (NF 1) or reduced per fragment into LDS (NF 2: no registers left)
        f32x2_t rs1[4], rs2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { rs1[q] = f32x2_t{0.f, 0.f}; rs2[q] = f32x2_t{0.f, 0.f}; }
        const bool wave_live = ntile0 * 32 < Cout;                       // a wave whose columns all lie past Cout (96 columns on 64-column blocks) multiplies nothing

        // ---- prologue: item 0 staged synchronously into buffer 0; register path: item 1 in flight in registers
        __syncthreads();                                                 // tables
        int k1 = 0, j1 = 0;                                              // (tile, slice) of the NEXT item (register path: the one held in px)
        Tile t1 = fetch_tile(0);
        Item i1 = item_of(0);
        if constexpr (DMA) {
            dma_item(t1, i1, 0u);
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t1, i1, i);
            load_norm(i1);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) commit_w(i, j);
                commit_st(bufs, i);
            }
        }
        if (++j1 == nk) { j1 = 0; ++k1; t1 = fetch_tile(k1); }
        i1 = item_of(j1);
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_v(t1, i1, i);
        }
        uint4 bq[RB][3][NF];
        uint32_t wofs_cur = item_of(0).wofs;
#pragma unroll
        for (int g = 0; g < RB - 1; ++g)
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) load_b(wofs_cur, g, kd, bq[g][kd]);
        __syncthreads();

        int kc = 0, jc = 0;                                              // (tile, slice) of the current item
        Tile tc = fetch_tile(0);
#ifndef KD_PRIO
#define KD_PRIO 5
#endif
        if (KD_PRIO == 1 && !late) __builtin_amdgcn_s_setprio(1);        // measured: priority only swaps which wave of a SIMD runs ahead (tools/kd_prof.sh)
#ifdef KD_PROF
        unsigned long long pf_loop = 0, pf_bar = 0, pf_epi = 0;
        const unsigned long long pf_c0 = __builtin_readcyclecounter(), pf_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int it = 0; it < nitems; ++it) {
#ifdef KD_PROF
            const unsigned long long q0 = __builtin_readcyclecounter();
            unsigned long long q1 = q0;
#endif
            const char* buf = bufs + (it & 1) * HB;
            char* nxt = bufs + ((it + 1) & 1) * HB;
            const bool last = jc == nk - 1;                              // this item completes a tile
            asm volatile("" : "+s"(ts_item));
            // register path: item it + 1 sits in px (tile t1, slice i1), item it + 2 is requested by the hooks.  DMA path: item it + 1 is requested now.
            int k2 = k1, j2 = j1 + 1;
            if (j2 == nk) { j2 = 0; ++k2; }
            Tile t2 = t1;
            Item i2 = i1;
            const uint32_t wofs_next = i1.wofs;                          // weights of the next item: the ring runs across the barrier
            if constexpr (DMA) {
#ifndef KD_SKIP_STAGE
                dma_item(t1, i1, (uint32_t)((it + 1) & 1) * HB);
#endif
            } else {
                if (j2 == 0) t2 = fetch_tile(k2);
                i2 = item_of(j2);
                load_norm(i1);
            }
            const uint32_t a_base = (uint32_t)(a_lane);
            auto fetch_a = [&](int g, int dp) {                          // static: group g = kh * 3 + kw, halo plane dp
                const int kh = g / 3, kw = g % 3;
                return *(const uint4*)(buf + a_base + ((dp * HH + kh) * HW + kw) * PITCH);
            };
            // epilogue operands of fragment f of the tile this item may complete (f = nf * TD + d): requested from inside the MFMA loop -- every item issues
            // the loads (address select: a branch around VMEM costs a vmcnt(0) at the join), items that end no tile read one cached dummy vector
            const int d0 = tc.org & 1023, h0 = (tc.org >> 10) & 1023, w0 = (int)(tc.org >> 20);
            const uint32_t dstride = (uint32_t)(pH * pW);
            auto out_vox = [&](int ps, int d, uint32_t& vx) {            // voxel of rows er0 + 16 ps of fragment plane d; false: outside the volume
                const int h = h0 + 2 * hp + rhs[ps], w = w0 + rw[ps];
                vx = (uint32_t)(((n * pD + d0 + d) * pH + h) * pW + w);
                return h < pH && w < pW && d0 + d < pD;
            };
            uint4 ev[EVF][2];
            auto epi_load = [&](int f, bool real) {                      // f static
                const int nf = f / TD, d = f % TD;
                const bf16_t* ex; uint32_t eld; int ec0; bool cok;
                ep_desc(nf, ex, eld, ec0, cok);
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    uint32_t vx;
                    const bool ok = out_vox(ps, d, vx) && real && cok;
                    const bf16_t* ptr = ok ? ex + (size_t)(vx * eld + (uint32_t)ec0) : dummy;
                    ev[f % EVF][ps] = *(const uint4*)ptr;
                }
            };
#ifndef KD_AD
#define KD_AD 5
#endif
            constexpr int AD = NF == 1 ? ((EPI != 0 && NORM) ? KD_AD - 2 : KD_AD) : 2, AR = AD + 1;   // (residual / mask operands in flight: fewer fragments ahead, no spill)                       // fragment ring / prefetch distance
            uint4 aq[AR];
            if (!wave_live) {                                            // a dead wave only stages its share (register path) and meets the others at the barrier
                if constexpr (!DMA) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) commit_w(i, j);
                        commit_st(nxt, i);
                        issue_v(t2, i2, i);
                    }
                }
                goto item_done;
            }
#pragma unroll
            for (int s = 0; s < AD; ++s) aq[s % AR] = fetch_a(s / HD, s % HD);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                // progress-based priority: the wave of a SIMD that is BEHIND in the item outranks its partner, so the two advance together and cover each
                // other's operand waits (by age alone the older wave runs ahead and the younger one finishes the last third of the item alone)
                if (KD_PRIO == 4) {
                    if (g == 0) __builtin_amdgcn_s_setprio(3);
                    if (g == 3) __builtin_amdgcn_s_setprio(2);
                    if (g == 6) __builtin_amdgcn_s_setprio(1);
                }
                if (KD_PRIO == 5) {
                    if (g == 0) __builtin_amdgcn_s_setprio(3);
                    if (g == 2) __builtin_amdgcn_s_setprio(2);
                    if (g == 4) __builtin_amdgcn_s_setprio(1);
                    if (g == 6) __builtin_amdgcn_s_setprio(0);
                }
#pragma unroll
                for (int dp = 0; dp < HD; ++dp) {
                    const int s = g * HD + dp;
#ifndef KD_SKIP_A
                    if (s + AD < 9 * HD) aq[(s + AD) % AR] = fetch_a((s + AD) / HD, (s + AD) % HD);
#endif
#ifndef KD_SKIP_B
                    if (dp < 3) {                                        // weights of group g + RB - 1 (possibly of the next item), tap kd = dp
                        const int gn = g + RB - 1;
                        if (gn < 9) load_b(wofs_cur, gn, dp, bq[gn % RB][dp]);
                        else load_b(wofs_next, gn - 9, dp, bq[gn % RB][dp]);
                    }
#endif
                    if (EPI != 0) {
#pragma unroll
                        for (int f = 0; f < EVF; ++f)
                            if (s == 6 + 2 * f) epi_load(f, last);
                    }
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
#ifndef KD_SKIP_STAGE                                                    // ablation switches (tools/kd_ablate.sh): where the time of an item goes
                    if constexpr (!DMA) {
                        // hooks: 25 pieces per item (4 words + store / next load per vector) spread over the 54 steps; the two waves of a SIMD one step apart
#pragma unroll
                        for (int i = 0; i < NV; ++i)
#pragma unroll
                            for (int j = 0; j < 5; ++j) {
                                const int at = ((i * 5 + j) * 54) / 25 + (late ? 1 : 0);
                                if (s == (at < 54 ? at : 53)) {
                                    if (j < 4) commit_w(i, j);
                                    else { commit_st(nxt, i); issue_v(t2, i2, i); }
                                }
                            }
                    }
#endif
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
            q1 = __builtin_readcyclecounter();
#endif
#ifdef KD_SKIP_EPI
            if (last && it == nitems - 1) {
#else
            if (last) {
#endif
                // -------------------------------------------------------------- wave-private epilogue of this tile
                const int hi = lane >> 5, col_l = lane & 31;
#pragma unroll
                for (int f = 0; f < NF * TD; ++f) {
                    const int nf = f / TD, d = f % TD;
                    const int col0 = (ntile0 + nf) * 32 + cg * 8;        // first output column of this lane's vectors
                    float emu[8], ers[8];
                    if (EPI == 1) {
                        const float4* e4 = (const float4*)(emr + (((nhalf ? NF0 : 0) + nf) * 32 + cg * 8) * 2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float4 v = e4[j]; emu[2 * j] = v.x; ers[2 * j] = v.y; emu[2 * j + 1] = v.z; ers[2 * j + 1] = v.w; }
                    }
                    // statistics as packed pairs (v_pk_add_f32 / v_pk_fma_f32): NF 1 accumulates straight into the block's running sums
                    f32x2_t s1[4], s2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { s1[q] = RSTAT ? rs1[q] : f32x2_t{0.f, 0.f}; s2[q] = RSTAT ? rs2[q] : f32x2_t{0.f, 0.f}; }
                    // 1) fragment -> scratch, fragment-row major (same wave: its LDS accesses are ordered)
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * SCR_ROW + col_l] = acc[d][nf][r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[d][nf][r] = 0.f;
                    // 2) scratch -> 16-byte vectors; two elements per operation where the ISA has packed f32 forms, ONE bf16 conversion per pair (its halves,
                    //    re-expanded, are the rounded values the statistics need)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = er0 + 16 * ps;
                        const float4* sp = (const float4*)(scr + row * SCR_ROW + cg * 8);
                        const float4 ta = sp[0], tb = sp[1];
                        const f32x2_t v2[4] = {{ta.x, ta.y}, {ta.z, ta.w}, {tb.x, tb.y}, {tb.z, tb.w}};
                        uint32_t vx;
                        const bool ok = out_vox(ps, d, vx) && col0 < Cout;
                        const uint4 e4 = ev[f % EVF][ps];
                        const uint32_t ew[4] = {e4.x, e4.y, e4.z, e4.w};
                        uint32_t ow[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x2_t v = v2[q];
                            f32x2_t xn = {0.f, 0.f};
                            if (EPI != 0) {
                                const f32x2_t xx = {__uint_as_float(ew[q] << 16), __uint_as_float(ew[q] & 0xffff0000u)};
                                if (EPI == 2) v = v + xx;
                                else {
                                    const f32x2_t nm = {-emu[2 * q], -emu[2 * q + 1]}, rs = {ers[2 * q], ers[2 * q + 1]};
                                    xn = (xx + nm) * rs;                 // == (x - mean) * rstd, the expression every data-gradient epilogue evaluates
                                }
                            }
                            const float a0 = (ok && (EPI != 1 || xn[0] > 0.f)) ? v[0] : 0.f, a1 = (ok && (EPI != 1 || xn[1] > 0.f)) ? v[1] : 0.f;
                            const uint32_t w = f2bf2(a0, a1);
                            ow[q] = w;
                            const f32x2_t r = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                            s1[q] = s1[q] + r;
                            s2[q] = __builtin_elementwise_fma(r, EPI == 1 ? xn : r, s2[q]);
                        }
                        const u32x4_t pk = {ow[0], ow[1], ow[2], ow[3]};
                        const uint32_t ocol = (osplit && (uint32_t)col0 >= osplit) ? (uint32_t)col0 - osplit + opart : (uint32_t)col0;
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ors, ok ? (vx * (uint32_t)ldo + ocol) * 2u : 0xFFFFFFF0u, 0, 0);
                    }
                    if (EPI != 0 && f + EVF < NF * TD) epi_load(f + EVF, true);   // the slot just consumed takes the operands of fragment f + EVF
                    // 3) statistics of this fragment: lanes with the same column group hold partial sums over their rows -> reduce, accumulate per wave in LDS
                    if (RSTAT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { rs1[q] = s1[q]; rs2[q] = s2[q]; }
                    } else if (partp) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
#pragma unroll
                            for (int o = 4; o < 64; o <<= 1) {
                                s1[q][0] += __shfl_xor(s1[q][0], o, 64); s1[q][1] += __shfl_xor(s1[q][1], o, 64);
                                s2[q][0] += __shfl_xor(s2[q][0], o, 64); s2[q][1] += __shfl_xor(s2[q][1], o, 64);
                            }
                        }
                        if (lane < 4) {
                            float4* dst = (float4*)(sw + nf * 64 + cg * 16);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float4 a = dst[q];
                                a.x += s1[q][0]; a.y += s2[q][0]; a.z += s1[q][1]; a.w += s2[q][1];
                                dst[q] = a;
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);                   // keep the scheduler from interleaving all fragments (register pressure)
                }
            }
        item_done:
#ifdef KD_PROF
            const unsigned long long q2 = __builtin_readcyclecounter();
#endif
            if constexpr (DMA) {
                // this item's DMA pieces were its first VMEM operations: at most the weight fragments requested for the next item may still be in flight
                // (a tile's last item also issued the epilogue's stores after them: drain)
                if (last) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"((RB - 1) * 3 * NF) : "memory");
            }
            __syncthreads();                                             // item it consumed, item it + 1 complete in the other buffer
#ifdef KD_PROF
            const unsigned long long q3 = __builtin_readcyclecounter();
            pf_loop += q1 - q0; pf_epi += q2 - q1; pf_bar += q3 - q2;
#endif
            // advance the item cursors
            if (++jc == nk) { jc = 0; ++kc; tc = fetch_tile(kc); }
            wofs_cur = wofs_next;
            if constexpr (DMA) {
                if (j2 == 0) t1 = fetch_tile(k2);
                i1 = item_of(j2);
            } else { t1 = t2; i1 = i2; }
            k1 = k2; j1 = j2;
        }
#ifdef KD_PROF
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
            g_kd_prof[wave * 4] = pf_loop; g_kd_prof[wave * 4 + 1] = pf_bar; g_kd_prof[wave * 4 + 2] = pf_epi; g_kd_prof[wave * 4 + 3] = (unsigned long long)nitems;
            if (wave == 0) { g_kd_prof[32] = __builtin_readcyclecounter() - pf_c0; g_kd_prof[33] = __builtin_amdgcn_s_memrealtime() - pf_r0; }
        }
#endif
        // statistics: ONE partial row per (block, h pair); this wave's columns
        if (RSTAT && partp) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {
                    rs1[q][0] += __shfl_xor(rs1[q][0], o, 64); rs1[q][1] += __shfl_xor(rs1[q][1], o, 64);
                    rs2[q][0] += __shfl_xor(rs2[q][0], o, 64); rs2[q][1] += __shfl_xor(rs2[q][1], o, 64);
                }
            }
            if (lane < 4) {
                float4* dst = (float4*)(sw + cg * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = make_float4(rs1[q][0], rs2[q][0], rs1[q][1], rs2[q][1]);
            }
        }
        if (partp && lane < 4) {
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(partp, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int col0 = (ntile0 + nf) * 32 + cg * 8;
                const uint32_t poff = col0 < Cout ? (uint32_t)(((((size_t)n * gx + blockIdx.x) * 4 + hp) * Cout + col0) * 8) : 0xFFFFFFF0u;
                const float4* src = (const float4*)(sw + nf * 64 + cg * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = src[q];
                    u32x4_t pv;
                    pv[0] = __float_as_uint(a.x); pv[1] = __float_as_uint(a.y); pv[2] = __float_as_uint(a.z); pv[3] = __float_as_uint(a.w);
                    __builtin_amdgcn_raw_buffer_store_b128(pv, prs, poff == 0xFFFFFFF0u ? poff : poff + q * 16, 0, 0);
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
    const bool norm = p.a.mr != nullptr;
    const size_t smem = 2 * (size_t)HB + (norm ? (size_t)(p.a.C + p.b.C) * 8 : 0) + (NF0 + NF1) * 256 + NW * 512 + NW * (size_t)SCR_BYTES + ((tiles + gx - 1) / gx + 4) * 16;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    dim3 grid(gx, gy, p.N), block(NT);
#define KD_LAUNCH(E, NRM)                                                                                              \
    {                                                                                                                  \
        auto k = igemm_kd_kernel<NF0, NF1, E, NRM>;                                                                    \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);              \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                                               \
    }
    if constexpr (NF0 == 1) {                                            // register staging: one column fragment per wave (two do not fit the register file)
        if (epi == 1) { if (norm) KD_LAUNCH(1, true) else KD_LAUNCH(1, false) }
        else if (p.res) { if (norm) KD_LAUNCH(2, true) else KD_LAUNCH(2, false) }
        else { if (norm) KD_LAUNCH(0, true) else KD_LAUNCH(0, false) }
    } else {
        if (norm) return RS_ERR_UNSUPPORTED;
        if (epi == 1) KD_LAUNCH(1, false)
        else if (p.res) KD_LAUNCH(2, false)
        else KD_LAUNCH(0, false)
    }
#undef KD_LAUNCH
#ifdef KD_PROF
    if (getenv("RSUPER_KD_PROF")) {
        unsigned long long h[34];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_kd_prof), sizeof(h));
        fprintf(stderr, "kd_prof epi%d norm%d bn%d (ticks per item):", epi, (int)norm, p.bn);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d loop %.0f epi %.0f bar %.0f |", w, (double)h[w * 4] / h[w * 4 + 3], (double)h[w * 4 + 2] / h[w * 4 + 3], (double)h[w * 4 + 1] / h[w * 4 + 3]);
        fprintf(stderr, " items %llu; %llu ticks in %.1f us (100 MHz counter) = %.0f MHz\n", h[3], h[32], h[33] / 100.0, h[32] / (h[33] / 100.0));
    }
#endif
    return rs_check_launch();
}

}  // namespace

// bf16, 64 / 96 / 128-column blocks (96 / 128: raw sources only); both sources normalised or both raw; 24-bit voxel arithmetic in the staging addresses
bool rs_igemm_kd_supported(const IgemmParams& p, int dtype) {
    if (dtype != RS_BF16 || (p.bn != 64 && p.bn != 96 && p.bn != 128) || p.ntiles % (p.bn / 32)) return false;
    if (p.b.C > 0 && (p.a.mr != nullptr) != (p.b.mr != nullptr)) return false;
    if (p.bn != 64 && p.a.mr != nullptr) return false;
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
