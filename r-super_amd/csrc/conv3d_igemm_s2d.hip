// Data gradient of the 3x3x3 / STRIDE 2 / pad 1 convolution (a transposed convolution) as an implicit GEMM on MFMA (gfx950, bf16, channels-last):
// the persistent kernel for the strided [conv1 | shortcut] GEMM of down_block(pool=False) -- BasicBlock(in, out, stride=2),
// rsuper_train/model/dim3/unet_utils.py:38-39, conv_layers.py:29-38,82-84; autograd of F.conv3d(stride=2) in the reference.  Same operation,
// arguments and epilogue (ReLU mask of the forward input, InstanceNorm-backward sums) as mode 2 of conv3d_igemm_s2.hip, which stays the f32 kernel
// and the A/B reference; the arithmetic is the same parity-class decomposition
//
//     dx[2q + r] = sum_t W[t]^T dy[q + (r + 1 - t) / 2]      per axis: r = 0 takes t = 1 (dy[q]); r = 1 takes t = 2 (dy[q]) and t = 0 (dy[q + 1])
//
// -- 27 (class, tap) pairs, each a plain tap on the half grid at an offset in {0, +1}^3 -- laid out for the machine:
//   * block = 8 matrix waves (two per SIMD) over a 4 x 4 x 16 brick of the half grid; wave = one position fragment (depth q_d, h pair) x ALL 8 classes
//     (8 accumulators = the 8 x 8 x 32 full-resolution voxels 2q + class of its 32 positions) x one 32-column fragment (blockIdx.y walks the columns);
//   * K is staged 16 channels of [dy1 | dOut] at a time: the haloed brick (5 x 5 x 17 rows, 48-byte pitch, 20 KB) AND the item's 27 weight fragments
//     (27 KB) by LDS-DMA (`buffer_load_dwordx4 ... lds`: no registers, no arithmetic, 6 wave instructions per wave and item) into two buffers, ONE
//     barrier per item.  Per item a wave reads its 8 activation fragments (one per offset) and the 27 weight fragments from LDS and issues 27 MFMAs;
//     the parity-class kernel had one 4-wave block per CU (486 registers) and fetched every weight fragment through L1 per wave;
//   * persistent blocks over the bricks of a sample (XCD-aware order), wave-private epilogue through LDS scratch that walks the 8 classes, scatters to
//     2q + class, applies the ReLU mask of the forward input there (operands requested ahead) and accumulates the InstanceNorm-backward sums in
//     registers (one partial row per (block, wave)).
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int TD = 4, TH = 4, TW = 16, HD = TD + 1, HH = TH + 1, HW = TW + 1;
constexpr int HROWS = HD * HH * HW;                 // 425 halo rows
constexpr int PITCH = 48;
constexpr int NPA = 20;                             // 1 KB DMA pieces of the halo image (425 x 48 = 20400 bytes)
constexpr int NPB = 27;                             // weight fragments of an item (1 KB each)
constexpr int A_BYTES = NPA * 1024, HB = (NPA + NPB) * 1024 + 1024;      // buffer stride (one spare KB: piece 47 of the round-robin below is never issued)
constexpr int NP = (NPA + NPB + 7) / 8;             // DMA pieces per wave and item
constexpr int NT = 512, NW = 8;
constexpr int SCR_ROW = 36, SCR_BYTES = 32 * SCR_ROW * 4;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s2d_dma16(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff, uint32_t lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(lds_byte), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void row_to_hw_nt(int i, int& hs, int& w) {
    hs = (int)((0xF00F0FF0u >> i) & 1u);
    const unsigned long long t = i < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
    w = (int)((t >> ((i & 15) * 4)) & 15ull);
}

struct Item { int c, C; uint32_t rowb, nrec; uint64_t base; uint32_t wofs; };
struct Tile { uint32_t base, bad0, bad1, org; };

// the 27 (offset, class, weight fragment) triples in issue order: offsets 0..7 (bit per axis: 0 -> dy[q], 1 -> dy[q + 1]); per axis an offset bit 0 serves
// class bit 0 with fragment 1 and class bit 1 with fragment 0, an offset bit 1 serves class bit 1 with fragment 2 (fragment = 2 - tap: flipped packing)
struct Pair { int off, cls, frag; };
template <int... Is, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }
__host__ __device__ constexpr Pair pair_of(int i) {
    int k = 0;
    for (int o = 0; o < 8; ++o) {
        const int od = (o >> 2) & 1, oh = (o >> 1) & 1, ow = o & 1;
        for (int cd = od; cd < 2; ++cd)
            for (int ch = oh; ch < 2; ++ch)
                for (int cw = ow; cw < 2; ++cw) {
                    if (k == i) {
                        const int fd = od ? 2 : (cd ? 0 : 1), fh = oh ? 2 : (ch ? 0 : 1), fw = ow ? 2 : (cw ? 0 : 1);
                        return Pair{o, cd * 4 + ch * 2 + cw, (fd * 3 + fh) * 3 + fw};
                    }
                    ++k;
                }
    }
    return Pair{0, 0, 0};
}

// p.D/H/W = the half-resolution grid (the dy sources p.a | p.b live there), FD/FH/FW the full-resolution one (p.out, p.ea).
__global__ __launch_bounds__(NT, 2) void igemm_s2d_kernel(IgemmParams p, int FD, int FH, int FW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto U = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto UP = [](const void* q) {
        const uint64_t a = (uint64_t)q;
        return (const void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
    };
    const int Ca = U(p.a.C), Cb = U(p.b.C), lda = U(p.a.ld), ldb = U(p.b.ld);
    const void* const xa = UP(p.a.x); const void* const xb = UP(p.b.x);
    const int eC = U(p.ea.C), eld = U(p.ea.ld);
    const bf16_t* const ex = (const bf16_t*)UP(p.ea.x);
    const float* const emra = (const float*)UP(p.ea.mr);
    const int pD = p.D, pH = p.H, pW = p.W, pN = p.N, Cout = p.Cout, ldo = p.ldo, ntiles = p.ntiles;
    const void* const wpk = p.wp; void* const outp = p.out; float* const partp = p.part;
    char* bufs = smem;                                                  // 2 x HB: [halo image | 27 weight fragments]
    float* emr = (float*)(smem + 2 * HB);                               // [32][mean, rstd] of the forward input, this block's columns
    char* scr_base = (char*)(emr + 64);
    uint4* dtab = (uint4*)(scr_base + NW * SCR_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qd = wave >> 1, hp = wave & 1;                            // position fragment of this wave
    const int n = blockIdx.z;
    const int tiles_w = (pW + TW - 1) / TW, tiles_h = (pH + TH - 1) / TH, tiles_d = (pD + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int gx = (int)gridDim.x;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / gx + 1 : 0;
    const int nkA = (Ca + 15) / 16, nkB = (Cb + 15) / 16, nk = nkA + nkB;
    const int nchA = (Ca + 31) / 32;
    const int nitems = my_tiles * nk;
    const uint32_t tapstride = (uint32_t)ntiles * 2048u;
    const uint32_t nvox_src = (uint32_t)(pN * pD * pH * pW), nvox_out = (uint32_t)(pN * FD * FH * FW);
    const uint32_t rowbA = (uint32_t)lda * 2u, rowbB = (uint32_t)ldb * 2u;
    const uint32_t nrecA = nvox_src * rowbA, nrecB = Cb ? nvox_src * rowbB : 0u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int ntile0 = blockIdx.y;                                      // this block's 32-column fragment

    for (int i = tid; i < 64; i += NT) {
        const int col = ntile0 * 32 + (i >> 1);
        emr[i] = col < Cout ? emra[((size_t)n * eC + col) * 2 + (i & 1)] : ((i & 1) ? 1.f : 0.f);
    }
    {
        // entry k = k-th brick of this block (XCD-aware order): (voxel of the brick origin, ~valid (hd | hh << 5) | bit 31, ~valid hw, d0 | h0 << 10 | w0 << 20)
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
            auto bad = [](int o, int len, int nh) {                     // bits i in [0, nh) with o + i >= len   (closed form: the loop crashed instruction selection)
                const int nv = len - o;                                  // positions inside
                const uint32_t all = (1u << nh) - 1u;
                return nv >= nh ? 0u : (nv <= 0 ? all : (all & ~((1u << nv) - 1u)));
            };
            const int lD = live ? pD : 0, lH = live ? pH : 0, lW = live ? pW : 0;     // a dead entry: nothing is inside
            const uint32_t b0 = bad(d0, lD, HD) | (bad(h0, lH, HH) << 5) | 0x80000000u, b1 = bad(w0, lW, HW);
            uint32_t vb = (uint32_t)(((n * pD + d0) * pH + h0) * pW + w0), og = (uint32_t)(d0 | (h0 << 10) | (w0 << 20));
            asm volatile("" : "+v"(vb), "+v"(og));                       // (ROCm 7.2 clang crashes in instruction selection on the plain form of this store)
            dtab[k] = make_uint4(vb, b0, b1, og);
        }
    }
    auto fetch_tile = [&](int k) {
        const uint4 v = dtab[k];
        Tile t;
        t.base = __builtin_amdgcn_readfirstlane(v.x); t.bad0 = __builtin_amdgcn_readfirstlane(v.y);
        t.bad1 = __builtin_amdgcn_readfirstlane(v.z); t.org = __builtin_amdgcn_readfirstlane(v.w);
        return t;
    };
    auto item_of = [&](int j) {                                         // j-th 16-channel slice of [a | b] (mask arithmetic: conv3d_igemm_kd.hip)
        Item it;
        const uint32_t mb = j >= nkA ? 0xFFFFFFFFu : 0u;
        const uint64_t mb64 = j >= nkA ? ~0ull : 0ull;
        const int jj = j - (int)((uint32_t)nkA & mb);
        it.c = jj * 16;
        it.C = (int)((uint32_t)Ca ^ (((uint32_t)Ca ^ (uint32_t)Cb) & mb));
        it.rowb = rowbA ^ ((rowbA ^ rowbB) & mb);
        it.nrec = nrecA ^ ((nrecA ^ nrecB) & mb);
        it.base = (uint64_t)xa ^ (((uint64_t)xa ^ (uint64_t)xb) & mb64);
        const int ch = (int)((uint32_t)nchA & mb) + (jj >> 1);
        it.wofs = (uint32_t)((ch * 54 + (jj & 1)) * ntiles) * 1024u + (uint32_t)ntile0 * 1024u;
        return it;
    };

    // ---- staging by LDS-DMA: piece q = wave + 8 k of a buffer.  q < 20: bytes [1024 q, 1024 q + 1024) of the halo image -- lane l moves 16-byte unit
    //      u = 64 q + l = (halo row u / 3, slot u % 3; slot 2 is the pad of the 48-byte pitch, rows >= 425 do not exist: out-of-range offset, zeros).
    //      20 <= q < 47: weight fragment q - 20 of the item (lane-linear, as packed).
    uint32_t dvo[NP];                                                    // halo pieces: voxel offset of the unit's row | slot << 24 | invalid << 25
    auto dma_geom = [&](int k, int& hd, int& hh, int& hw, int& slot, bool& valid) {
        const int u = 64 * (wave + 8 * k) + lane;
        const int r = u / 3;
        slot = u - 3 * r;
        hd = r / (HH * HW);
        const int rem = r - hd * (HH * HW);
        hh = rem / HW; hw = rem - hh * HW;
        valid = r < HROWS && slot < 2;
    };
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        int hd, hh, hw, slot; bool valid;
        dma_geom(k, hd, hh, hw, slot, valid);
        dvo[k] = valid ? ((uint32_t)((hd * pH + hh) * pW + hw) | ((uint32_t)slot << 24)) : (1u << 25);
    }
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, 0x7FFFFFFF, 0x00020000);
    auto dma_item = [&](const Tile& t, const Item& it, uint32_t buf_off) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)it.base, 0, it.nrec, 0x00020000);
        uint32_t lbase = lds0 + buf_off + (uint32_t)wave * 1024u;
        asm volatile("" : "+s"(lbase));
        const bool boundary = ((t.bad0 & 0x3FFu) | t.bad1) != 0u;        // wave-uniform: a brick that touches a face of the volume (or a dead entry)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int q = wave + 8 * k;                                  // wave-uniform
            if (q >= NPA + NPB) continue;
            if (q < NPA) {
                const uint32_t v = dvo[k];
                const uint32_t slot = (v >> 24) & 1u;
                bool ok = (v >> 25) == 0u && it.c + (int)slot * 8 < it.C;
                if (boundary) {
                    int hd, hh, hw, sl; bool valid;
                    dma_geom(k, hd, hh, hw, sl, valid);
                    ok = ok && (((t.bad0 >> hd) | (t.bad0 >> (5 + hh)) | (t.bad1 >> hw)) & 1u) == 0u;
                }
                const uint32_t voff = ok ? __umul24(t.base + (v & 0xFFFFFFu), it.rowb) + slot * 16u : 0xFFFFFFF0u;
                s2d_dma16(rs, voff, (uint32_t)(it.c * 2), lbase + (uint32_t)k * 8192u);
            } else {
                s2d_dma16(wrs, (uint32_t)lane * 16u, it.wofs + (uint32_t)(q - NPA) * tapstride, lbase + (uint32_t)k * 8192u);
            }
        }
    };

    // A fragments: lane -> (row of the h pair, w) by row_to_hw, 16-byte half lane >> 5; offset o adds ((od * HH + oh) * HW + ow) rows
    int hs, wl;
    row_to_hw_nt(lane & 31, hs, wl);
    const int a_lane = ((qd * HH + 2 * hp + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
    const int b_lane = A_BYTES + lane * 16;
    // epilogue geometry: lane -> 16-byte column group cg of rows er0, er0 + 16 of a fragment
    const int cg = lane & 3, er0 = lane >> 2;
    int rhs[2], rw_[2];
    row_to_hw_nt(er0, rhs[0], rw_[0]);
    row_to_hw_nt(er0 + 16, rhs[1], rw_[1]);
    float* scr = (float*)(scr_base + wave * SCR_BYTES);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(outp, 0, nvox_out * (uint32_t)ldo * 2u, 0x00020000);
    const int col0 = ntile0 * 32 + cg * 8;                               // first output column of this lane's vectors
    const bool cok = col0 < Cout;

    f32x16_t acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x2_t rs1[4], rs2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { rs1[q] = f32x2_t{0.f, 0.f}; rs2[q] = f32x2_t{0.f, 0.f}; }

    // ---- prologue: item 0 staged synchronously into buffer 0
    __syncthreads();                                                     // tables
    int k1 = 0, j1 = 0;                                                  // (tile, slice) of the NEXT item
    Tile t1 = fetch_tile(0);
    Item i1 = item_of(0);
    dma_item(t1, i1, 0u);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    if (++j1 == nk) { j1 = 0; ++k1; t1 = fetch_tile(k1); }
    i1 = item_of(j1);
    __syncthreads();

    int kc = 0, jc = 0;
    Tile tc = fetch_tile(0);
    for (int it = 0; it < nitems; ++it) {
        const char* buf = bufs + (it & 1) * HB;
        const bool last = jc == nk - 1;                                  // this item completes a brick
        dma_item(t1, i1, (uint32_t)((it + 1) & 1) * HB);                 // item it + 1 into the other buffer
        int k2 = k1, j2 = j1 + 1;
        if (j2 == nk) { j2 = 0; ++k2; }
        // ---- 27 MFMAs: activation fragments one offset ahead, weight fragments WD pairs ahead
        constexpr int WD = 4, WR = WD + 1;
        uint4 aq[2], bq[WR];
        auto fetch_a = [&](int o) {
            return *(const uint4*)(buf + a_lane + ((((o >> 2) & 1) * HH + ((o >> 1) & 1)) * HW + (o & 1)) * PITCH);
        };
        aq[0] = fetch_a(0);
        sfor<WD>([&](auto I) {
            constexpr int i = decltype(I)::value;
            bq[i % WR] = *(const uint4*)(buf + b_lane + pair_of(i).frag * 1024);
        });
        __builtin_amdgcn_s_setprio(3);
        sfor<27>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr Pair pr = pair_of(i);
            if (i == 7) __builtin_amdgcn_s_setprio(2);
            if (i == 14) __builtin_amdgcn_s_setprio(1);
            if (i == 21) __builtin_amdgcn_s_setprio(0);
            if constexpr (i + WD < 27) bq[(i + WD) % WR] = *(const uint4*)(buf + b_lane + pair_of(i + WD).frag * 1024);
            // first pair of an offset: request the next offset's fragment
            if constexpr ((i == 0 || pair_of(i > 0 ? i - 1 : 0).off != pr.off) && pr.off + 1 < 8) aq[(pr.off + 1) & 1] = fetch_a(pr.off + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma32<bf16_t>(acc[pr.cls], aq[pr.off & 1], bq[i % WR]);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (last) {
            // -------------------------------------------------------------- wave-private epilogue of this brick: the 8 classes of this wave's 32 positions
            const int hi = lane >> 5, col_l = lane & 31;
            const int d0 = tc.org & 1023, h0 = (tc.org >> 10) & 1023, w0 = (int)(tc.org >> 20);
            float emu[8], ers[8];
            {
                const float4* e4 = (const float4*)(emr + cg * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float4 v = e4[j]; emu[2 * j] = v.x; ers[2 * j] = v.y; emu[2 * j + 1] = v.z; ers[2 * j + 1] = v.w; }
            }
            auto out_vox = [&](int ps, int c, uint32_t& vx) {            // full-resolution voxel of rows er0 + 16 ps of class c; false: outside the volume
                const int d = 2 * (d0 + qd) + ((c >> 2) & 1), h = 2 * (h0 + 2 * hp + rhs[ps]) + ((c >> 1) & 1), w = 2 * (w0 + rw_[ps]) + (c & 1);
                vx = (uint32_t)(((n * FD + d) * FH + h) * FW + w);
                return d < FD && h < FH && w < FW;
            };
            uint4 ev[2][2];
            auto epi_load = [&](int c) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    uint32_t vx;
                    const bool ok = out_vox(ps, c, vx) && cok;
                    const bf16_t* ptr = ok ? ex + (size_t)(vx * (uint32_t)eld + (uint32_t)col0) : ex;
                    ev[c & 1][ps] = *(const uint4*)ptr;
                }
            };
            epi_load(0);
            epi_load(1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * SCR_ROW + col_l] = acc[c][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int row = er0 + 16 * ps;
                    const float4* sp = (const float4*)(scr + row * SCR_ROW + cg * 8);
                    const float4 ta = sp[0], tb = sp[1];
                    const f32x2_t v2[4] = {{ta.x, ta.y}, {ta.z, ta.w}, {tb.x, tb.y}, {tb.z, tb.w}};
                    uint32_t vx;
                    const bool ok = out_vox(ps, c, vx) && cok;
                    const uint4 e4 = ev[c & 1][ps];
                    const uint32_t ew[4] = {e4.x, e4.y, e4.z, e4.w};
                    uint32_t ow[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2_t v = v2[q];
                        const f32x2_t xx = {__uint_as_float(ew[q] << 16), __uint_as_float(ew[q] & 0xffff0000u)};
                        const f32x2_t nm = {-emu[2 * q], -emu[2 * q + 1]}, rs = {ers[2 * q], ers[2 * q + 1]};
                        const f32x2_t xn = (xx + nm) * rs;               // == (x - mean) * rstd, the expression every data-gradient epilogue evaluates
                        const float a0 = (ok && xn[0] > 0.f) ? v[0] : 0.f, a1 = (ok && xn[1] > 0.f) ? v[1] : 0.f;
                        const uint32_t wv = f2bf2(a0, a1);
                        ow[q] = wv;
                        const f32x2_t r = {__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u)};
                        rs1[q] = rs1[q] + r;
                        rs2[q] = __builtin_elementwise_fma(r, xn, rs2[q]);
                    }
                    const u32x4_t pk = {ow[0], ow[1], ow[2], ow[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(pk, ors, ok ? (vx * (uint32_t)ldo + (uint32_t)col0) * 2u : 0xFFFFFFF0u, 0, 0);
                }
                if (c + 2 < 8) epi_load(c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");               // this item's DMA pieces (and a brick's stores) have landed
        __syncthreads();                                                 // item it consumed, item it + 1 complete in the other buffer
        if (++jc == nk) { jc = 0; ++kc; tc = fetch_tile(kc); }
        if (j2 == 0) t1 = fetch_tile(k2);
        i1 = item_of(j2);
        k1 = k2; j1 = j2;
    }
    // InstanceNorm-backward sums: ONE partial row per (block, wave)
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
            const uint32_t poff = cok ? (uint32_t)(((((size_t)n * gx + blockIdx.x) * NW + wave) * Cout + col0) * 8) : 0xFFFFFFF0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4_t pv;
                pv[0] = __float_as_uint(rs1[q][0]); pv[1] = __float_as_uint(rs2[q][0]); pv[2] = __float_as_uint(rs1[q][1]); pv[3] = __float_as_uint(rs2[q][1]);
                __builtin_amdgcn_raw_buffer_store_b128(pv, prs, poff == 0xFFFFFFF0u ? poff : poff + q * 16, 0, 0);
            }
        }
    }
}

int s2d_tiles(int D, int H, int W) { return ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }
int s2d_grid_x(int tiles, int gy, int N) {
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}
size_t s2d_smem(int tiles, int gx) { return 2 * (size_t)HB + 256 + NW * (size_t)SCR_BYTES + ((tiles + gx - 1) / gx + 4) * 16; }

}  // namespace

// data gradient, bf16, raw dy sources whose channels are multiples of 16; p.Cout = the forward input's channels
bool rs_igemm_s2d_supported(const IgemmParams& p, int dtype, int FD, int FH, int FW) {
    if (dtype != RS_BF16 || p.a.mr || p.b.mr || !p.ea.mr || !p.ea.x) return false;
    if ((p.a.C % 16) || (p.b.C % 16) || p.a.C < 16) return false;
    if (p.D > 1020 || p.H > 1020 || p.W > 4000) return false;
    if ((unsigned long long)p.N * FD * FH * FW >= (1ull << 24)) return false;     // __umul24 of the voxel index: the data gradient writes and masks on the FULL grid
    const int tiles = s2d_tiles(p.D, p.H, p.W), gy = (p.Cout + 31) / 32;
    return s2d_smem(tiles, s2d_grid_x(tiles, gy, p.N)) <= 160 * 1024 && gy <= p.ntiles;
}

int rs_igemm_s2d_part_rows(int n_cols, int N, int D, int H, int W) {
    return s2d_grid_x(s2d_tiles(D, H, W), (n_cols + 31) / 32, N) * NW;
}

int rs_launch_igemm_s2d(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st) {
    const int tiles = s2d_tiles(p.D, p.H, p.W), gy = (p.Cout + 31) / 32;
    const int gx = s2d_grid_x(tiles, gy, p.N);
    const size_t smem = s2d_smem(tiles, gx);
    (void)hipFuncSetAttribute((const void*)igemm_s2d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(igemm_s2d_kernel, dim3(gx, gy, p.N), dim3(NT), smem, st, p, FD, FH, FW);
    return rs_check_launch();
}
