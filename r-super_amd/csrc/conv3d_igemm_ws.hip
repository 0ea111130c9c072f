// 3x3x3 / stride 1 / pad 1 convolution, weight-stationary implicit GEMM on MFMA (gfx950, bf16), channels-last.
//
// Same operation, arguments and fused prologue / epilogues as conv3d_igemm.hip (rsuper_train/model/dim3/conv_layers.py:
// 29-51 ConvNormAct inside BasicBlock :86-94, forward and data gradient), re-laid-out around what the PMC counters of the
// producer/consumer kernel showed (profiles/r02_pmc_conv.md): at one 1-KiB LDS fragment + one 1-KiB L1 weight fragment per
// MFMA plus an epilogue that transposes through LDS, the LDS pipe is as busy as the MFMA pipe (7.4 of 8 cycles per MFMA)
// and neither gets past ~30 %; a first weight-stationary version with one wave per SIMD was instruction-issue bound (one
// wave issues ~1 instruction per 4.4 cycles: ~1000 instructions per 108 MFMAs).
//
// Block = 8 waves (two per SIMD), output tile 4 x 4 x 16 voxels x 32 output columns, persistent over the tiles of a sample.
//   * Wave (d, kh): depth slice d of the tile (two output fragments = h-row pairs (p, p+2), p = 0, 1) and K half kh
//     (channels 16 kh .. 16 kh + 15 of the 32-channel chunk).  Its 27 weight fragments (one per tap) stay in REGISTERS
//     (108 VGPRs); single-chunk layers load them once per block, otherwise the next chunk's fragment is loaded into the
//     same registers right after its last use.  No per-MFMA L1 weight traffic.
//   * Activation fragments are re-used from registers: the halo fragment at (depth D, rows (s, s+2), kw) is the operand of
//     every (output fragment p, tap) with d + kd = D and p + kh = s: 36 LDS fragment reads feed 54 MFMAs per wave and item.
//   * Operand roles are swapped (A = weights, B = activations): the accumulator layout is lane = voxel, registers = output
//     channels; after v_permlane32_swap every lane holds 16-byte channel vectors of its voxel and the epilogue (residual /
//     ReLU mask / statistics / store) runs from registers.  The two K halves of a pair exchange the half of the channels
//     they do not finalise through 4 KB of LDS per wave (16 ds_write_b32 + 16 ds_read_b32) -- the only LDS traffic of the
//     epilogue.  The kh = 1 waves load their weights with the output columns rotated by 16, so both halves run the same
//     code: registers 0..7 are the channels a wave finalises, 8..15 the ones it sends.
//   * All eight waves stage the next item themselves, one dword slice of a 16-byte vector per fragment in the shadow of
//     the MFMAs (issue-early / write-late through registers, norm + ReLU applied while staging, double-buffered halo).
#include "common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HROWS = HD * HH * HW;             // 648 halo rows
constexpr int PITCH = 80;                       // 64 data + 16 pad: conflict-free ds_read_b128 (see row_to_hw)
constexpr int HB = HROWS * PITCH;               // 51840 bytes per halo buffer
constexpr int NVEC = 6;                         // staging vectors per thread: rows (tid >> 2) + 128 i; the 6th only in wave 0
constexpr int XB = 8 * 16 * 64 * 4;             // exchange scratch: [wave][reg][lane] f32
constexpr int KC = 32, KP = 8;
constexpr int NT_THREADS = 512;
#ifndef PRIO_PERIOD
#define PRIO_PERIOD 6
#endif

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mma_wa(f32x16_t& acc, const uint4& w, const uint4& x) {    // acc[cout][voxel] += W^T x
    union { uint4 u; bf16x8_t v; } ua, ub;
    ua.u = w; ub.u = x;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
}

__device__ __forceinline__ void swap32(float& lo, float& hi) {   // lo.lanes[32..63] <-> hi.lanes[0..31]
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}

#ifdef RS_WS_ABLATE
__device__ unsigned long long g_ws_prof[8 * 8];                     // block 0: [wave][loop, sync, epilogue, items, ...] shader cycles
#endif

constexpr int s_order(int i) { return i == 0 ? 0 : i == 1 ? 3 : i == 2 ? 1 : 2; }   // alternates the two accumulators

// EPI: 0 forward, 1 data gradient (ReLU mask + IN-backward sums), 2 forward + residual
// MULTI: more than one K chunk (weights re-streamed per item); NORM: sources carry (mean, rstd) -> fused IN + ReLU prologue
// ABL (measurement builds only, RS_WS_ABLATE): 2 = no staging work, 4 = no MFMAs, 8 = no fragment reads
template <int EPI, bool MULTI, bool NORM, int ABL = 0>
__global__ __launch_bounds__(NT_THREADS, 1) void igemm_ws_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bufs = smem;                                                // 2 x HB
    float* xch = (float*)(smem + 2 * HB);                             // [8][16][64]
    float* nrm_lds = (float*)(smem + 2 * HB + XB);                    // [(Ca + Cb) / 2][sc0, sc1, nb0, nb1]  (NORM)
    float* emr_lds = nrm_lds + 2 * (NORM ? (p.a.C + p.b.C) : 0);      // [32][mean, rstd] of the epilogue source (EPI 1)
    int4* tt_lds = (int4*)(emr_lds + 64);                             // [my_tiles]: (halo base voxel, ~valid-halo mask, tile voxel, d0 | h0 << 10 | w0 << 20 | full << 30)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave & 3, kh_ = wave >> 2;                         // depth slice, K half
    const int n = blockIdx.z;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = MULTI ? nchA + nchB : 1;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nitems = my_tiles * nch;
    const int ntile = blockIdx.y;                                     // this block's 32-column tile
    const int hi = lane >> 5;
    int hs_l, wl_l;
    row_to_hw(lane & 31, hs_l, wl_l);

    if (NORM) {
        // x_hat = max(x * sc + nb, 0) with sc = rstd, nb = -mean * rstd, stored per channel PAIR as (sc0, sc1, nb0, nb1)
        for (int i = tid; i < (p.a.C + p.b.C) / 2; i += NT_THREADS) {
            const int c = 2 * i;
            const float* m = c < p.a.C ? p.a.mr + ((size_t)n * p.a.C + c) * 2 : p.b.mr + ((size_t)n * p.b.C + c - p.a.C) * 2;
            ((float4*)nrm_lds)[i] = make_float4(m[1], m[3], -m[0] * m[1], -m[2] * m[3]);
        }
    }
    if (EPI == 1) {
        for (int i = tid; i < 64; i += NT_THREADS) {
            const int col = blockIdx.y * 32 + (i >> 1);
            float v = (i & 1) ? 1.f : 0.f;
            if (col < p.Cout) v = col < p.ea.C ? p.ea.mr[((size_t)n * p.ea.C + col) * 2 + (i & 1)] : p.eb.mr[((size_t)n * p.eb.C + col - p.ea.C) * 2 + (i & 1)];
            emr_lds[i] = v;
        }
    }

    // Tile table (once per block, vector ALU): the k-th tile of this block in the XCD-aware order of conv3d_igemm.hip (linear
    // workgroup id b runs on XCD b % 8; every XCD gets a contiguous run of tiles), its halo validity as bit ranges
    // (hd in [max(0, 1 - d0), min(5, D - d0)], likewise h, w) and its voxel indices -- no div / mod left in the item loop.
    {
        const bool xcd_remap = (gridDim.x & 7) == 0 && tiles >= 64;
        for (int k = tid; k < my_tiles; k += NT_THREADS) {
            int t = (int)blockIdx.x + k * (int)gridDim.x;
            if (xcd_remap) {
                const int q = tiles >> 3, r = tiles & 7, xcd = t & 7, kk = t >> 3;
                t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
            }
            int tw, th, td;
            rs_tile_coords(t, tiles_w, tiles_h, (p.D + TD - 1) / TD, tw, th, td);
            const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
            auto range = [](int o, int len, int nh) {
                const int lo = o >= 1 ? 0 : 1 - o;
                int hi_ = len - o; if (hi_ > nh - 1) hi_ = nh - 1;
                return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
            };
            const uint32_t tm = range(d0, p.D, HD) | (range(h0, p.H, HH) << 6) | (range(w0, p.W, HW) << 12);
            const int full = (d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W) ? 1 : 0;
            tt_lds[k] = make_int4(((n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1, (int)~tm, ((n * p.D + d0) * p.H + h0) * p.W + w0,
                                  d0 | (h0 << 10) | (w0 << 20) | (full << 30));
        }
    }
    auto tile_entry = [&](int k) {                                    // wave-uniform -> scalar registers
        const int4 e = tt_lds[k < my_tiles ? k : 0];
        return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y), __builtin_amdgcn_readfirstlane(e.z),
                         __builtin_amdgcn_readfirstlane(e.w));
    };

    // ------------------------------------------------------------------ staging state (per-thread constants)
    const int slot = tid & 3;
    int vdelta[NVEC];
    uint32_t pm[NVEC];                                                // one-hot (hd | hh << 6 | hw << 12); bit 30: row past the halo
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int r = (tid >> 2) + 128 * i;
        const int hd = r / (HH * HW);
        const int rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        vdelta[i] = (hd * p.H + hh) * p.W + hw;
        pm[i] = r < HROWS ? ((1u << hd) | (1u << (6 + hh)) | (1u << (12 + hw))) : (1u << 30);
    }
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    const int st_off = (tid >> 2) * PITCH + slot * 16;                // LDS byte offset of this thread's vector 0
    const bool w0_stage = wave == 0;                                  // rows 640..647 (6th vector): threads 0..31 of wave 0

    // context of the item whose global loads are being issued (wave-uniform except cb)
    struct IssueCtx { __amdgpu_buffer_rsrc_t rs; uint32_t rowb, cb, ntm; int base; };
    auto make_ctx = [&](int k, int ch, bool valid) {
        IssueCtx c;
        const bool isB = ch >= nchA;
        const ConvSrc& src = isB ? p.b : p.a;
        const int cc = (isB ? ch - nchA : ch) * KC + slot * KP;
        c.rowb = (uint32_t)src.ld * 2u;
        c.rs = __builtin_amdgcn_make_buffer_rsrc((void*)src.x, 0, nvox_total * c.rowb, 0x00020000);
        c.cb = cc < src.C ? (uint32_t)cc * 2u : 0xFFFFFFFFu;
        const int4 e = tile_entry(k);
        c.ntm = valid ? (uint32_t)e.y : 0xFFFFFFFFu;
        c.base = e.x;
        return c;
    };
    uint4 pre[NVEC];
    uint32_t vmask = 0;                                               // bit i: pre[i] is a real voxel (gets norm + ReLU)
    auto issue_vec = [&](const IssueCtx& c, int i, uint32_t& vm) {
        const bool ok = ((pm[i] & c.ntm) == 0u) && c.cb != 0xFFFFFFFFu;
        const uint32_t off = ok ? (uint32_t)(c.base + vdelta[i]) * c.rowb + c.cb : 0xFFFFFFFFu;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(c.rs, off, 0, 0);    // out-of-range -> zeros, no branch
        pre[i] = make_uint4(q[0], q[1], q[2], q[3]);
        vm |= ok ? (1u << i) : 0u;
    };
    // normalisation constants of this thread's 8 channels of chunk `ch`: LDS byte offset of its 4 (sc0, sc1, nb0, nb1) entries
    auto norm_off = [&](int ch) {
        const bool isB = ch >= nchA;
        const int cc = (isB ? ch - nchA : ch) * KC + slot * KP;
        const int cl = cc < (isB ? p.b.C : p.a.C) ? cc : 0;
        return (((isB ? p.a.C : 0) + cl) / 2) * 16;
    };
    // one dword (two channels) of staged vector i: x_hat pair in place; bit i of vm clear -> padding / out of range -> stays 0
    float4 snr[4];                                                    // (sc0, sc1, nb0, nb1) of this thread's four channel pairs
    auto load_norm = [&](int noff) {
        if (!NORM) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) snr[j] = *(const float4*)((const char*)nrm_lds + noff + j * 16);
    };
    auto norm_dword = [&](int i, int j, uint32_t vm) {
        if (!NORM) return;
        uint32_t* q = j == 0 ? &pre[i].x : j == 1 ? &pre[i].y : j == 2 ? &pre[i].z : &pre[i].w;
        const uint32_t w = *q;
        const float4 sn = snr[j];
        f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        const f32x2_t s2 = {sn.x, sn.y}, b2 = {sn.z, sn.w};
        x = __builtin_elementwise_fma(x, s2, b2);
        i16x2_t v = __builtin_bit_cast(i16x2_t, f2bf2(x[0], x[1]));
        const i16x2_t z = {0, 0};
        v = __builtin_elementwise_max(v, z);
        const uint32_t m = (uint32_t)((int32_t)(vm << (31 - i)) >> 31);
        *q = __builtin_bit_cast(uint32_t, v) & m;
    };
    auto store_vec = [&](char* buf, int i) {
        if (i < NVEC - 1 || tid < 32) *(uint4*)(buf + st_off + i * (128 * PITCH)) = pre[i];      // 6th vector: rows 640..647 only
    };
    auto commit_vec = [&](char* buf, int i, uint32_t vm) {
#pragma unroll
        for (int j = 0; j < 4; ++j) norm_dword(i, j, vm);
        store_vec(buf, i);
    };

    // ------------------------------------------------------------------ weights: 27 fragments (this wave's K half) in registers
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t wstep16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((size_t)p.ntiles * 64 * 16));   // bytes per (chunk, tap, k-step)
    const uint32_t wn_off = (uint32_t)__builtin_amdgcn_readfirstlane(ntile * 1024 + kh_ * (int)((size_t)p.ntiles * 64 * 16));
    // the kh = 1 waves take output column (m + 16) % 32 in fragment row m: their registers 0..7 (after the swaps) are the
    // channels 16..31 they finalise
    const uint32_t lane16 = (uint32_t)((lane & 32) | ((lane + 16 * kh_) & 31)) * 16u;
    uint4 wf[27];
    auto load_w = [&](uint32_t chbase, int tap) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, chbase + (uint32_t)(tap * 2) * wstep16, 0);
        wf[tap] = make_uint4(q[0], q[1], q[2], q[3]);
    };
#pragma unroll
    for (int t = 0; t < 27; ++t) load_w(wn_off, t);

    // ------------------------------------------------------------------ activation fragments
    const int a_base0 = ((wd * HH + hs_l * 2) * HW + wl_l) * PITCH + hi * 16 + kh_ * 32;
    constexpr int NFRAG = 3 * 3 * 4;                                  // (kw, Drel, s)
    auto frag_off = [](int f) {
        const int kw = f / 12, r2 = f % 12, dr = r2 / 4, s = s_order(r2 % 4);
        return ((dr * HH + s) * HW + kw) * PITCH;
    };
#ifndef RS_WS_ADIST
#define RS_WS_ADIST 2
#endif
    constexpr int ADIST = RS_WS_ADIST, AR = ADIST + 1;
    uint4 aq[AR];

    f32x16_t acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float s1[KP], s2[KP];                                             // running statistics of the 8 channels this lane finalises
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    // epilogue constants: the wave finalises channel group kh (16 channels, 8 per lane half) of both fragments of its depth slice
    const uint32_t col = (uint32_t)(ntile * 32 + kh_ * 16 + hi * 8);
    const bool cok = col < (uint32_t)p.Cout;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, nvox_total * (uint32_t)p.ldo * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? p.res : p.out), 0, nvox_total * (uint32_t)(EPI == 2 ? p.ldr : p.ldo) * 2u, 0x00020000);
    float* xw = xch + (wave * 16) * 64 + lane;                        // this wave's slots; the partner's: wave ^ 4
    const float* xr = xch + ((wave ^ 4) * 16) * 64 + lane;
    uint4 ev[2];                                                      // prefetched epilogue operands (residual / forward input)

    // ------------------------------------------------------------------ prologue: item 0 staged synchronously, item 1 in flight
    __syncthreads();                                                  // nrm_lds / emr_lds / tile table visible
    {
        const IssueCtx c0 = make_ctx(0, 0, nitems > 0);
        load_norm(norm_off(0));
        uint32_t vm0 = 0;
#pragma unroll
        for (int i = 0; i < NVEC - 1; ++i) issue_vec(c0, i, vm0);
        if (w0_stage) issue_vec(c0, NVEC - 1, vm0);
#pragma unroll
        for (int i = 0; i < NVEC - 1; ++i) commit_vec(bufs, i, vm0);
        if (w0_stage) { commit_vec(bufs, NVEC - 1, vm0); }
        const IssueCtx c1 = make_ctx(nch > 1 ? 0 : 1, nch > 1 ? 1 : 0, nitems > 1);
        vmask = 0;
#pragma unroll
        for (int i = 0; i < NVEC - 1; ++i) issue_vec(c1, i, vmask);
        if (w0_stage) issue_vec(c1, NVEC - 1, vmask);
    }
    __syncthreads();

    int k_cur = 0, ch_cur = 0;                                        // (tile, chunk) of item `it`
#ifdef RS_WS_ABLATE
    unsigned long long pf_loop = 0, pf_sync = 0, pf_epi = 0;
#endif
    for (int it = 0; it < nitems; ++it) {
#ifdef RS_WS_ABLATE
        const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
        const char* cur = bufs + (it & 1) * HB;
        char* nxt = bufs + ((it + 1) & 1) * HB;
        // (tile, chunk) of items it + 1 and it + 2
        int k1 = k_cur, ch1 = ch_cur + 1;
        if (ch1 == nch) { ch1 = 0; ++k1; }
        int k2 = k1, ch2 = ch1 + 1;
        if (ch2 == nch) { ch2 = 0; ++k2; }
        IssueCtx c2;
        const uint32_t vm_commit = vmask;
        uint32_t vm_issue = 0;
        load_norm(norm_off(ch1));                                    // constants of the chunk committed during this item (dead after the commits)
        const uint32_t wb_nxt = wn_off + (uint32_t)__builtin_amdgcn_readfirstlane(ch1 * 54) * wstep16;
        const bool last_chunk = !MULTI || ch_cur == nch - 1;
        uint32_t vox[2] = {0u, 0u};
        bool vok[2] = {false, false};
        if (last_chunk) {
            const int4 e = tile_entry(k_cur);
            const int d0 = e.w & 1023, h0 = (e.w >> 10) & 1023, w0 = (e.w >> 20) & 1023;
            const bool full = (e.w >> 30) & 1;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int d = d0 + wd, h = h0 + pp + 2 * hs_l, w = w0 + wl_l;
                vok[pp] = cok && (full || (d < p.D && h < p.H && w < p.W));
                vox[pp] = (uint32_t)(e.z + (wd * p.H + pp + 2 * hs_l) * p.W + wl_l);
            }
        }
        auto load_ev = [&]() {                                        // epilogue operands: requested after the commits, consumed after the MFMA loop
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                if (EPI == 2) {                                        // buffer load: out-of-range voxels read zeros
                    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rrs, vok[pp] ? (vox[pp] * (uint32_t)p.ldr + col) * 2u : 0xFFFFFFFFu, 0, 0);
                    ev[pp] = make_uint4(q[0], q[1], q[2], q[3]);
                    continue;
                }
                const bool useb = col >= (uint32_t)p.ea.C;
                const bf16_t* ptr = vok[pp] ? (useb ? (const bf16_t*)p.eb.x + (size_t)(vox[pp] * (uint32_t)p.eb.ld + col - (uint32_t)p.ea.C)
                                                    : (const bf16_t*)p.ea.x + (size_t)(vox[pp] * (uint32_t)p.ea.ld + col))
                                            : (const bf16_t*)p.ea.x;
                ev[pp] = *(const uint4*)ptr;
            }
        };

        // ---------------------------------------------------------------- MFMA stream with the staging in its shadow
        const int a_base = a_base0;
#pragma unroll
        for (int f = 0; f < ADIST; ++f) aq[f] = (ABL & 8) ? make_uint4(tid, 1, 2, 3) : *(const uint4*)(cur + a_base + frag_off(f));
        constexpr int C0 = 1, CX = C0 + 4 * (NVEC - 1) + 1;
        static_assert(CX + 1 < NFRAG, "staging must fit into one item");
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            const int kw = f / 12, r2 = f % 12, dr = r2 / 4, s = s_order(r2 % 4);
            if (f + ADIST < NFRAG && !(ABL & 8)) aq[(f + ADIST) % AR] = *(const uint4*)(cur + a_base + frag_off(f + ADIST));
            // the two waves of a SIMD take turns at the issue arbiter (otherwise the older one runs ahead and the block waits
            // for the younger one at the barrier)
#ifndef WS_PRIO_MODE
#define WS_PRIO_MODE 1
#endif
            if (WS_PRIO_MODE == 0) { if (f % PRIO_PERIOD == 0) { if (((f / PRIO_PERIOD) & 1) ^ kh_) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
            else {                                   // progress-based (conv3d_igemm_kd.hip): the wave that is behind in the item outranks its partner
                if (f == 0) __builtin_amdgcn_s_setprio(3);
                if (f == NFRAG / 4) __builtin_amdgcn_s_setprio(2);
                if (f == NFRAG / 2) __builtin_amdgcn_s_setprio(1);
                if (f == (3 * NFRAG) / 4) __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int kh = s - pp;
                if (kh < 0 || kh > 2) continue;
                if (!(ABL & 4)) mma_wa(acc[pp], wf[(dr * 3 + kh) * 3 + kw], aq[f % AR]);
            }
            if (!(ABL & 2)) {
                // f = 0: tile / chunk context of item it + 2; then per vector i: four dword slices (the 16-byte LDS write with
                // the 4th) and, one fragment later, the global load of the same vector for item it + 2 into the freed registers
                if (f == 0) c2 = make_ctx(k2, ch2, it + 2 < nitems);
                if (f >= C0 && f < C0 + 4 * (NVEC - 1)) {
                    norm_dword((f - C0) / 4, (f - C0) % 4, vm_commit);
                    if ((f - C0) % 4 == 3) store_vec(nxt, (f - C0) / 4);
                }
                if (f >= C0 + 4 && f <= C0 + 4 * (NVEC - 1) && (f - C0) % 4 == 0) issue_vec(c2, (f - C0) / 4 - 1, vm_issue);
                if (f == CX && w0_stage) { commit_vec(nxt, NVEC - 1, vm_commit); issue_vec(c2, NVEC - 1, vm_issue); }
            }
            if (EPI != 0 && f == CX + 1 && last_chunk) load_ev();
            if (MULTI && (f + 1) % 12 == 0) {                         // kw group done: its 9 taps are free for the next chunk
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) load_w(wb_nxt, t9 * 3 + kw);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        vmask = vm_issue;
#ifdef RS_WS_ABLATE
        const unsigned long long pt1 = __builtin_readcyclecounter();
        pf_loop += pt1 - pt0;
#endif

        if (last_chunk) {
            // ------------------------------------------------------------ epilogue from the accumulators
            float a[2][16];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[pp][r] = acc[pp][r];
#pragma unroll
                for (int j = 0; j < 4; ++j) { swap32(a[pp][j], a[pp][4 + j]); swap32(a[pp][8 + j], a[pp][12 + j]); }
            }
            if (!MULTI) __syncthreads();                              // the partner has read the previous tile's exchange slots
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int q = 0; q < 8; ++q) xw[(pp * 8 + q) * 64] = a[pp][8 + q];
            __syncthreads();                                          // exchange visible; item it fully read, item it + 1 fully written
#ifdef RS_WS_ABLATE
            const unsigned long long pt2 = __builtin_readcyclecounter();
            pf_sync += pt2 - pt1;
#endif
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float v[KP];
#pragma unroll
                for (int q = 0; q < KP; ++q) v[q] = a[pp][q] + xr[(pp * 8 + q) * 64];
                const bool ok = vok[pp];
                if (EPI == 2) {
                    float rr[KP];
                    unpack16<bf16_t>(ev[pp], rr);
#pragma unroll
                    for (int q = 0; q < KP; ++q) v[q] += rr[q];
                }
                uint4 pk;
                if (EPI != 1) {
#pragma unroll
                    for (int q = 0; q < KP; ++q) v[q] = ok ? v[q] : 0.f;
                    pk = pack16<bf16_t>(v);
                    float r8[KP];
                    unpack16<bf16_t>(pk, r8);
#pragma unroll
                    for (int q = 0; q < KP; ++q) { s1[q] += r8[q]; s2[q] += r8[q] * r8[q]; }
                } else {
                    float xx[KP], xn[KP];
                    unpack16<bf16_t>(ev[pp], xx);
                    const float4* e4 = (const float4*)(emr_lds + 2 * (kh_ * 16 + hi * 8));
#pragma unroll
                    for (int q = 0; q < KP / 2; ++q) {
                        const float4 t = e4[q];
                        xn[2 * q] = (xx[2 * q] - t.x) * t.y; xn[2 * q + 1] = (xx[2 * q + 1] - t.z) * t.w;
                    }
#pragma unroll
                    for (int q = 0; q < KP; ++q) v[q] = (ok && xn[q] > 0.f) ? v[q] : 0.f;
                    pk = pack16<bf16_t>(v);
                    float r8[KP];
                    unpack16<bf16_t>(pk, r8);
#pragma unroll
                    for (int q = 0; q < KP; ++q) { s1[q] += r8[q]; s2[q] += r8[q] * xn[q]; }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pk), ors,
                                                       ok ? (vox[pp] * (uint32_t)p.ldo + col) * 2u : 0xFFFFFFFFu, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][r] = 0.f;
            }
#ifdef RS_WS_ABLATE
            pf_epi += __builtin_readcyclecounter() - pt2;
#endif
        } else {
            __syncthreads();                                          // item it fully read, item it + 1 fully written
#ifdef RS_WS_ABLATE
            pf_sync += __builtin_readcyclecounter() - pt1;
#endif
        }
        k_cur = k1; ch_cur = ch1;
    }
#ifdef RS_WS_ABLATE
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
        g_ws_prof[wave * 8 + 0] = pf_loop; g_ws_prof[wave * 8 + 1] = pf_sync; g_ws_prof[wave * 8 + 2] = pf_epi; g_ws_prof[wave * 8 + 3] = (unsigned long long)nitems;
    }
#endif

    // ------------------------------------------------------------------ statistics: one partial row per (block, depth slice)
    if (p.part) {
#pragma unroll
        for (int q = 0; q < KP; ++q) {
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { s1[q] += __shfl_xor(s1[q], o, 64); s2[q] += __shfl_xor(s2[q], o, 64); }
        }
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, 0x7FFFFFFF, 0x00020000);
        const bool ok = (lane & 31) == 0 && cok;
        const uint32_t poff = ok ? (uint32_t)(((((size_t)n * gridDim.x + blockIdx.x) * 4 + wd) * p.Cout + col) * 8) : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < KP; q += 2) {
            u32x4_t pv;
            pv[0] = __float_as_uint(s1[q]); pv[1] = __float_as_uint(s2[q]); pv[2] = __float_as_uint(s1[q + 1]); pv[3] = __float_as_uint(s2[q + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(pv, prs, ok ? poff + q * 8 : poff, 0, 0);
        }
    }
}

int ws_grid_x(int tiles, int gy, int N) {                             // ~one persistent block per CU
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}

template <bool MULTI, bool NORM>
int launch_ws(const IgemmParams& p, int epi, hipStream_t st) {
    const int tiles = ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const int gy = p.ntiles;
    const int gx = ws_grid_x(tiles, gy, p.N);
    const size_t smem = 2 * (size_t)HB + XB + (size_t)(NORM ? (p.a.C + p.b.C) : 0) * 8 + 64 * 4 + (size_t)((tiles + gx - 1) / gx) * 16;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    dim3 grid(gx, gy, p.N), block(NT_THREADS);
#define RS_WS_LAUNCH(E, A)                                                                                   \
    {                                                                                                        \
        auto k = igemm_ws_kernel<E, MULTI, NORM, A>;                                                         \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);    \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                                     \
        return rs_check_launch();                                                                            \
    }
#ifdef RS_WS_ABLATE
    struct ProfDump {                                                 // RSUPER_WS_PROF=1: print block 0's phase cycles after every launch (synchronises)
        static void run() {
            unsigned long long h[64];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_prof), sizeof(h));
            for (int w = 0; w < 8; ++w) fprintf(stderr, "ws_prof wave %d: items %llu loop %llu sync %llu epi %llu (cycles per item: %.0f %.0f %.0f)\n", w, h[w * 8 + 3], h[w * 8], h[w * 8 + 1],
                                                h[w * 8 + 2], (double)h[w * 8] / h[w * 8 + 3], (double)h[w * 8 + 1] / h[w * 8 + 3], (double)h[w * 8 + 2] / h[w * 8 + 3]);
        }
    };
    static const int prof = getenv("RSUPER_WS_PROF") ? atoi(getenv("RSUPER_WS_PROF")) : 0;
    struct ProfGuard { int on; ~ProfGuard() { if (on) ProfDump::run(); } } guard{prof};
    static const int abl = getenv("RSUPER_WS_ABL") ? atoi(getenv("RSUPER_WS_ABL")) : 0;
    if (abl && ((!MULTI && NORM && epi == 0 && !p.res) || (MULTI && !NORM && epi == 1))) {
        if (!MULTI) { if (abl == 2) RS_WS_LAUNCH(0, 2) if (abl == 4) RS_WS_LAUNCH(0, 4) if (abl == 8) RS_WS_LAUNCH(0, 8) if (abl == 12) RS_WS_LAUNCH(0, 12) if (abl == 14) RS_WS_LAUNCH(0, 14) }
        else { if (abl == 2) RS_WS_LAUNCH(1, 2) if (abl == 4) RS_WS_LAUNCH(1, 4) if (abl == 8) RS_WS_LAUNCH(1, 8) if (abl == 12) RS_WS_LAUNCH(1, 12) if (abl == 14) RS_WS_LAUNCH(1, 14) }
    }
#endif
    if (epi == 0 && p.res) RS_WS_LAUNCH(2, 0)
    if (epi == 0) RS_WS_LAUNCH(0, 0)
    RS_WS_LAUNCH(1, 0)
#undef RS_WS_LAUNCH
}

}  // namespace

// Weight-stationary kernel: bf16, bn 32, both sources normalised (forward) or both raw (data gradient).
bool rs_igemm_ws_supported(const IgemmParams& p, int dtype, int epi) {
    if (dtype != RS_BF16 || p.bn != 32) return false;
    const bool na = p.a.mr != nullptr, nb = p.b.C > 0 ? p.b.mr != nullptr : na;
    if (na != nb) return false;
    (void)epi;
    return true;
}

int rs_launch_igemm_ws(const IgemmParams& p, int epi, hipStream_t st) {
    const bool multi = (p.a.C + KC - 1) / KC + (p.b.C + KC - 1) / KC > 1;
    const bool norm = p.a.mr != nullptr;
    if (multi) return norm ? launch_ws<true, true>(p, epi, st) : launch_ws<true, false>(p, epi, st);
    return norm ? launch_ws<false, true>(p, epi, st) : launch_ws<false, false>(p, epi, st);
}
