// Volume-fitted implicit-GEMM 3x3x3 convolution for the LOW-RESOLUTION levels (24^3 / 12^3 / 6^3 at batch 2), bf16, gfx950.
//
// Same operation, operands, prologue / epilogue fusion and packed-weight layout as conv3d_igemm.hip (forward and data gradient of
// every nn.Conv3d(k=3, bias=False) of rsuper_train/model/dim3/conv_layers.py:29-51,86-94); what changes is the decomposition:
//
//   * the classic / producer-consumer kernels tile the volume in 4x4x16-voxel bricks: a 24^3 or 12^3 level wastes 25 % of the
//     MFMA rows (W = 24 / 12 against a 16-wide brick), a 6^3 level 79 %, and at batch 2 these levels give 18-144 bricks for
//     256 CUs (profiles/r02_conv_layers_b2_vs_b8.txt: the same kernels run 1.5-2x faster per sample at batch 8);
//   * here a block owns a BOX of TD x TH x TW voxels chosen to divide the volume (4x4x8 for 24^3, 4x4x4 for 12^3) and the GEMM
//     rows are the FLAT voxel index inside the box, so a fragment (32 rows) may span several h-rows / d-planes of the box:
//     no padding rows except in the last fragment of a box;
//   * the four waves of a block split the REDUCTION (27 taps x 2 k-steps of every 32-channel chunk, round-robin), each wave
//     holding the whole MFR x NFR fragment tile: per k-step a wave reads MFR activation fragments from LDS and NFR weight
//     fragments from L2 for MFR x NFR MFMAs (0.75 operand fetches per MFMA for the 4 x 2 tile; a 2 x 2 wave grid over the same
//     tile needs 1.5), no operand is fetched twice by different waves, and M tiles of 64-128 voxels give 216-864 blocks per
//     launch; the four partial accumulators are reduce-scattered through LDS before the fused epilogue.
//
// LDS halo layout: row (hd, hh, hw) of the (TD+2)(TH+2)(TW+2) halo lives at hd*SD + hh*SH + hw*80 bytes with
// SH = TW*80 (mod 256) and SD = TH*SH (mod 256): walking the box in flat order advances the bank offset by exactly 80 bytes
// per voxel ACROSS row and plane boundaries, so any 16 consecutive flat rows (one ds_read_b128 lane group) hit 16 distinct
// 16-byte bank groups -- conflict-free for every box shape.
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int RP = 80;                                   // bytes per halo row: 64 data + 16 pad

template <int TD_, int TH_, int TW_> struct Box {
    static constexpr int TD = TD_, TH = TH_, TW = TW_;
    static constexpr int ROWS = TD * TH * TW;
    static constexpr int MFR = (ROWS + 31) / 32;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HROWS = HD * HH * HW;
    static constexpr int SH = TW * RP + 256;             // >= HW * RP, = TW * RP (mod 256)
    static constexpr int SD = TH * SH + 256 * ((SH + 127) / 128);   // >= HH * SH, = TH * SH (mod 256)
    static constexpr int HALO = HD * SD;
    static constexpr int NVEC = HD;                      // 16-byte staging vectors per thread per chunk: one per halo plane
    static_assert(SH >= HW * RP && SD >= HH * SH, "halo rows must not overlap");
    static_assert(2 * SD + 2 * SH + 2 * RP + 48 < 65536, "tap offsets must fit the ds_read offset field");
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#ifdef RS_BOX_PROF
__device__ unsigned long long g_box_prof[64 * 32];        // [block slot][stamp]
#define BOX_STAMP(k) do { if (lane == 0 && wave == 0 && blockIdx.x % 16 == 0 && blockIdx.x / 16 < 64) g_box_prof[(blockIdx.x / 16) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define BOX_STAMP(k)
#endif

// EPI: 0 forward (optional residual, statistics of the output), 1 data gradient (ReLU mask, InstanceNorm-backward sums)
// SPLIT: the block handles the chunk range [cb, cb + nk) of split blockIdx-derived `sp` and writes the raw f32 tile to the workspace
//        p.ws[sp][n][voxel][Cout] (cross-block split of the reduction for volumes of at most one box: box_splitk_epilogue_kernel finishes)
template <typename B, int NFR, int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void igemm_box_kernel(IgemmParams p, int bd, int bh, int bw) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = 32, KP = 8;
    constexpr int MFR = B::MFR, F = MFR * NFR, BN = NFR * 32;
    constexpr int NVEC = B::NVEC;
    float* mr_lds = (float*)(smem + 2 * B::HALO);          // [Ca + Cb][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave & 1, hsel = wave >> 1;              // this wave's k-step (0/1) and tap-parity phase

    // ---- block -> (sample, column group, box); XCD-contiguous order: linear workgroup id b runs on XCD b % 8, so every XCD
    //      (private L2) gets a contiguous run of boxes of one (sample, column group): shared halo rows and one weight slab per L2
    int L;
    {
        const int nt = gridDim.x, b = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = b & 7, k = b >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int boxes = bd * bh * bw;
    const int ngroups = p.ntiles / NFR;
    const int box = L % boxes; L /= boxes;
    const int sp = SPLIT ? L % p.nsplit : 0;
    if (SPLIT) L /= p.nsplit;
    const int ng = L % ngroups;
    const int n = L / ngroups;
    int t = box;
    const int tw = t % bw; t /= bw;
    const int th = t % bh; t /= bh;
    const int d0 = t * B::TD, h0 = th * B::TH, w0 = tw * B::TW;

    BOX_STAMP(0);
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = nchA + nchB;
    const int cper = SPLIT ? (nch + p.nsplit - 1) / p.nsplit : nch;      // chunks per split
    const int cb = sp * cper;                                  // first chunk of this block; `ch` below counts from it
    const int nk = SPLIT ? (nch - cb < cper ? nch - cb : cper) : nch;
    const bool normA = p.a.mr != nullptr, normB = p.b.mr != nullptr;
    if (normA) for (int i = tid; i < 2 * p.a.C; i += 256) mr_lds[i] = p.a.mr[(size_t)n * 2 * p.a.C + i];
    if (normB) for (int i = tid; i < 2 * p.b.C; i += 256) mr_lds[2 * p.a.C + i] = p.b.mr[(size_t)n * 2 * p.b.C + i];

    // ---- staging geometry: thread t < PL owns position (hh, hw, 16-byte slot) of a halo d-plane and walks the HD planes, so
    //      vector i is voxel vox0 + i*H*W at LDS offset lbase + i*SD: one base register each, plane validity is wave-uniform
    constexpr int PL = B::HH * B::HW * 4;                    // 16-byte vectors per halo plane
    static_assert(PL <= 256 && NVEC == B::HD, "one halo plane per staging round");
    const int slot = tid & 3;
    int vox0, lbase;
    bool hw_ok;
    uint32_t dmask = 0;                                      // bit i: plane d0 - 1 + i lies inside the volume
    {
        const int hh = tid / (B::HW * 4), hw = (tid % (B::HW * 4)) >> 2;
        const int h = h0 - 1 + hh, w = w0 - 1 + hw;
        hw_ok = tid < PL && h >= 0 && h < p.H && w >= 0 && w < p.W;
        vox0 = ((n * p.D + d0 - 1) * p.H + h) * p.W + w;
        lbase = tid < PL ? hh * B::SH + hw * RP + slot * 16 : -1;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) dmask |= (d0 - 1 + i >= 0 && d0 - 1 + i < p.D) ? (1u << i) : 0u;
    }
    int plane = p.H * p.W;
    // The NVEC planes of a chunk travel in two groups (planes [0, NH) and [NH, NVEC)) that share the NH registers of `pre`:
    // group 0 of chunk c + 1 is issued in the middle of chunk c - 1's MFMA phase and written to LDS at the start of chunk c's,
    // group 1 is issued there and written in the middle of chunk c's -- half a chunk of MFMAs covers the load latency, and only
    // NH vectors are live at any time (the full set cost 12 more registers: spills, and a spill reload waits for vmcnt(0)).
    constexpr int NH = (NVEC + 1) / 2;
    uint4 pre[NH];
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    auto issue_to = [&](int ch, int i, uint4& dst) {         // out-of-volume voxels / channel slots past C: hardware zeros
        const int cc = cb + ch;
        const bool isB = cc >= nchA;
        const ConvSrc& src = isB ? p.b : p.a;
        const int c = (isB ? cc - nchA : cc) * KC + slot * KP;
        const uint32_t rowb = (uint32_t)src.ld * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src.x, 0, nvox_total * rowb, 0x00020000);
        const bool ok = hw_ok && ((dmask >> i) & 1u) && c < src.C;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (uint32_t)(vox0 + i * plane) * rowb + (uint32_t)c * 2u : 0xFFFFFFFFu, 0, 0);
        dst = make_uint4(q[0], q[1], q[2], q[3]);
    };
    auto issue_one = [&](int ch, int i) { issue_to(ch, i, pre[i % NH]); };
    auto issue_group = [&](int ch, int g) {
#pragma unroll
        for (int i = g * NH; i < (g ? NVEC : NH); ++i) issue_one(ch, i);
    };
    auto commit_from = [&](int ch, int i0, int i1, const uint4* src_q) {   // planes [i0, i1) of chunk ch (src_q[i - i0]) -> halo buffer ch & 1
        const int cc = cb + ch;
        const bool isB = cc >= nchA;
        const ConvSrc& src = isB ? p.b : p.a;
        const int c = (isB ? cc - nchA : cc) * KC + slot * KP;
        const bool norm = src.mr != nullptr && c < src.C && hw_ok;      // (a cached flag ends up in a spilled VGPR)
        float sc_[KP], nb_[KP];
        if (norm) {
            const float* mr = mr_lds + 2 * ((isB ? p.a.C : 0) + c);
#pragma unroll
            for (int j = 0; j < KP; ++j) { sc_[j] = mr[2 * j + 1]; nb_[j] = -mr[2 * j] * mr[2 * j + 1]; }
        }
        char* buf = smem + (ch & 1) * B::HALO + lbase;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            uint4 q = src_q[i - i0];
            if (norm && ((dmask >> i) & 1u)) q = norm_relu16<T>(q, sc_, nb_);     // padding stays zero AFTER the activation
            if (lbase >= 0) *(uint4*)(buf + i * B::SD) = q;
        }
    };
    auto commit = [&](int ch, int g) { commit_from(ch, g * NH, g ? NVEC : NH, pre); };   // group g of chunk ch (held in pre)

    // ---- A-fragment bases: lane l holds flat row 32 mf + row_hw_packed(l & 31) (lane groups of ds_read_b128 = runs of 16
    //      consecutive flat rows), k half l >> 5; rows past the box clamp to row 0 (their results are never stored)
    int abase[MFR];
#pragma unroll
    for (int mf = 0; mf < MFR; ++mf) {
        int r = 32 * mf + row_hw_packed(lane & 31);
        if (r >= B::ROWS) r = 0;
        const int dd = r / (B::TH * B::TW), hh = (r / B::TW) % B::TH, ww = r % B::TW;
        abase[mf] = dd * B::SD + hh * B::SH + ww * RP + (lane >> 5) * 16 + ks * 32;
    }

    f32x16_t acc[MFR][NFR];

    // ---- weight fragments: buffer loads, per-lane VGPR offset (lane * 16 + nf KiB), wave-uniform SGPR step offsets
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t wstep16 = (uint32_t)__builtin_amdgcn_readfirstlane(p.ntiles * 64 * 16);   // bytes per (chunk, tap, k-step)
    const uint32_t wn_off = (uint32_t)__builtin_amdgcn_readfirstlane(ng * NFR * 1024) + (uint32_t)ks * wstep16;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    constexpr int RB = 3;                                    // weight ring depth (k-steps); 14 + 13 steps per chunk pair = 0 (mod 3)
    uint4 bq[RB][NFR];
    auto load_b = [&](int ch, int tap, uint4* dst) {         // tap is a compile-time constant at every call site
        const uint32_t so = wn_off + ((uint32_t)(cb + ch) * 54u + (uint32_t)tap * 2u) * wstep16;
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + nf * 1024, so, 0);
            dst[nf] = make_uint4(q[0], q[1], q[2], q[3]);
        }
    };
    auto tap_off = [](int tap) {
        const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
        return kd * B::SD + kh * B::SH + kw * RP;
    };

    // One chunk: this wave's taps are T0, T0 + 2, ... (NJ = 14 / 13 of them) at its k-step `ks`; ring slot of step j = (Q0 + j) % 3.
    // On entry halo buffer ch & 1 is complete and `pre` holds group 0 of chunk ch + 1; the chunk ends with the block barrier.
    auto chunk = [&](auto T0c, auto Q0c, int ch) {
        constexpr int T0 = decltype(T0c)::value, Q0 = decltype(Q0c)::value;
        constexpr int NJ = T0 ? 13 : 14;
        // LLVM hoists every per-plane voxel offset and every per-step weight offset (a multiply each) out of the chunk loop and then
        // spills them -- a spill reload waits for vmcnt(0), i.e. for the whole weight ring.  Opaque copies keep them in the loop.
        // (inline-asm results count as divergent: the wave-uniform values are re-derived with readfirstlane, or every weight load
        // turns into a waterfall loop.)
        {
            int pv = plane, wv = (int)wstep16;
            asm volatile("" : "+v"(vox0), "+v"(lbase), "+v"(pv), "+v"(wv));
            plane = __builtin_amdgcn_readfirstlane(pv);
            wstep16 = (uint32_t)__builtin_amdgcn_readfirstlane(wv);
        }
        const int nx = ch + 1 < nk ? ch + 1 : 0;            // weight prefetch past the last chunk wraps (harmless re-load)
        if (ch + 1 < nk) { commit(ch + 1, 0); issue_group(ch + 1, 1); }
        // MFMA phase in units of AU activation fragments: the fragments of unit u + 1 are read from LDS while unit u's MFMAs issue
        constexpr int AU = (MFR % 2 == 0) ? 2 : 1, G = MFR / AU;
        uint4 aq[2][AU];
        auto load_a = [&](int u, uint4* dst) {
            const int off = tap_off(T0 + 2 * (u / G));
#pragma unroll
            for (int i = 0; i < AU; ++i) dst[i] = *(const uint4*)(smem + abase[(u % G) * AU + i] + off);
        };
        load_a(0, aq[0]);
#pragma unroll
        for (int u = 0; u < NJ * G; ++u) {
            const int j = u / G, g = u % G;
            if (u + 1 < NJ * G) load_a(u + 1, aq[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < AU; ++i)
#pragma unroll
                for (int nf = 0; nf < NFR; ++nf) mma32<T>(acc[g * AU + i][nf], aq[u & 1][i], bq[(Q0 + j) % RB][nf]);
            if (g == G - 1 && j == NJ / 2) {
                if (ch + 1 < nk) commit(ch + 1, 1);
                if (ch + 2 < nk) issue_group(ch + 2, 0);
            }
            if (g == G - 1) {
                if (j + RB < NJ) load_b(ch, T0 + 2 * (j + RB), bq[(Q0 + j) % RB]);
                else load_b(nx, (1 - T0) + 2 * (j + RB - NJ), bq[(Q0 + j) % RB]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the other halo buffer next: toggle the fragment bases in place
#pragma unroll
        for (int mf = 0; mf < MFR; ++mf) abase[mf] += (ch & 1) ? -B::HALO : B::HALO;
        __syncthreads();
    };
    auto run = [&](auto Hc) {
        constexpr int H = decltype(Hc)::value;               // tap parity of this wave in even chunks
        constexpr int QB = (H ? 13 : 14) % RB;               // ring phase at the second chunk of a pair
        for (int ch = 0; ch < nk; ch += 2) {
            chunk(std::integral_constant<int, H>{}, std::integral_constant<int, 0>{}, ch);
            if (ch + 1 < nk) chunk(std::integral_constant<int, 1 - H>{}, std::integral_constant<int, QB>{}, ch + 1);
        }
    };

    // ---- prologue: the first weight fragments go out before anything else (L2 latency under the halo staging), then chunk 0 is
    //      staged synchronously with group 0 of chunk 1 in flight
#pragma unroll
    for (int r = 0; r < RB; ++r) load_b(0, hsel + 2 * r, bq[r]);
    {
        uint4 p0[NVEC];                                      // the accumulators are not live yet: all planes of chunk 0 at once
#pragma unroll
        for (int i = 0; i < NVEC; ++i) issue_to(0, i, p0[i]);
        __syncthreads();                                     // mr_lds visible
        commit_from(0, 0, NVEC, p0);
    }
    if (nk > 1) issue_group(1, 0);
    __syncthreads();
#pragma unroll
    for (int mf = 0; mf < MFR; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
    BOX_STAMP(1);
    if (hsel == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    BOX_STAMP(2);

    // ------------------------------------------------------------------ K-partial reduction fused into the epilogue
    // Per group of GM row fragments: every wave writes ITS partial sums voxel-major as f32 ([wave][flat row][BN] + 16 B pad),
    // then all 256 threads walk the rows with 16-byte vectors, add the four partials in wave order (deterministic), apply
    // residual / ReLU mask, round, accumulate the InstanceNorm partial sums and store.  The residual / forward-input vectors of
    // a group are requested before its scratch is written, so their latency hides behind the LDS round trip.
    // (The halo buffers are dead: the last chunk ended with a barrier.)
    constexpr int EPF = BN + 4;
    constexpr int CG = BN / KP;                              // 16-byte column groups per voxel
    constexpr int RPT = 256 / CG;                            // row stride between a thread's vectors
    constexpr int GM = MFR >= 2 ? 2 : 1;                     // row fragments per group: 4 x 64 x EPF floats = 68 KB at BN = 64
    constexpr int NGRP = (MFR + GM - 1) / GM;
    constexpr int GROWS = GM * 32;
    constexpr int NV = GROWS / RPT;
    static_assert(GROWS % RPT == 0, "a group is a whole number of thread passes");
    float* sc2 = (float*)smem;
    const int cg = tid % CG, pr0 = tid / CG;
    const int col0 = ng * BN + cg * KP;
    const bool cok = col0 < p.Cout;
    const bool useb = EPI == 1 && cok && col0 >= p.ea.C;      // columns past Cout keep source a (never loaded from: ok[] is false)
    const ConvSrc& es = useb ? p.eb : p.ea;
    const int ecol0 = useb ? col0 - p.ea.C : (cok ? col0 : 0);
    float emu[KP], ers[KP];
    if (EPI == 1 && cok && !SPLIT) {
#pragma unroll
        for (int j = 0; j < KP; ++j) { emu[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2]; ers[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2 + 1]; }
    }
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const bool need_ld = !SPLIT && (EPI == 1 || p.res != nullptr);
    const T* ld_base = EPI == 1 ? (const T*)es.x + ecol0 : (const T*)p.res + (cok ? col0 : 0);
    const uint32_t ld_ld = EPI == 1 ? (uint32_t)es.ld : (uint32_t)p.ldr;
#pragma unroll
    for (int gq = 0; gq < NGRP; ++gq) {
        uint32_t vox[NV];
        bool ok[NV];
        uint4 ev[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pr = gq * GROWS + pr0 + j * RPT;       // flat row of the box
            const int dd = pr / (B::TH * B::TW), hh = (pr / B::TW) % B::TH, ww = pr % B::TW;
            const int d = d0 + dd, h = h0 + hh, w = w0 + ww;
            ok[j] = cok && pr < B::ROWS && d < p.D && h < p.H && w < p.W;
            vox[j] = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
            ev[j] = make_uint4(0, 0, 0, 0);
            if (need_ld) ev[j] = *(const uint4*)(ok[j] ? ld_base + (size_t)(vox[j] * ld_ld) : ld_base);     // address select, no branch
        }
        if (gq) __syncthreads();                             // previous group's scratch consumed
        {
            const int col_l = lane & 31, hi = lane >> 5;
            float* dst = sc2 + wave * (GROWS * EPF);
#pragma unroll
            for (int m = 0; m < GM; ++m) {
                const int mf = gq * GM + m;
                if (mf < MFR) {
#pragma unroll
                    for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int i0 = (r & 3) + 8 * (r >> 2);
                            const int row = 32 * m + (hi ? row_hw_packed(i0 + 4) : row_hw_packed(i0));
                            dst[row * EPF + nf * 32 + col_l] = acc[mf][nf][r];
                        }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (ok[j]) {
                const int lr = pr0 + j * RPT;                // row inside the group
                float v[KP];
#pragma unroll
                for (int k = 0; k < KP; ++k) v[k] = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const float4* sp4 = (const float4*)(sc2 + w4 * (GROWS * EPF) + lr * EPF + cg * KP);
#pragma unroll
                    for (int k4 = 0; k4 < KP / 4; ++k4) { const float4 t4 = sp4[k4]; v[k4 * 4] += t4.x; v[k4 * 4 + 1] += t4.y; v[k4 * 4 + 2] += t4.z; v[k4 * 4 + 3] += t4.w; }
                }
                if constexpr (SPLIT) {                       // raw f32 tile of this split: the epilogue kernel adds the splits
                    float4* wd = (float4*)(p.ws + ((size_t)sp * nvox_total + vox[j]) * (size_t)p.Cout + col0);
                    wd[0] = make_float4(v[0], v[1], v[2], v[3]);
                    wd[1] = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    if (EPI == 0) {
                        if (p.res) {
                            float rr[KP];
                            unpack16<T>(ev[j], rr);
#pragma unroll
                            for (int k = 0; k < KP; ++k) v[k] += rr[k];
                        }
#pragma unroll
                        for (int k = 0; k < KP; ++k) { v[k] = Elem<T>::rnd(v[k]); s1[k] += v[k]; s2[k] += v[k] * v[k]; }
                    } else {
                        float xx[KP];
                        unpack16<T>(ev[j], xx);
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            const float xn = (xx[k] - emu[k]) * ers[k];
                            v[k] = Elem<T>::rnd(xn > 0.f ? v[k] : 0.f);
                            s1[k] += v[k]; s2[k] += v[k] * xn;
                        }
                    }
                    *(uint4*)((T*)p.out + (size_t)(vox[j] * (uint32_t)p.ldo + (uint32_t)col0)) = pack16<T>(v);
                }
            }
        }
    }
    BOX_STAMP(4);
    if (p.part && !SPLIT) {
        __syncthreads();
        float* red = (float*)smem;                           // [RPT][BN][2]
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            red[(pr0 * BN + cg * KP + k) * 2] = s1[k];
            red[(pr0 * BN + cg * KP + k) * 2 + 1] = s2[k];
        }
        __syncthreads();
        for (int cl = tid; cl < BN; cl += 256) {
            float a = 0.f, b = 0.f;
            for (int m = 0; m < RPT; ++m) { a += red[(m * BN + cl) * 2]; b += red[(m * BN + cl) * 2 + 1]; }
            const int col = ng * BN + cl;
            if (col < p.Cout) {
                float* pp = p.part + (((size_t)n * boxes + box) * p.Cout + col) * 2;
                pp[0] = a; pp[1] = b;
            }
        }
    }
    BOX_STAMP(5);
}

// Second pass of the cross-block split: out = epilogue(sum over the splits of ws[s][voxel][Cout]).  Block = 32 voxels x 64 columns
// (8 column groups of 8), thread = one 16-byte output vector; per-block partial statistics row = voxel block (part rows =
// ceil(vox / 32) per sample), reduced over the 32 voxel threads through LDS in a fixed order.
template <int EPI>
__global__ __launch_bounds__(256) void box_splitk_epilogue_kernel(IgemmParams p) {
    typedef bf16_t T;
    constexpr int KP = 8;
    __shared__ float red[32][64][2];
    const int tid = threadIdx.x, cg = tid & 7, vr = tid >> 3;
    const int n = blockIdx.z, vb = blockIdx.x;
    const int vox_s = p.D * p.H * p.W;
    const int vloc = vb * 32 + vr;
    const int col0 = blockIdx.y * 64 + cg * KP;
    const bool ok = vloc < vox_s && col0 < p.Cout;
    const uint32_t vox = (uint32_t)(n * vox_s + (vloc < vox_s ? vloc : 0));
    const size_t nvox_total = (size_t)p.N * vox_s;
    float v[KP], s1[KP], s2[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { v[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
    if (ok) {
        for (int s = 0; s < p.nsplit; ++s) {
            const float4* w = (const float4*)(p.ws + ((size_t)s * nvox_total + vox) * (size_t)p.Cout + col0);
            const float4 a = w[0], b = w[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (EPI == 0) {
            if (p.res) {
                float rr[KP];
                unpack16<T>(*(const uint4*)((const T*)p.res + (size_t)(vox * (uint32_t)p.ldr + (uint32_t)col0)), rr);
#pragma unroll
                for (int k = 0; k < KP; ++k) v[k] += rr[k];
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) { v[k] = Elem<T>::rnd(v[k]); s1[k] = v[k]; s2[k] = v[k] * v[k]; }
        } else {
            const bool useb = col0 >= p.ea.C;
            const ConvSrc& es = useb ? p.eb : p.ea;
            const int ecol0 = useb ? col0 - p.ea.C : col0;
            float xx[KP];
            unpack16<T>(*(const uint4*)((const T*)es.x + (size_t)(vox * (uint32_t)es.ld + (uint32_t)ecol0)), xx);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float mu = es.mr[((size_t)n * es.C + ecol0 + k) * 2], rs = es.mr[((size_t)n * es.C + ecol0 + k) * 2 + 1];
                const float xn = (xx[k] - mu) * rs;
                v[k] = Elem<T>::rnd(xn > 0.f ? v[k] : 0.f);
                s1[k] = v[k]; s2[k] = v[k] * xn;
            }
        }
        *(uint4*)((T*)p.out + (size_t)(vox * (uint32_t)p.ldo + (uint32_t)col0)) = pack16<T>(v);
    }
    if (p.part) {
#pragma unroll
        for (int k = 0; k < KP; ++k) { red[vr][cg * KP + k][0] = s1[k]; red[vr][cg * KP + k][1] = s2[k]; }
        __syncthreads();
        if (tid < 64) {
            float a = 0.f, b = 0.f;
            for (int m = 0; m < 32; ++m) { a += red[m][tid][0]; b += red[m][tid][1]; }
            const int col = blockIdx.y * 64 + tid;
            if (col < p.Cout) {
                float* pp = p.part + (((size_t)n * gridDim.x + vb) * p.Cout + col) * 2;
                pp[0] = a; pp[1] = b;
            }
        }
    }
}

template <typename B, int NFR, bool SPLIT>
int launch_box(const IgemmParams& p, int epi, hipStream_t st) {
    const int bd = (p.D + B::TD - 1) / B::TD, bh = (p.H + B::TH - 1) / B::TH, bw = (p.W + B::TW - 1) / B::TW;
    if (p.ntiles % NFR) return RS_ERR_ARG;
    const int ngroups = p.ntiles / NFR;
    const size_t main_b = 2 * (size_t)B::HALO + (((size_t)(p.a.C + p.b.C) * 8 + 15) / 16) * 16;
    const size_t epi_b = (size_t)4 * (B::MFR >= 2 ? 64 : 32) * (NFR * 32 + 4) * 4;      // four partial copies of one row group
    const size_t red_b = (size_t)(256 / (NFR * 4)) * NFR * 32 * 2 * 4;
    size_t smem = main_b;
    if (epi_b > smem) smem = epi_b;
    if (red_b > smem) smem = red_b;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    if (SPLIT && (!p.ws || p.nsplit < 1)) return RS_ERR_ARG;
    dim3 grid((unsigned)(bd * bh * bw * ngroups * p.N * (SPLIT ? p.nsplit : 1))), block(256);
    if (epi == 0) {
        auto k = igemm_box_kernel<B, NFR, 0, SPLIT>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p, bd, bh, bw);
    } else {
        auto k = igemm_box_kernel<B, NFR, 1, SPLIT>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p, bd, bh, bw);
    }
#ifdef RS_BOX_PROF
    if (getenv("RSUPER_BOX_PROF")) {
        static unsigned long long h[64 * 32];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_box_prof), sizeof(h));
        for (int b = 0; b < 64; b += 3) {
            if (!h[b * 32]) continue;
            fprintf(stderr, "box_prof blk %4d: prologue %6llu main %7llu epi %6llu stats %6llu | total %7llu\n", b * 16,
                    h[b * 32 + 1] - h[b * 32], h[b * 32 + 2] - h[b * 32 + 1], h[b * 32 + 4] - h[b * 32 + 2], h[b * 32 + 5] - h[b * 32 + 4], h[b * 32 + 5] - h[b * 32]);
        }
        memset(h, 0, sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_box_prof), h, sizeof(h));
    }
#endif
    if (SPLIT) {
        const int vox_s = p.D * p.H * p.W;
        dim3 g2((unsigned)((vox_s + 31) / 32), (unsigned)((p.Cout + 63) / 64), (unsigned)p.N);
        if (epi == 0) hipLaunchKernelGGL(box_splitk_epilogue_kernel<0>, g2, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(box_splitk_epilogue_kernel<1>, g2, dim3(256), 0, st, p);
    }
    return rs_check_launch();
}

typedef Box<4, 4, 8> BoxA;      // 128 voxels, 4 fragments: 24^3-class volumes (W, H multiples of 8 / 4)
typedef Box<4, 4, 4> BoxB;      // 64 voxels, 2 fragments: 12^3-class volumes
typedef Box<6, 6, 6> BoxC;      // 216 voxels, 7 fragments x 32 columns: a whole 6^3 sample per block, reduction split over blocks

}  // namespace

// Box shape for a volume: 0 = none, 1 = 4x4x8 (x 64 columns), 2 = 4x4x4 (x 64 columns).  The shape with the better row
// utilisation wins; on a tie the larger box (half the weight traffic per MFMA) if it still yields a block per CU.
int rs_box_config(int N, int D, int H, int W, int n_cols) {
    auto boxes = [&](int td, int th, int tw) { return ((D + td - 1) / td) * ((H + th - 1) / th) * ((W + tw - 1) / tw); };
    const long vox = (long)D * H * W;
    const int nA = boxes(4, 4, 8), nB = boxes(4, 4, 4);
    const double uA = (double)vox / ((double)nA * 128), uB = (double)vox / ((double)nB * 64);
    const int groups = (n_cols + 63) / 64;
    if (uA >= 0.95 * uB && (long)nA * N * groups >= 256) return 1;
    return 2;
}

// Split factor of config 3 (one 6x6x6 box per sample, 32-column blocks, chunks dealt to `nsplit` blocks): about one block per CU.
int rs_box_nsplit(int N, int n_cols, int nch) {
    const int groups = (n_cols + 31) / 32;
    int s = 256 / (N * groups > 0 ? N * groups : 1);
    if (s < 1) s = 1;
    if (s > nch) s = nch;
    const int cper = (nch + s - 1) / s;
    return (nch + cper - 1) / cper;
}

int rs_box_part_rows(int cfg, int D, int H, int W) {
    if (cfg == 1) return ((D + 3) / 4) * ((H + 3) / 4) * ((W + 7) / 8);
    if (cfg == 3) return (D * H * W + 31) / 32;
    return ((D + 3) / 4) * ((H + 3) / 4) * ((W + 3) / 4);
}

int rs_launch_igemm_box(const IgemmParams& p, int cfg, int epi, hipStream_t st) {
    if (cfg == 3) return p.bn == 32 ? launch_box<BoxC, 1, true>(p, epi, st) : RS_ERR_ARG;
    if (p.bn != 64) return RS_ERR_ARG;
    if (cfg == 1) return launch_box<BoxA, 2, false>(p, epi, st);
    if (cfg == 2) return launch_box<BoxB, 2, false>(p, epi, st);
    return RS_ERR_ARG;
}
