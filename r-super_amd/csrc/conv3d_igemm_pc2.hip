// 3x3x3 / stride 1 / pad 1 convolution: producer / consumer implicit GEMM, second generation (gfx950, bf16).
//
// Same operation, arguments and fused prologue / epilogues as conv3d_igemm.hip (rsuper_train/model/dim3/conv_layers.py:
// 29-51 ConvNormAct inside BasicBlock :86-94, forward and data gradient).  The producer half (waves 4-7: global loads,
// InstanceNorm + ReLU, LDS writes, two items of loads in flight) is the one of igemm_pc_kernel; the consumer half is rebuilt
// from what round 2 measured on that kernel (profiles/r02_pmc_conv.md, tools/ubench):
//   * LDS as busy as the matrix pipe (6 - 7.4 LDS cycles per MFMA): one 1-KiB activation fragment per MFMA plus an epilogue
//     that transposes every accumulator through LDS.  Here a consumer wave owns output fragments (d, p) = depth slice x
//     h-row pair (p, p + 2); the halo fragment at (depth D, rows (s, s + 2), kw, k-step) is the operand of every (output
//     fragment, tap) with d + kd = D and p + kh = s, so 96 (64-column blocks: 4 fragments per wave) or 72 (32-column
//     blocks: 2 fragments) fragment reads feed 216 / 108 MFMAs per wave and item instead of 216 / 108.
//   * The epilogue runs from registers: operand roles are swapped (A = weights, B = activations), the accumulator layout is
//     lane = voxel, registers = output channels, and after 8 v_permlane32_swap every lane holds two 16-byte channel vectors
//     of its voxel -- residual / ReLU mask / statistics / store without touching LDS.
//   * Weight fragments: the 9 (kd, kh) taps of one (kw, k-step) group sit in registers (36 VGPRs) while the next group's 9
//     are loading into a second set; every fragment is fetched once per item and wave.
#include "common.hpp"
#include "kernels.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HROWS = HD * HH * HW;             // 648 halo rows
constexpr int PITCH = 80;                       // bytes per LDS row: 64 data + 16 pad (conflict-free b128, see row_to_hw)
constexpr int HALO_BYTES = HROWS * PITCH;       // 51840
constexpr int HB = 11 * 64 * PITCH;             // halo buffer padded to 704 rows: every producer thread's 11th vector has a home (56320)
constexpr int NVEC = (HROWS * 4 + 255) / 256;   // 16-byte vectors per producer thread per item (11)
constexpr int KC = 32, KP = 8;
#ifndef PC2_PRODUCER_PRIO
#define PC2_PRODUCER_PRIO 2
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
#ifdef RS_PC2_PROF
__device__ unsigned long long g_pc2_prof[8 * 8];                     // block 0: consumers [loop, epilogue, barrier], producers [commit, issue, barrier]
#define PC2_T(x) const unsigned long long x = __builtin_readcyclecounter();
#define PC2_ACC(dst, a, b) dst += (b) - (a);
#else
#define PC2_T(x)
#define PC2_ACC(dst, a, b)
#endif
constexpr int s_order(int i) { return i == 0 ? 0 : i == 1 ? 3 : i == 2 ? 1 : 2; }   // alternates the h-pair accumulators

// NT: 32-column tiles per block.  1: consumer wave w = depth slice w (2 fragments);  2: consumer wave w = (depth pair w >> 1,
// column tile w & 1) (4 fragments).  EPI: 0 forward, 1 data gradient, 2 forward + residual.
template <int NT, int EPI>
__global__ __launch_bounds__(512, 1) void igemm_pc2_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bufs = smem;                                                // 2 x HB
    float* mr_lds = (float*)(smem + 2 * HB);                  // [Ca + Cb][2] prologue statistics
    float* emr_lds = mr_lds + 2 * (p.a.C + p.b.C);                    // [32 * NT][2] epilogue statistics (EPI 1)
    constexpr int ND = NT == 1 ? 1 : 2;                               // depth slices per consumer wave
    constexpr int NDR = ND + 2;                                       // halo depth slices a consumer wave reads

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int n = blockIdx.z;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = nchA + nchB;
    const bool normA = p.a.mr != nullptr, normB = p.b.mr != nullptr;
    const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nitems = my_tiles * nch;

    if (normA) for (int i = tid; i < 2 * p.a.C; i += 512) mr_lds[i] = p.a.mr[(size_t)n * 2 * p.a.C + i];
    if (normB) for (int i = tid; i < 2 * p.b.C; i += 512) mr_lds[2 * p.a.C + i] = p.b.mr[(size_t)n * 2 * p.b.C + i];
    if (EPI == 1) {
        for (int i = tid; i < 64 * NT; i += 512) {
            const int col = blockIdx.y * 32 * NT + (i >> 1);
            float v = (i & 1) ? 1.f : 0.f;
            if (col < p.Cout) v = col < p.ea.C ? p.ea.mr[((size_t)n * p.ea.C + col) * 2 + (i & 1)] : p.eb.mr[((size_t)n * p.eb.C + col - p.ea.C) * 2 + (i & 1)];
            emr_lds[i] = v;
        }
    }
    __syncthreads();

    // XCD-aware tile order (same bijection as conv3d_igemm.hip)
    const bool xcd_remap = (gridDim.x & 7) == 0 && tiles >= 64;
    auto tile_origin = [&](int k, int& d0, int& h0, int& w0) {
        int t = (int)blockIdx.x + k * (int)gridDim.x;
        if (xcd_remap) {
            const int q = tiles >> 3, r = tiles & 7, xcd = t & 7, kk = t >> 3;
            t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
        }
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        d0 = t * TD; h0 = th * TH; w0 = tw * TW;
    };

    if (producer) {
        // ------------------------------------------------------------------ producer waves
        // Straight-line and branch-free per item: with a branch around the loads (tile geometry, "is there a next item") the
        // compiler can no longer count the loads in flight and waits for ALL of them (s_waitcnt vmcnt(0)) before the first
        // LDS write of an item -- i.e. for the loads issued a moment ago, not only for the item being written: the full
        // memory latency (6 - 8.5k cycles per item, RS_PC2_PROF) sat in the producers of igemm_pc_kernel.  Bounds are bit
        // tests against per-tile masks, items past the end load nothing (all offsets out of range) and write zeros into
        // the buffer nobody reads.
        // The producers share their SIMDs with the (older) consumer waves, which win every issue arbitration: RS_PC2_PROF showed
        // the ~330 staging instructions of an item taking 7 - 11k cycles (the data had long arrived) and the consumers then
        // waiting at the barrier.  With a higher priority the producers take the ~1.3k issue cycles they need when they need them.
        __builtin_amdgcn_s_setprio(PC2_PRODUCER_PRIO);
        const int ptid = tid - 256;
        const int slot = ptid & 3;
        const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
        int vdelta[NVEC];
        uint32_t pm[NVEC];                                            // one-hot (hd | hh << 6 | hw << 12); bit 30: row past the halo
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int r = (ptid >> 2) + 64 * i;
            const int hd = r / (HH * HW);
            const int rem = r - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            vdelta[i] = (hd * p.H + hh) * p.W + hw;
            pm[i] = r < HROWS ? ((1u << hd) | (1u << (6 + hh)) | (1u << (12 + hw))) : (1u << 30);
        }
        const int st_off = (ptid >> 2) * PITCH + slot * 16;
        uint4 preA[NVEC], preB[NVEC];                                 // two items in flight
        uint32_t vmA = 0, vmB = 0;
        bool fastA = false, fastB = false;                            // item lies inside the volume: no padding to keep at zero
        auto issue = [&](int it, uint4* pre, uint32_t& vmask_pre, bool& fast) {
            const bool valid = it < nitems;
            const int itc = valid ? it : 0;
            const int k = itc / nch, ch = itc - k * nch;
            const bool isB = ch >= nchA;
            const ConvSrc& src = isB ? p.b : p.a;
            const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
            const uint32_t rowb = (uint32_t)src.ld * 2u;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src.x, 0, nvox_total * rowb, 0x00020000);
            const uint32_t cb = c < src.C ? (uint32_t)c * 2u : 0xFFFFFFFFu;
            int d0, h0, w0;
            tile_origin(k, d0, h0, w0);
            auto range = [](int o, int len, int nh) {                 // valid halo coordinates as a bit range (scalar ALU)
                const int lo = o >= 1 ? 0 : 1 - o;
                int hi_ = len - o; if (hi_ > nh - 1) hi_ = nh - 1;
                return hi_ < lo ? 0u : (((2u << hi_) - 1u) & ~((1u << lo) - 1u));
            };
            const uint32_t tm = range(d0, p.D, HD) | (range(h0, p.H, HH) << 6) | (range(w0, p.W, HW) << 12);
            const uint32_t ntm = valid ? ~tm : 0xFFFFFFFFu;
            const int base = ((n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1;
            fast = valid && tm == 0x3FFFFFFFu && (src.C & 31) == 0;   // whole halo and every channel slot valid (wave-uniform)
            vmask_pre = 0;
#pragma unroll
            for (int i = 0; i < NVEC; ++i) {
                const bool ok = ((pm[i] & ntm) == 0u) && cb != 0xFFFFFFFFu;
                const uint32_t off = ok ? (uint32_t)(base + vdelta[i]) * rowb + cb : 0xFFFFFFFFu;
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);   // out of range -> zeros
                pre[i] = make_uint4(q[0], q[1], q[2], q[3]);
                vmask_pre |= ok ? (1u << i) : 0u;
            }
        };
        auto commit = [&](int it, const uint4* pre, const uint32_t vmask_pre, const bool fast) {   // writes item `it` (held in pre) into buffer it & 1
            const int itc = it < nitems ? it : 0;
            const int ch = itc % nch;
            const bool isB = ch >= nchA;
            const ConvSrc& src = isB ? p.b : p.a;
            const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
            const bool norm = (isB ? normB : normA);                  // wave-uniform
            const int cl = c < src.C ? c : 0;
            float sc_[KP], nb_[KP];
            const float* mr = mr_lds + 2 * ((isB ? p.a.C : 0) + cl);
#pragma unroll
            for (int j = 0; j < KP; ++j) { sc_[j] = norm ? mr[2 * j + 1] : 1.f; nb_[j] = norm ? -mr[2 * j] * mr[2 * j + 1] : 0.f; }
            char* lds_st = bufs + (it & 1) * HB + st_off;
            if (norm && fast) {
                // interior item: nothing to mask (rows past the halo hold normalised zeros in the pad rows nobody reads);
                // the producers' VALU issue is the bottleneck of the forward path (one slot per co-resident MFMA)
#pragma unroll
                for (int i = 0; i < NVEC; ++i) *(uint4*)(lds_st + i * (64 * PITCH)) = norm_relu16<bf16_t>(pre[i], sc_, nb_);
                return;
            }
#pragma unroll
            for (int i = 0; i < NVEC; ++i) {
                uint4 q = pre[i];
                if (norm) {                                           // uniform branch around pure VALU work
                    const uint4 nq = norm_relu16<bf16_t>(q, sc_, nb_);
                    const uint32_t m = (uint32_t)((int32_t)(vmask_pre << (31 - i)) >> 31);      // padding stays zero AFTER the activation
                    q = make_uint4(nq.x & m, nq.y & m, nq.z & m, nq.w & m);
                }
                *(uint4*)(lds_st + i * (64 * PITCH)) = q;             // rows 648..703 of the padded buffer take the 11th vector's spill-over
            }
        };
        issue(0, preA, vmA, fastA); commit(0, preA, vmA, fastA);
        issue(1, preB, vmB, fastB);
        issue(2, preA, vmA, fastA);
        __syncthreads();                                              // item 0 visible
        int it = 0;
#ifdef RS_PC2_PROF
        unsigned long long pf[3] = {0, 0, 0};
#endif
        for (; it + 1 < nitems; it += 2) {
            PC2_T(q0)
#ifdef RS_PC2_PROF
            __builtin_amdgcn_s_waitcnt(0xF7B);                        // vmcnt(11): the item about to be written has landed
            { const unsigned long long qw = __builtin_readcyclecounter(); if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) g_pc2_prof[wave * 8 + 4] += qw - q0; }
#endif
            commit(it + 1, preB, vmB, fastB);
            PC2_T(q1)
            issue(it + 3, preB, vmB, fastB);
            PC2_T(q2)
            __syncthreads();
            PC2_T(q3)
            commit(it + 2, preA, vmA, fastA);
            PC2_T(q4)
            issue(it + 4, preA, vmA, fastA);
            PC2_T(q5)
            __syncthreads();
            PC2_T(q6)
            PC2_ACC(pf[0], q0, q1) PC2_ACC(pf[1], q1, q2) PC2_ACC(pf[2], q2, q3) PC2_ACC(pf[0], q3, q4) PC2_ACC(pf[1], q4, q5) PC2_ACC(pf[2], q5, q6)
        }
        if (it < nitems) {                                            // odd item count: one more barrier to match the consumers
            commit(it + 1, preB, vmB, fastB);
            __syncthreads();
        }
#ifdef RS_PC2_PROF
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) { for (int i = 0; i < 3; ++i) g_pc2_prof[wave * 8 + i] = pf[i]; g_pc2_prof[wave * 8 + 3] = nitems; }
#endif
    } else {
        // ------------------------------------------------------------------ consumer waves
        const int wm = NT == 1 ? wave : wave >> 1, wnt = NT == 1 ? 0 : wave & 1;
        const int d0w = wm * ND;
        const int ntile = blockIdx.y * NT + wnt;                      // this wave's 32-column tile
        const int hi = lane >> 5;
        int hs_l, wl_l;
        row_to_hw(lane & 31, hs_l, wl_l);
        const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);

        // weights: 9 (kd, kh) taps of one (kw, k-step) group in registers, the next group's 9 loading into the other set
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7FFFFFFF, 0x00020000);
        const uint32_t wstep16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((size_t)p.ntiles * 64 * 16));   // bytes per (chunk, tap, k-step)
        const uint32_t wn_off = (uint32_t)__builtin_amdgcn_readfirstlane(ntile * 1024);
        const uint32_t lane16 = (uint32_t)lane * 16u;
        uint4 wq[2][9];
        auto load_wgroup = [&](uint32_t chbase, int g, uint4* dst) {  // group g = kw * 2 + ks
            const int kw = g >> 1, ks = g & 1;
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, chbase + (uint32_t)((t9 * 3 + kw) * 2 + ks) * wstep16, 0);
                dst[t9] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        };

        // activation fragments
        const int a_base0 = ((d0w * HH + hs_l * 2) * HW + wl_l) * PITCH + hi * 16;
        constexpr int GF = NDR * 4;                                   // fragments per (kw, ks) group
        constexpr int NFRAG = 6 * GF;
        auto frag_off = [](int f) {
            const int g = f / GF, r2 = f % GF, kw = g >> 1, ks = g & 1;
            const int dr = r2 / 4, s = s_order(r2 % 4);
            return ((dr * HH + s) * HW + kw) * PITCH + ks * 32;
        };
        constexpr int ADIST = 3, AR = 4;
        uint4 aq[AR];

        f32x16_t acc[ND][2];
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        float s1[2][KP], s2[2][KP];                                   // running statistics of this lane's 16 output channels
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < KP; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }

        const uint32_t col_lo = (uint32_t)(ntile * 32 + hi * 8);      // first of the lane's two 8-channel groups (second: + 16)
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, nvox_total * (uint32_t)p.ldo * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? p.res : p.out), 0,
                                                                             nvox_total * (uint32_t)(EPI == 2 ? p.ldr : p.ldo) * 2u, 0x00020000);
        uint4 ev[ND][2][2];                                           // prefetched epilogue operands (residual / forward input)

        load_wgroup(wn_off, 0, wq[0]);
        __syncthreads();                                              // item 0 visible
#ifdef RS_PC2_PROF
        unsigned long long cf[3] = {0, 0, 0};
#endif
        for (int it = 0; it < nitems; ++it) {
            PC2_T(c0)
            const int ch = it % nch;
            const char* cur = bufs + (it & 1) * HB;
            const uint32_t cb_cur = wn_off + (uint32_t)__builtin_amdgcn_readfirstlane(ch * 54) * wstep16;
            const uint32_t cb_nxt = wn_off + (uint32_t)__builtin_amdgcn_readfirstlane((ch + 1 == nch ? 0 : ch + 1) * 54) * wstep16;
            const bool last_chunk = ch == nch - 1;
            int d0 = 0, h0 = 0, w0 = 0;
            if (last_chunk) tile_origin(it / nch, d0, h0, w0);
            auto load_ev = [&]() {                                    // epilogue operands: requested mid-item, consumed after the MFMA loop
#pragma unroll
                for (int dl = 0; dl < ND; ++dl)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const int d = d0 + d0w + dl, h = h0 + pp + 2 * hs_l, w = w0 + wl_l;
                        const bool vok = d < p.D && h < p.H && w < p.W;
                        const uint32_t vox = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const uint32_t col = col_lo + 16u * g;
                            const bool ok = vok && col < (uint32_t)p.Cout;
                            if (EPI == 2) {
                                const auto q = __builtin_amdgcn_raw_buffer_load_b128(rrs, ok ? (vox * (uint32_t)p.ldr + col) * 2u : 0xFFFFFFFFu, 0, 0);
                                ev[dl][pp][g] = make_uint4(q[0], q[1], q[2], q[3]);
                            } else {
                                const bool useb = col >= (uint32_t)p.ea.C;
                                const bf16_t* ptr = ok ? (useb ? (const bf16_t*)p.eb.x + (size_t)(vox * (uint32_t)p.eb.ld + col - (uint32_t)p.ea.C)
                                                               : (const bf16_t*)p.ea.x + (size_t)(vox * (uint32_t)p.ea.ld + col))
                                                       : (const bf16_t*)p.ea.x;
                                ev[dl][pp][g] = *(const uint4*)ptr;
                            }
                        }
                    }
            };

#pragma unroll
            for (int f = 0; f < ADIST; ++f) aq[f] = *(const uint4*)(cur + a_base0 + frag_off(f));
#pragma unroll
            for (int f = 0; f < NFRAG; ++f) {
                const int g = f / GF, r2 = f % GF;
                const int dr = r2 / 4, s = s_order(r2 % 4);
                if (f + ADIST < NFRAG) aq[(f + ADIST) % AR] = *(const uint4*)(cur + a_base0 + frag_off(f + ADIST));
                // next group's weights: into the other register set at the start of every group (the last group of an item
                // fetches group 0 of the next item's chunk)
                if (r2 == 0) { if (g < 5) load_wgroup(cb_cur, g + 1, wq[(g + 1) & 1]); else load_wgroup(cb_nxt, 0, wq[0]); }
                if (EPI != 0 && f == 3 * GF && last_chunk) load_ev();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dl = 0; dl < ND; ++dl) {
                    const int kd = dr - dl;
                    if (kd < 0 || kd > 2) continue;
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const int kh = s - pp;
                        if (kh < 0 || kh > 2) continue;
                        mma_wa(acc[dl][pp], wq[g & 1][kd * 3 + kh], aq[f % AR]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }

            PC2_T(c1)
            if (last_chunk) {
                // ------------------------------------------------------------ epilogue straight from the accumulators
#pragma unroll
                for (int dl = 0; dl < ND; ++dl)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        float a[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) a[r] = acc[dl][pp][r];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { swap32(a[j], a[4 + j]); swap32(a[8 + j], a[12 + j]); }
                        const int d = d0 + d0w + dl, h = h0 + pp + 2 * hs_l, w = w0 + wl_l;
                        const bool vok = d < p.D && h < p.H && w < p.W;
                        const uint32_t vox = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const uint32_t col = col_lo + 16u * g;
                            const bool ok = vok && col < (uint32_t)p.Cout;
                            float v[KP];
#pragma unroll
                            for (int q = 0; q < KP; ++q) v[q] = a[g * 8 + q];
                            if (EPI == 2) {
                                float rr[KP];
                                unpack16<bf16_t>(ev[dl][pp][g], rr);
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
                                for (int q = 0; q < KP; ++q) { s1[g][q] += r8[q]; s2[g][q] += r8[q] * r8[q]; }
                            } else {
                                float xx[KP], xn[KP];
                                unpack16<bf16_t>(ev[dl][pp][g], xx);
                                const float4* e4 = (const float4*)(emr_lds + 2 * (wnt * 32 + g * 16 + hi * 8));
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
                                for (int q = 0; q < KP; ++q) { s1[g][q] += r8[q]; s2[g][q] += r8[q] * xn[q]; }
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pk), ors,
                                                                   ok ? (vox * (uint32_t)p.ldo + col) * 2u : 0xFFFFFFFFu, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[dl][pp][r] = 0.f;
                    }
            }
            PC2_T(c2)
            __syncthreads();
#ifdef RS_PC2_PROF
            cf[0] += c1 - c0; cf[1] += c2 - c1; cf[2] += __builtin_readcyclecounter() - c2;
#endif
        }
#ifdef RS_PC2_PROF
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) { for (int i = 0; i < 3; ++i) g_pc2_prof[wave * 8 + i] = cf[i]; g_pc2_prof[wave * 8 + 3] = nitems; }
#endif

        // statistics: one partial row per (block, consumer M part)
        if (p.part) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int q = 0; q < KP; ++q) {
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) { s1[g][q] += __shfl_xor(s1[g][q], o, 64); s2[g][q] += __shfl_xor(s2[g][q], o, 64); }
                }
            constexpr int RPB = NT == 1 ? 4 : 2;                      // partial rows per block
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t col = col_lo + 16u * g;
                const bool ok = (lane & 31) == 0 && col < (uint32_t)p.Cout;
                const uint32_t poff = ok ? (uint32_t)(((((size_t)n * gridDim.x + blockIdx.x) * RPB + wm) * p.Cout + col) * 8) : 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < KP; q += 2) {
                    u32x4_t pv;
                    pv[0] = __float_as_uint(s1[g][q]); pv[1] = __float_as_uint(s2[g][q]); pv[2] = __float_as_uint(s1[g][q + 1]); pv[3] = __float_as_uint(s2[g][q + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(pv, prs, ok ? poff + q * 8 : poff, 0, 0);
                }
            }
        }
    }
}

int pc2_grid_x(int tiles, int gy, int N) {                            // ~one persistent block per CU
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}

template <int NT>
int launch_pc2(const IgemmParams& p, int epi, hipStream_t st) {
    const int tiles = ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const int gy = p.ntiles / NT;
    dim3 grid(pc2_grid_x(tiles, gy, p.N), gy, p.N), block(512);
    const size_t smem = 2 * (size_t)HB + (size_t)(p.a.C + p.b.C) * 8 + 64 * NT * 4;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
#define RS_PC2_LAUNCH(E)                                                                                     \
    {                                                                                                        \
        auto k = igemm_pc2_kernel<NT, E>;                                                                    \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);    \
        hipLaunchKernelGGL(k, grid, block, smem, st, p);                                                     \
        return rs_check_launch();                                                                            \
    }
#ifdef RS_PC2_PROF
    struct Guard { ~Guard() {
        if (!getenv("RSUPER_PC2_PROF")) return;
        unsigned long long h[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pc2_prof), sizeof(h));
        for (int w = 0; w < 8; w += 1) fprintf(stderr, "pc2_prof wave %d (%s): items %llu | per item: %s %.0f %s %.0f barrier %.0f | data wait per even item %.0f\n", w, w < 4 ? "consumer" : "producer", h[w * 8 + 3],
                                               w < 4 ? "mfma-loop" : "commit", (double)h[w * 8] / h[w * 8 + 3], w < 4 ? "epilogue" : "issue", (double)h[w * 8 + 1] / h[w * 8 + 3], (double)h[w * 8 + 2] / h[w * 8 + 3],
                                               2.0 * h[w * 8 + 4] / h[w * 8 + 3]);
        static const unsigned long long z[64] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pc2_prof), z, sizeof(z));
    } } guard;
#endif
    if (epi == 0 && p.res) RS_PC2_LAUNCH(2)
    if (epi == 0) RS_PC2_LAUNCH(0)
    RS_PC2_LAUNCH(1)
#undef RS_PC2_LAUNCH
}

}  // namespace

// bf16, bn 32 / 64; writes the same partial rows as igemm_pc_kernel (one per (block, consumer M part)).
int rs_launch_igemm_pc2(const IgemmParams& p, int epi, hipStream_t st) {
    if (p.bn == 32) return launch_pc2<1>(p, epi, st);
    if (p.bn == 64 && p.ntiles % 2 == 0) return launch_pc2<2>(p, epi, st);
    return RS_ERR_ARG;
}
