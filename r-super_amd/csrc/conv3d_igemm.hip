// 3x3x3 / stride 1 / pad 1 convolution as an implicit GEMM on MFMA (gfx950), channels-last (NDHWC).
//
//   M = voxels (block tile 4x4x16 = 256), N = output channels (32/64/128 per block), K = 27 taps x Cin.
//
// Replaces, on the hot path, every nn.Conv3d(k=3, bias=False) forward and its autograd data-gradient
// launched by rsuper_train/model/dim3/conv_layers.py:29-38,46-51 (ConvNormAct, preact) inside BasicBlock
// (conv_layers.py:86-94).  Fused into this kernel:
//   * prologue: x_hat = relu((x - mean) * rstd) applied while the haloed input tile is staged into LDS
//     (InstanceNorm3d eps=1e-4 + ReLU of ConvNormAct, conv_layers.py:40-43) -- x_hat is never materialised;
//     zero padding is applied AFTER the activation, as the reference's conv does;
//   * two input sources (channel concat of up_block, unet_utils.py:71, or the [dY1 | dOut] pair of the
//     fused conv1+shortcut data-gradient) without a concat copy;
//   * epilogue FWD  : + residual (BasicBlock `out += shortcut`, conv_layers.py:92), per-block partial
//                     sum / sum-of-squares of the stored output for the NEXT InstanceNorm;
//   * epilogue DGRAD: g = acc * [x_hat > 0] (ReLU backward) and partial sums of g and g*x_n for the
//                     InstanceNorm backward reductions.
//
// Data flow per K chunk (64 bytes of channels: 32 bf16 / 16 f32):
//   global --(16-B loads, norm+relu in registers)--> LDS halo tile [6][6][18] rows x 80-B pitch
//   LDS --ds_read_b128--> A fragments;  packed weights (fragment order, L1/L2 resident) --> B fragments
//   v_mfma_f32_32x32x16_bf16 (or 4x v_mfma_f32_32x32x2_f32 in the f32 parity mode), f32 accumulate.
#include <cstring>
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int TD = 4, TH = 4, TW = 16;          // output tile (voxels)
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HROWS = HD * HH * HW;             // 648 halo rows
constexpr int PITCH = 80;                       // bytes per LDS row: 64 data + 16 pad (conflict-free b128)
constexpr int HALO_BYTES = HROWS * PITCH;       // 51840

// Staging is split in two (issue-early / write-late): `stage_issue` starts the 16-byte global loads of the NEXT
// K chunk into registers right before the MFMA phase of the current one, `stage_commit` applies norm+ReLU and
// writes LDS after the barrier -- HBM/L2 latency hides under the MFMAs.
constexpr int NVEC = (HROWS * 4 + 255) / 256;   // 16-byte vectors per thread per chunk (11)

// EPI: 0 = forward (residual + stats of output), 1 = dgrad (relu mask + IN-backward sums)
// KSPLIT: 1 = waves tile M x N;  2 = 32-column layers: waves = 2 (M halves) x 2 (tap halves), so weight fragments are
//         only 2x (not 4x) redundant in L1; the two tap-partial accumulators of each M half are exchanged through
//         LDS (8 KB per wave) before the epilogue.
template <typename T, int WM, int MF, int WN, int NF, int EPI, int KSPLIT>
__global__ __launch_bounds__(256, (MF * NF <= 4 && sizeof(T) == 2) ? 3 : 2) void igemm_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    float* mr_lds = (float*)(smem + HALO_BYTES);      // [Ca + Cb][2]
    constexpr int KC = Elem<T>::KC;
    constexpr int KP = Elem<T>::KP;
    constexpr int BN32 = WN * NF;                     // 32-column tiles per block
    static_assert(KSPLIT == 1 || (KSPLIT == 2 && WM == 2 && WN == 1 && MF == 4 && NF == 1), "tap-split config is 2 x (128 x 32) x 2 tap halves");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = KSPLIT == 1 ? wave / WN : (wave & 1), wn = KSPLIT == 1 ? wave % WN : 0;
    const int kt = KSPLIT == 1 ? 0 : (wave >> 1);     // tap half of this wave
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH;
    // XCD-aware tile order: the dispatcher places linear workgroup id b on XCD b % 8 (8 private L2s).  Give every XCD
    // a contiguous run of spatial tiles so neighbouring tiles (which share halo rows) hit the same L2.  Bijective for any
    // tile count; a different placement only changes speed.
    int t;
    {
        const int nt = gridDim.x, b = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = b & 7, k = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tile_id = t;
    int tw, th, td;
    rs_tile_coords(t, tiles_w, tiles_h, (p.D + TD - 1) / TD, tw, th, td);
    const int n = blockIdx.z;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = nchA + nchB;
    const bool normA = p.a.mr != nullptr, normB = p.b.mr != nullptr;

    // per-sample mean/rstd of every input channel -> LDS
    if (normA) for (int i = tid; i < 2 * p.a.C; i += 256) mr_lds[i] = p.a.mr[(size_t)n * 2 * p.a.C + i];
    if (normB) for (int i = tid; i < 2 * p.b.C; i += 256) mr_lds[2 * p.a.C + i] = p.b.mr[(size_t)n * 2 * p.b.C + i];

    // per-lane LDS byte offset of the A-fragment row for each m-fragment (tap (0,0,0), k-step 0)
    // fragment f = wm*MF + mf (0..7 in the block) covers d = f/2, h pair = f%2; MF is even, so the per-fragment
    // part ((mf/2)*HH + (mf%2)*2)*HW*PITCH is a compile-time constant folded into the ds_read offset.
    int hs, wl;
    row_to_hw(lane & 31, hs, wl);
    const int a_base = (((wm * MF / 2) * HH + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
    auto a_const = [](int mf) { return (((mf >> 1) * HH + (mf & 1) * 2) * HW) * PITCH; };

    f32x16_t acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

    const uint4* wp = (const uint4*)p.wp;
    const int ntile0 = blockIdx.y * BN32 + wn * NF;
    const size_t wstep = (size_t)p.ntiles * 64;       // uint4 per (chunk, tap, kstep)

    // Staging geometry is per-thread constant: vector i of this thread is halo row (tid>>2) + 64*i, 16-byte slot
    // tid&3 -> its voxel index (or -1 when outside the volume / tile), its LDS address and its 8 (bf16) / 4 (f32)
    // channels never change between K chunks, so no coordinate math is left in the chunk loop.
    const int slot = tid & 3;
    int vi[NVEC];
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const int hd = r / (HH * HW);
        const int rem = r - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
        const bool ok = r < HROWS && d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W;
        vi[i] = ok ? ((n * p.D + d) * p.H + h) * p.W + w : -1;
    }
    char* lds_st = halo + (tid >> 2) * PITCH + slot * 16;

    uint4 pre[NVEC];
    // Halo loads go through a buffer descriptor: voxels outside the volume (and channel slots past C) get an offset
    // beyond num_records, for which the hardware returns zeros -- no exec-mask branches around the 11 loads.
    const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
    auto issue = [&](int ch) {
        const bool isB = ch >= nchA;
        const ConvSrc& src = isB ? p.b : p.a;
        const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
        const uint32_t rowb = (uint32_t)src.ld * (uint32_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src.x, 0, nvox_total * rowb, 0x00020000);
        const uint32_t cb = c < src.C ? (uint32_t)c * (uint32_t)sizeof(T) : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const uint32_t off = (vi[i] >= 0 && cb != 0xFFFFFFFFu) ? (uint32_t)vi[i] * rowb + cb : 0xFFFFFFFFu;
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            pre[i] = make_uint4(q[0], q[1], q[2], q[3]);
        }
    };
    auto commit = [&](int ch) {
        const bool isB = ch >= nchA;
        const ConvSrc& src = isB ? p.b : p.a;
        const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
        const bool norm = (isB ? normB : normA) && c < src.C;
        float sc_[KP], nb_[KP];                        // x_hat = max(x * rstd - mean * rstd, 0)
        if (norm) {
            const float* mr = mr_lds + 2 * ((isB ? p.a.C : 0) + c);
#pragma unroll
            for (int j = 0; j < KP; ++j) { sc_[j] = mr[2 * j + 1]; nb_[j] = -mr[2 * j] * mr[2 * j + 1]; }
        }
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            uint4 q = pre[i];
            if (norm && vi[i] >= 0) {
                q = norm_relu16<T>(q, sc_, nb_);
            }
            if ((tid >> 2) + 64 * i < HROWS) *(uint4*)(lds_st + i * (64 * PITCH)) = q;
        }
    };

    // configs whose accumulators already take 128 VGPRs stage synchronously (the co-resident block covers the
    // latency); the others keep the next chunk's loads in flight during the MFMA phase.
    constexpr bool PF = MF * NF <= 4;
    if (PF) issue(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                              // previous chunk fully consumed (and mr_lds visible)
        if (!PF) issue(ch);
        commit(ch);
        __syncthreads();
        if (PF && ch + 1 < nch) issue(ch + 1);        // loads fly during the MFMA phase below

        // ---- MFMA phase, software pipelined with static register indices:
        //   unit u = (step, group of AU m-fragments); A fragments of unit u+1 are read from LDS while unit u's MFMAs
        //   issue; B fragments (weights, L1/L2 resident) run RB steps ahead in a register ring.
        const uint4* wch = wp + (size_t)ch * 27 * 2 * wstep + (size_t)ntile0 * 64 + lane;
        constexpr int AU = 2;                                     // A fragments per pipeline unit (register budget)
        constexpr int G = MF / AU;                                // units per step
        constexpr int NSTEP = KSPLIT == 1 ? 54 : 28;              // (tap, k-step) pairs handled by this wave
#ifndef RS_CL_RB
#define RS_CL_RB 2
#endif
        constexpr int RB = RS_CL_RB;                              // B ring depth (steps); swept 2 / 3 / 4 (round 3)
        const int wv = KSPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane(kt);
        auto step_tap = [&](int st) { return KSPLIT == 1 ? st >> 1 : wv + KSPLIT * (st >> 1); };
        auto tap_off = [&](int tap) {
            const int tc = tap < 27 ? tap : 26;
            const int kd = tc / 9, kh = (tc - kd * 9) / 3, kw = tc - kd * 9 - kh * 3;
            return ((kd * HH + kh) * HW + kw) * PITCH;
        };
        auto load_b = [&](int st, uint4* dst) {
            const int tap = step_tap(st);
            const bool ok = tap < 27;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                dst[nf] = ok ? wch[(size_t)(tap * 2 + (st & 1)) * wstep + nf * 64] : make_uint4(0, 0, 0, 0);
        };
        auto load_a = [&](int u, uint4* dst) {
            const int st = u / G, g = u % G;
            const int off = tap_off(step_tap(st)) + (st & 1) * 32;
#pragma unroll
            for (int i = 0; i < AU; ++i) dst[i] = *(const uint4*)(halo + a_base + a_const(g * AU + i) + off);
        };
        uint4 bq[RB][NF];
        uint4 aq[2][AU];
#pragma unroll
        for (int r = 0; r < RB; ++r) load_b(r, bq[r]);
        load_a(0, aq[0]);
#pragma unroll
        for (int u = 0; u < NSTEP * G; ++u) {
            const int st = u / G, g = u % G;
            if (u + 1 < NSTEP * G) load_a(u + 1, aq[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of this unit's MFMAs
#pragma unroll
            for (int i = 0; i < AU; ++i)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) mma32<T>(acc[g * AU + i][nf], aq[u & 1][i], bq[st % RB][nf]);
            if (g == G - 1 && st + RB < NSTEP) load_b(st + RB, bq[st % RB]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ------------------------------------------------------------------ epilogue
    __syncthreads();                                  // halo region is reused as reduction scratch
    constexpr int MFE = MF / KSPLIT;                  // m-fragments per wave in the epilogue
    constexpr int WME = WM * KSPLIT;
    const int wme = KSPLIT == 1 ? wm : wm * 2 + kt;
    f32x16_t eacc[MFE][NF];
    if constexpr (KSPLIT == 1) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) eacc[mf][nf] = acc[mf][nf];
    } else {
        // wave (wm, kt) owns fragments [2*kt, 2*kt+2) of its M half: it publishes the other two fragments' partial
        // sums (8 KB) and adds the partner's (same wm, other tap half = wave ^ 2) partials for the ones it owns.
        float* sc = (float*)smem;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sc[wave * 2048 + (mf * 16 + r) * 64 + lane] = kt ? acc[mf][0][r] : acc[2 + mf][0][r];
        __syncthreads();
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                eacc[mf][0][r] = (kt ? acc[2 + mf][0][r] : acc[mf][0][r]) + sc[(wave ^ 2) * 2048 + (mf * 16 + r) * 64 + lane];
        __syncthreads();
    }
    // ---- cooperative epilogue through LDS: accumulators are written voxel-major as f32 ([voxel][BN] + 16 B pad), then
    //      all 256 threads walk the tile with 16-byte vectors (coalesced residual / forward-input loads and output
    //      stores) and keep per-column partial sums for the InstanceNorm statistics.  FP fragments (32 voxels each)
    //      per pass keep the scratch below the halo buffer size.
    constexpr int BN = BN32 * 32;
    constexpr int FP = BN == 128 ? 2 : BN == 64 ? 4 : 8;
    constexpr int NPASS = 8 / FP;
    constexpr int EPF = BN + 4;                       // scratch row pitch in floats
    constexpr int CG = BN / KP;                       // 16-byte column groups per voxel
    constexpr int RPT = 256 / CG;                     // voxel stride between a thread's vectors
    constexpr int NV = FP * 32 / RPT;                 // vectors per thread per pass
    static_assert(FP * 32 * EPF * 4 <= HALO_BYTES, "epilogue scratch must fit in the halo buffer");
    float* sc2 = (float*)smem;
    const int col_l = lane & 31, hi = lane >> 5;
    const int cg = tid % CG, pr0 = tid / CG;
    const int col0 = blockIdx.y * BN + cg * KP;       // first output column of this thread's vectors
    const bool cok = col0 < p.Cout;
    const bool inb = d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;
    // dgrad: forward-input source (and its statistics) of this thread's columns
    const ConvSrc& es = (EPI == 1 && col0 >= p.ea.C) ? p.eb : p.ea;
    const int ecol0 = (EPI == 1 && col0 >= p.ea.C) ? col0 - p.ea.C : col0;
    float emu[KP], ers[KP];
    if (EPI == 1 && cok) {
#pragma unroll
        for (int j = 0; j < KP; ++j) { emu[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2]; ers[j] = es.mr[((size_t)n * es.C + ecol0 + j) * 2 + 1]; }
    }
    float s1[KP], s2[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
#pragma unroll
        for (int mf = 0; mf < MFE; ++mf) {
            const int f = wme * MFE + mf;             // block fragment: d = f/2, h pair = f%2
            if (f / FP == q) {
                const int rowbase = (((f - q * FP) >> 1) * TH + (f & 1) * 2) * TW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hw0 = row_hw_packed((r & 3) + 8 * (r >> 2)), hw1 = row_hw_packed((r & 3) + 8 * (r >> 2) + 4);
                    const int hw = hi ? hw1 : hw0;
                    float* dst = sc2 + (rowbase + (hw >> 4) * TW + (hw & 15)) * EPF + wn * NF * 32 + col_l;
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) dst[nf * 32] = eacc[mf][nf][r];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pr = pr0 + j * RPT;             // voxel of this pass: (pr/64, (pr/16)%4, pr%16)
            const int d = d0 + q * (FP / 2) + (pr >> 6), h = h0 + ((pr >> 4) & 3), w = w0 + (pr & 15);
            if (cok && (inb || (d < p.D && h < p.H && w < p.W))) {
                float v[KP];
                const float4* sp = (const float4*)(sc2 + pr * EPF + cg * KP);
#pragma unroll
                for (int k4 = 0; k4 < KP / 4; ++k4) { const float4 t4 = sp[k4]; v[k4 * 4] = t4.x; v[k4 * 4 + 1] = t4.y; v[k4 * 4 + 2] = t4.z; v[k4 * 4 + 3] = t4.w; }
                const uint32_t vox = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
                if (EPI == 0) {
                    if (p.res) {
                        float rr[KP];
                        unpack16<T>(*(const uint4*)((const T*)p.res + (size_t)(vox * (uint32_t)p.ldr + (uint32_t)col0)), rr);
#pragma unroll
                        for (int k = 0; k < KP; ++k) v[k] += rr[k];
                    }
#pragma unroll
                    for (int k = 0; k < KP; ++k) { v[k] = Elem<T>::rnd(v[k]); s1[k] += v[k]; s2[k] += v[k] * v[k]; }
                } else {
                    float xx[KP];
                    unpack16<T>(*(const uint4*)((const T*)es.x + (size_t)(vox * (uint32_t)es.ld + (uint32_t)ecol0)), xx);
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const float xn = (xx[k] - emu[k]) * ers[k];
                        v[k] = Elem<T>::rnd(xn > 0.f ? v[k] : 0.f);
                        s1[k] += v[k]; s2[k] += v[k] * xn;
                    }
                }
                *(uint4*)((T*)p.out + (size_t)(vox * (uint32_t)p.ldo + (uint32_t)col0)) = pack16<T>(v);
            }
        }
        __syncthreads();
    }
    if (p.part) {
        float* red = (float*)smem;                    // [RPT][BN][2]
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            red[(pr0 * BN + cg * KP + k) * 2] = s1[k];
            red[(pr0 * BN + cg * KP + k) * 2 + 1] = s2[k];
        }
        __syncthreads();
        for (int cl = tid; cl < BN; cl += 256) {
            float a = 0.f, b = 0.f;
            for (int m = 0; m < RPT; ++m) { a += red[(m * BN + cl) * 2]; b += red[(m * BN + cl) * 2 + 1]; }
            const int col = blockIdx.y * BN + cl;
            if (col < p.Cout) {
                float* pp = p.part + (((size_t)n * gridDim.x + tile_id) * p.Cout + col) * 2;
                pp[0] = a; pp[1] = b;
            }
        }
    }
}

// =====================================================================================================================
// Producer / consumer (wave-specialised) persistent variant.
//   512 threads: waves 0-3 = consumers (MFMA + wave-private epilogue), waves 4-7 = producers (global loads, norm+ReLU,
//   LDS writes).  Each SIMD hosts one consumer and one producer wave, so the producers' VALU/VMEM/LDS-write work overlaps
//   the consumers' MFMAs every cycle instead of alternating with them behind barriers.  Two halo buffers; ONE block
//   barrier per item (item = K chunk of a tile); blocks are persistent over spatial tiles of one sample so the producer
//   runs ahead across tile boundaries (its loads for item i+2 are in flight while it commits item i+1).
//   The epilogue needs no cross-wave exchange: every consumer wave transposes its own fragments through a private LDS
//   scratch and writes per-(tile, wm) partial statistics.
// =====================================================================================================================
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int MF, int NF> struct PcCfg {
    static constexpr int SCR_ROW = 32 * NF + 4;                       // floats per scratch row
    static constexpr int SCR_BYTES = 32 * SCR_ROW * 4;                // per consumer wave (one fragment row-block)
};

template <typename T, int WM, int MF, int WN, int NF, int EPI>
__global__ __launch_bounds__(512, 1) void igemm_pc_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = Elem<T>::KC;
    constexpr int KP = Elem<T>::KP;
    constexpr int BN32 = WN * NF;
    static_assert(WM * WN == 4 && WM * MF == 8, "4 consumer waves cover 256 voxels x BN columns");
    char* bufs = smem;                                                // 2 x HALO_BYTES
    float* mr_lds = (float*)(smem + 2 * HALO_BYTES);                  // [Ca + Cb][2]
    char* scr_base = smem + 2 * HALO_BYTES + ((p.a.C + p.b.C) * 8 + 15) / 16 * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    // Several column groups (gridDim.y > 1: the 192-column data gradient of up3.0 on 64-column blocks): the groups of one (tile set, sample) read the same
    // halo rows, but their linear workgroup ids differ by gridDim.x -- 42 there -- and land on different XCDs (private L2 each): every group pulled the
    // rows over the fabric again (617 MB fetched for 141 MB of operands).  The blocks are renumbered so that consecutive ids share an XCD (as wg_block_map
    // does for the weight gradient) and decoded column-group-fastest: the groups of a tile set run on ONE XCD.  Placement only.
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bn_ = (int)blockIdx.z;
    if (gridDim.y > 1) {
        const unsigned gx = gridDim.x, gy = gridDim.y, NB = gx * gy * gridDim.z, L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned q = NB >> 3, r = NB & 7, xc = L & 7;
        const unsigned V = xc * q + (xc < r ? xc : r) + (L >> 3);
        by = (int)(V % gy); bx = (int)((V / gy) % gx); bn_ = (int)(V / (gy * gx));
    }
    const int n = bn_;
    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d;
    const int nchA = (p.a.C + KC - 1) / KC, nchB = (p.b.C + KC - 1) / KC;
    const int nch = nchA + nchB;
    const bool normA = p.a.mr != nullptr, normB = p.b.mr != nullptr;
    const int my_tiles = (bx < tiles) ? (tiles - 1 - bx) / (int)gridDim.x + 1 : 0;
    const int nitems = my_tiles * nch;

    if (normA) for (int i = tid; i < 2 * p.a.C; i += 512) mr_lds[i] = p.a.mr[(size_t)n * 2 * p.a.C + i];
    if (normB) for (int i = tid; i < 2 * p.b.C; i += 512) mr_lds[2 * p.a.C + i] = p.b.mr[(size_t)n * 2 * p.b.C + i];
    __syncthreads();

    // XCD-aware tile order (same bijection as the classic kernel): the logical index L = blockIdx.x + k * gridDim.x of the
    // blocks running concurrently covers a contiguous range; linear workgroup id b lands on XCD b % 8, so L -> tile gives
    // every XCD (private L2) a contiguous run of tiles and the halo rows shared by neighbouring tiles are fetched once
    const bool xcd_remap = (gridDim.x & 7) == 0 && tiles >= 64;
    auto tile_origin = [&](int k, int& d0, int& h0, int& w0) {
        int t = bx + k * (int)gridDim.x;
        if (xcd_remap) {
            const int q = tiles >> 3, r = tiles & 7, xcd = t & 7, kk = t >> 3;
            t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
        }
        int tw, th, td;
        rs_tile_coords(t, tiles_w, tiles_h, (p.D + TD - 1) / TD, tw, th, td);
        d0 = td * TD; h0 = th * TH; w0 = tw * TW;
    };

    if (producer) {
        // ------------------------------------------------------------------ producer waves
        const int ptid = tid - 256;
        const int slot = ptid & 3;
        const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
        int vi[NVEC];
        uint4 preA[NVEC], preB[NVEC];                                 // two items in flight (global latency >> one item of MFMA work)
        uint32_t vmA = 0, vmB = 0;                                    // validity of the vectors held in preA / preB
        int vi_tile = -1;
        // per-thread constants: the halo row of vector i never changes, so an interior tile (wave-uniform test) costs one
        // add per vector instead of a div/mod decomposition plus six bounds checks
        int vdelta[NVEC];
        uint32_t rows_ok = 0;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int r = (ptid >> 2) + 64 * i;
            const int hd = r / (HH * HW);
            const int rem = r - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            vdelta[i] = (hd * p.H + hh) * p.W + hw;
            rows_ok |= r < HROWS ? (1u << i) : 0u;
        }
        auto setup_tile = [&](int k) {
            if (k == vi_tile) return;
            vi_tile = k;
            int d0, h0, w0;
            tile_origin(k, d0, h0, w0);
            if (d0 >= 1 && d0 + TD + 1 <= p.D && h0 >= 1 && h0 + TH + 1 <= p.H && w0 >= 1 && w0 + TW + 1 <= p.W) {
                const int base = ((n * p.D + d0 - 1) * p.H + h0 - 1) * p.W + w0 - 1;
#pragma unroll
                for (int i = 0; i < NVEC; ++i) vi[i] = ((rows_ok >> i) & 1u) ? base + vdelta[i] : -1;
                return;
            }
#pragma unroll
            for (int i = 0; i < NVEC; ++i) {
                const int r = (ptid >> 2) + 64 * i;
                const int hd = r / (HH * HW);
                const int rem = r - hd * (HH * HW);
                const int hh = rem / HW, hw = rem - hh * HW;
                const int d = d0 - 1 + hd, h = h0 - 1 + hh, w = w0 - 1 + hw;
                const bool ok = r < HROWS && d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W;
                vi[i] = ok ? ((n * p.D + d) * p.H + h) * p.W + w : -1;
            }
        };
        auto issue = [&](int it, uint4* pre, uint32_t& vmask_pre) {
            setup_tile(it / nch);
            const int ch = it % nch;
            const bool isB = ch >= nchA;
            const ConvSrc& src = isB ? p.b : p.a;
            const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
            const uint32_t rowb = (uint32_t)src.ld * (uint32_t)sizeof(T);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src.x, 0, nvox_total * rowb, 0x00020000);
            const uint32_t cb = c < src.C ? (uint32_t)c * (uint32_t)sizeof(T) : 0xFFFFFFFFu;
            vmask_pre = 0;
#pragma unroll
            for (int i = 0; i < NVEC; ++i) {
                const bool ok = vi[i] >= 0 && cb != 0xFFFFFFFFu;
                const uint32_t off = ok ? (uint32_t)vi[i] * rowb + cb : 0xFFFFFFFFu;
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                pre[i] = make_uint4(q[0], q[1], q[2], q[3]);
                vmask_pre |= ok ? (1u << i) : 0u;
            }
        };
        auto commit = [&](int it, const uint4* pre, const uint32_t vmask_pre) {   // writes item `it` (held in pre) into buffer it & 1
            const int ch = it % nch;
            const bool isB = ch >= nchA;
            const ConvSrc& src = isB ? p.b : p.a;
            const int c = (isB ? ch - nchA : ch) * KC + slot * KP;
            const bool norm = (isB ? normB : normA) && c < src.C;
            float sc_[KP], nb_[KP];
            if (norm) {
                const float* mr = mr_lds + 2 * ((isB ? p.a.C : 0) + c);
#pragma unroll
                for (int j = 0; j < KP; ++j) { sc_[j] = mr[2 * j + 1]; nb_[j] = -mr[2 * j] * mr[2 * j + 1]; }
            }
            char* lds_st = bufs + (it & 1) * HALO_BYTES + (ptid >> 2) * PITCH + slot * 16;
#pragma unroll
            for (int i = 0; i < NVEC; ++i) {
                uint4 q = pre[i];
                if (norm && ((vmask_pre >> i) & 1u)) {
                    q = norm_relu16<T>(q, sc_, nb_);
                }
                if ((ptid >> 2) + 64 * i < HROWS) *(uint4*)(lds_st + i * (64 * PITCH)) = q;
            }
        };
        // items alternate between the register sets: even items in A, odd items in B
        if (nitems > 0) { issue(0, preA, vmA); commit(0, preA, vmA); }
        if (nitems > 1) issue(1, preB, vmB);
        if (nitems > 2) issue(2, preA, vmA);
        __syncthreads();                                              // item 0 visible
        for (int it = 0; it < nitems; it += 2) {
            if (it + 1 < nitems) {
                commit(it + 1, preB, vmB);                            // loads were issued two items ago
                if (it + 3 < nitems) issue(it + 3, preB, vmB);
            }
            __syncthreads();
            if (it + 1 < nitems) {
                if (it + 2 < nitems) {
                    commit(it + 2, preA, vmA);
                    if (it + 4 < nitems) issue(it + 4, preA, vmA);
                }
                __syncthreads();
            }
        }
    } else {
        // ------------------------------------------------------------------ consumer waves
        const int wm = wave / WN, wn = wave % WN;
        int hs, wl;
        row_to_hw(lane & 31, hs, wl);
        const int a_base0 = (((wm * MF / 2) * HH + hs) * HW + wl) * PITCH + (lane >> 5) * 16;
        auto a_const = [](int mf) { return (((mf >> 1) * HH + (mf & 1) * 2) * HW) * PITCH; };
        const uint4* wp = (const uint4*)p.wp;
        const int ntile0 = by * BN32 + wn * NF;
        const size_t wstep = (size_t)p.ntiles * 64;
        float* scr = (float*)(scr_base + wave * PcCfg<MF, NF>::SCR_BYTES);
        constexpr int SROW = PcCfg<MF, NF>::SCR_ROW;
        // epilogue thread mapping inside the wave: 16-byte column groups x rows
        constexpr int CGW = 32 * NF / KP;                             // column groups of this wave's 32*NF columns
        constexpr int RPW = 64 / CGW;                                 // rows covered per pass
        const int cg = lane % CGW, er0 = lane / CGW;
        const int col0 = (ntile0 * 32) + cg * KP;                     // first output column of this lane's vectors
        const bool cok = col0 < p.Cout;
        const uint32_t nvox_total = (uint32_t)(p.N * p.D * p.H * p.W);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, nvox_total * (uint32_t)p.ldo * (uint32_t)sizeof(T), 0x00020000);
        const bool useb = EPI == 1 && col0 >= p.ea.C;                  // select VALUES once (a reference into kernarg would be re-read per use)
        const T* es_x = (const T*)(useb ? p.eb.x : p.ea.x);
        const uint32_t es_ld = (uint32_t)(useb ? p.eb.ld : p.ea.ld);
        const int es_C = useb ? p.eb.C : p.ea.C;
        const float* es_mr = useb ? p.eb.mr : p.ea.mr;
        const int ecol0 = useb ? col0 - p.ea.C : col0;
        float emu[KP], ers[KP];
        if (EPI == 1 && cok) {
#pragma unroll
            for (int j = 0; j < KP; ++j) { emu[j] = es_mr[((size_t)n * es_C + ecol0 + j) * 2]; ers[j] = es_mr[((size_t)n * es_C + ecol0 + j) * 2 + 1]; }
        }

        f32x16_t acc[MF][NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

        // epilogue operand (residual / forward input of the ReLU mask) is prefetched one fragment ahead; the first
        // fragment's vectors are requested before the last chunk's MFMA loop so their latency hides behind it
        constexpr int NPS = 32 / RPW;
        constexpr bool need_ld = EPI != 0;
        constexpr bool PRE = NF == 1;                                 // register budget: the 128-column config loads at fragment start instead
        uint4 ev[PRE ? 2 : 1][NPS];
        auto epi_vox = [&](int d0, int h0, int w0, bool inb, int mf, int ps, uint32_t& vox) {
            const int f = wm * MF + mf;
            int rhs, rw;
            row_to_hw(er0 + ps * RPW, rhs, rw);
            const int d = d0 + (f >> 1), h = h0 + (f & 1) * 2 + rhs, w = w0 + rw;
            vox = (uint32_t)(((n * p.D + d) * p.H + h) * p.W + w);
            return cok && (inb || (d < p.D && h < p.H && w < p.W));
        };
        auto epi_load = [&](int d0, int h0, int w0, bool inb, int mf, uint4* dst) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                uint32_t vox;
                const bool ok = epi_vox(d0, h0, w0, inb, mf, ps, vox);
                // branch-free (address select): divergent branches around VMEM make the compiler fall back to vmcnt(0)
                const T* ptr = EPI == 1 ? (ok ? es_x + (size_t)(vox * es_ld + (uint32_t)ecol0) : (const T*)p.ea.x)
                                        : (const T*)p.res + (ok ? (size_t)(vox * (uint32_t)p.ldr + (uint32_t)col0) : (size_t)0);
                dst[ps] = *(const uint4*)ptr;
            }
        };
        constexpr int AU = 2;
        constexpr int G = MF / AU;
        constexpr int NSTEP = 54;
        constexpr int ADIST = NF == 1 ? 3 : 2, AR = ADIST + 1;        // A-operand ring / prefetch distance in units (one MFMA wave per SIMD:
        constexpr int RB = (NF == 1) ? (G == 1 ? 18 : 9) : 3;  // latency must be covered by distance, not by other waves)
        static_assert(NSTEP % RB == 0, "the B ring continues across items: its size must divide the steps per item");
        auto tap_off = [&](int tap) {
            const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
            return ((kd * HH + kh) * HW + kw) * PITCH;
        };
        // B fragments: buffer loads with a per-lane VGPR offset (lane * 16) and wave-uniform SGPR step offsets -> no VGPR addresses
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7FFFFFFF, 0x00020000);
        const uint32_t wstep16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wstep * 16));
        const uint32_t wn_off = (uint32_t)__builtin_amdgcn_readfirstlane(ntile0 * 1024);
        const uint32_t lane16 = (uint32_t)lane * 16u;
        auto load_b_at = [&](uint32_t chbase, int st, uint4* dst) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + nf * 1024, chbase + (uint32_t)st * wstep16, 0);
                dst[nf] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        };
        uint4 bq[RB][NF];
        uint4 aq[AR][AU];
#pragma unroll
        for (int r = 0; r < RB; ++r) load_b_at(wn_off, r, bq[r]);
        float s1[KP], s2[KP];                                         // running statistics over all tiles of this block
#pragma unroll
        for (int j = 0; j < KP; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        __syncthreads();                                              // item 0 visible
        for (int it = 0; it < nitems; ++it) {
            const int ch = it % nch;
            const char* halo = bufs + (it & 1) * HALO_BYTES;
            const int a_base = a_base0;
            const uint32_t cb_cur = wn_off + (uint32_t)__builtin_amdgcn_readfirstlane(ch * 54) * wstep16;
            const uint32_t cb_nxt = wn_off + (uint32_t)__builtin_amdgcn_readfirstlane((ch + 1 == nch ? 0 : ch + 1) * 54) * wstep16;
            auto load_b = [&](int st, uint4* dst) {                   // the B stream continues into the next item before the barrier
                if (st < NSTEP) load_b_at(cb_cur, st, dst); else load_b_at(cb_nxt, st - NSTEP, dst);
            };
            auto load_a = [&](int u, uint4* dst) {
                const int st = u / G, g = u % G;
                const int off = tap_off(st >> 1) + (st & 1) * 32;
#pragma unroll
                for (int i = 0; i < AU; ++i) dst[i] = *(const uint4*)(halo + a_base + a_const(g * AU + i) + off);
            };
            int d0 = 0, h0 = 0, w0 = 0;
            bool inb = false;
            if (ch == nch - 1) {
                tile_origin(it / nch, d0, h0, w0);
                inb = d0 + TD <= p.D && h0 + TH <= p.H && w0 + TW <= p.W;
                if (need_ld && PRE) epi_load(d0, h0, w0, inb, 0, ev[0]);
            }
#pragma unroll
            for (int u = 0; u < ADIST; ++u) load_a(u, aq[u]);
#pragma unroll
            for (int u = 0; u < NSTEP * G; ++u) {
                const int st = u / G, g = u % G;
                if (u + ADIST < NSTEP * G) load_a(u + ADIST, aq[(u + ADIST) % AR]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < AU; ++i)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) mma32<T>(acc[g * AU + i][nf], aq[u % AR][i], bq[st % RB][nf]);
                if (g == G - 1) load_b(st + RB, bq[st % RB]);
                __builtin_amdgcn_sched_barrier(0);
            }

            if (ch == nch - 1) {
                // -------------------------------------------------------------- wave-private epilogue of this tile
                const int hi = lane >> 5, col_l = lane & 31;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    if (need_ld && PRE && mf + 1 < MF) epi_load(d0, h0, w0, inb, mf + 1, ev[(mf + 1) & 1]);
                    if (need_ld && !PRE) epi_load(d0, h0, w0, inb, mf, ev[0]);
                    // 1) fragment -> scratch, fragment-row major
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf) scr[row * SROW + nf * 32 + col_l] = acc[mf][nf][r];
                    }
                    // 2) scratch -> 16-byte vectors (same wave: LDS accesses of one wave are ordered)
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps) {
                        const int row = er0 + ps * RPW;
                        float v[KP];
                        const float4* sp = (const float4*)(scr + row * SROW + cg * KP);
#pragma unroll
                        for (int k4 = 0; k4 < KP / 4; ++k4) { const float4 t4 = sp[k4]; v[k4 * 4] = t4.x; v[k4 * 4 + 1] = t4.y; v[k4 * 4 + 2] = t4.z; v[k4 * 4 + 3] = t4.w; }
                        uint32_t vox;
                        const bool ok = epi_vox(d0, h0, w0, inb, mf, ps, vox);
                        {
                            if (EPI != 1) {
                                if (EPI == 2) {
                                    float rr[KP];
                                    unpack16<T>(ev[PRE ? (mf & 1) : 0][ps], rr);
#pragma unroll
                                    for (int q = 0; q < KP; ++q) v[q] += rr[q];
                                }
#pragma unroll
                                for (int q = 0; q < KP; ++q) { v[q] = ok ? Elem<T>::rnd(v[q]) : 0.f; s1[q] += v[q]; s2[q] += v[q] * v[q]; }
                            } else {
                                float xx[KP];
                                unpack16<T>(ev[PRE ? (mf & 1) : 0][ps], xx);
#pragma unroll
                                for (int q = 0; q < KP; ++q) {
                                    const float xn = (xx[q] - emu[q]) * ers[q];
                                    v[q] = Elem<T>::rnd((ok && xn > 0.f) ? v[q] : 0.f);
                                    s1[q] += v[q]; s2[q] += v[q] * xn;
                                }
                            }
                            const uint4 pk = pack16<T>(v);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pk), ors,
                                                                   ok ? (vox * (uint32_t)p.ldo + (uint32_t)col0) * (uint32_t)sizeof(T) : 0xFFFFFFFFu, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);             // keep the scheduler from interleaving all passes (register pressure)
                    }
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
                }
            }
            __syncthreads();
        }
        // statistics: ONE partial row per (block, wm) -- the block's tiles were accumulated in registers
        if (p.part) {
            // lanes with the same column group hold partial sums: reduce over the RPW row-lanes
#pragma unroll
            for (int q = 0; q < KP; ++q) {
#pragma unroll
                for (int o = CGW; o < 64; o <<= 1) { s1[q] += __shfl_xor(s1[q], o, 64); s2[q] += __shfl_xor(s2[q], o, 64); }
            }
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, 0x7FFFFFFF, 0x00020000);
            const uint32_t poff = (lane < CGW && cok) ? (uint32_t)(((((size_t)n * gridDim.x + bx) * WM + wm) * p.Cout + col0) * 8) : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < KP; q += 2) {
                u32x4_t pv;
                pv[0] = __float_as_uint(s1[q]); pv[1] = __float_as_uint(s2[q]); pv[2] = __float_as_uint(s1[q + 1]); pv[3] = __float_as_uint(s2[q + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(pv, prs, poff == 0xFFFFFFFFu ? poff : poff + q * 8, 0, 0);
            }
        }
    }
}

static int pc_grid_x(int tiles, int gy, int N) {                      // ~one persistent block per CU
    int gx = 256 / (gy * N > 0 ? gy * N : 1);
    if (gx < 1) gx = 1;
    return gx > tiles ? tiles : gx;
}

template <typename T, int WM, int MF, int WN, int NF>
int launch_pc(const IgemmParams& p, int epi, hipStream_t st) {
    const int tiles = ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const int gy = p.ntiles / (WN * NF);
    const int gx = pc_grid_x(tiles, gy, p.N);
    dim3 grid(gx, gy, p.N), block(512);
    const size_t smem = 2 * (size_t)HALO_BYTES + (((size_t)(p.a.C + p.b.C) * 8 + 15) / 16) * 16 + 4 * (size_t)PcCfg<MF, NF>::SCR_BYTES;
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    if (epi == 0 && p.res) {
        auto k = igemm_pc_kernel<T, WM, MF, WN, NF, 2>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    } else if (epi == 0) {
        auto k = igemm_pc_kernel<T, WM, MF, WN, NF, 0>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    } else {
        auto k = igemm_pc_kernel<T, WM, MF, WN, NF, 1>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    }
    return rs_check_launch();
}

template <typename T, int WM, int MF, int WN, int NF, int KSPLIT = 1>
int launch_cfg(const IgemmParams& p, int epi, hipStream_t st) {
    const int tiles = ((p.D + TD - 1) / TD) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    dim3 grid(tiles, p.ntiles / (WN * NF), p.N), block(256);
    const size_t smem = HALO_BYTES + (size_t)(p.a.C + p.b.C) * 2 * sizeof(float);
    if (smem > 160 * 1024) return RS_ERR_UNSUPPORTED;
    if (epi == 0) {
        auto k = igemm_kernel<T, WM, MF, WN, NF, 0, KSPLIT>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    } else {
        auto k = igemm_kernel<T, WM, MF, WN, NF, 1, KSPLIT>;
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
    }
    return rs_check_launch();
}

template <typename T>
int launch_dt(const IgemmParams& p, int epi, hipStream_t st) {
    const int bn32 = p.bn / 32;
    if (p.ntiles % bn32) return RS_ERR_ARG;
    switch (p.bn) {
        case 32: return launch_cfg<T, 2, 4, 1, 1, 2>(p, epi, st);
        case 64: return launch_cfg<T, 2, 4, 2, 1>(p, epi, st);
        case 128: return launch_cfg<T, 2, 4, 2, 2>(p, epi, st);
    }
    return RS_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------------------
// Weight packing into B-fragment order:  wp[chunk][tap][kstep][ntile][lane] (16 B each)
//   lane l of (chunk, tap, kstep, ntile) holds B[k .. k+KP-1][n] with n = ntile*32 + (l & 31),
//   k = kbase(chunk) + kstep*(KC/2) + (l >> 5)*KP.
// mode 0 (forward):  GEMM-K = forward Cin (two sources ka|kb => chunks are padded per source),
//                    GEMM-N = forward Cout, columns [0,na) from wa, [na,na+nb) from wb (fused conv1+shortcut);
//                    value = w[n][k][tap]                (weights are (Cout, Cin, 27) as in the state_dict)
// mode 1 (dgrad):    GEMM-K = forward Cout, source A = wa's rows (ka), source B = wb's rows (kb),
//                    GEMM-N = forward Cin (na);  value = w[k][n][26 - tap]   (flipped taps)
// One block re-orders one (chunk, kstep, ntile) slab = 32 columns x KC/2 k-values x 27 taps.  In the state_dict layout that
// slab is 32 (mode 0: one per column n) or KC/2 (mode 1: one per row k) contiguous runs of floats, so it is read with
// unit-stride loads into LDS and written back as 27 fully coalesced 1 KB fragment rows (64 lanes x 16 B).  The earlier
// gather-per-output-vector form read every float with a 108 B stride and cost 0.58 ms per step.
template <typename T>
__global__ __launch_bounds__(256) void pack_tile_kernel(PackBatch b, T* out) {
    constexpr int KC = Elem<T>::KC, KP = Elem<T>::KP, KH = KC / 2;
    constexpr int RS0 = KH * 27 + 1, RS1 = 32 * 27 + 1;               // odd run strides: conflict-free column reads
    extern __shared__ float sm[];
    int lo = 0, hi = b.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.blk_start[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const PackParams& q = b.q[lo];
    unsigned r = blockIdx.x - b.blk_start[lo];
    const int ks = r & 1; r >>= 1;
    const int ntile = r % q.ntiles, ch = r / q.ntiles;
    const int nchA = (q.ka + KC - 1) / KC;
    const bool isB = ch >= nchA;
    const int kbase = (isB ? ch - nchA : ch) * KC + ks * KH;
    const int klim = isB ? q.kb : q.ka;
    const int n0 = ntile * 32;
    const int tid = threadIdx.x;
    // loads are issued in batches of UB before any of them is consumed: a slab is only 54 (bf16) / 27 (f32) floats per
    // thread, so the kernel is pure latency unless many are in flight
    constexpr int PER = 32 * KH * 27 / 256, UB = PER % 18 == 0 ? 18 : 9;
    static_assert(PER % UB == 0, "slab size");
    // Loads go through buffer descriptors: an invalid element gets an out-of-range offset and the hardware returns zero, so
    // there is no branch around any load (a load inside a branch is waited for at the join, which would serialise the batch).
    const int na = q.na, nb = q.nb;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.wa, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(q.wb ? q.wb : q.wa), 0, 0x7FFFFFFF, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    if (q.mode == 0) {
        const int cin = q.ka + q.kb, kin0 = (isB ? q.ka : 0) + kbase;
        const int len = min(max(klim - kbase, 0), KH) * 27;
        const bool two = nb > 0 && n0 + 32 > na;                        // block-uniform: this tile has shortcut columns
        for (int it = 0; it < PER; it += UB) {
            uint32_t v[UB];
            int at[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = tid + (it + u) * 256;
                const int nl = e / (KH * 27), o = e - nl * (KH * 27), n = n0 + nl;
                at[u] = nl * RS0 + o;
                const bool ok = o < len;
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(ra, (ok && n < na) ? (uint32_t)((n * cin + kin0) * 27 + o) * 4u : OOB, 0, 0);
            }
            if (two) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = tid + (it + u) * 256;
                    const int nl = e / (KH * 27), o = e - nl * (KH * 27), n = n0 + nl;
                    const bool ok = o < len && n >= na && n < na + nb;
                    v[u] |= __builtin_amdgcn_raw_buffer_load_b32(rb, ok ? (uint32_t)(((n - na) * cin + kin0) * 27 + o) * 4u : OOB, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) sm[at[u]] = __builtin_bit_cast(float, v[u]);
        }
    } else {
        const __amdgpu_buffer_rsrc_t rw = isB ? rb : ra;
        // column sub-range (data gradient only): nb = first column << 16 | columns; na stays the row stride (forward Cin)
        const int noff = nb >> 16, ncols = nb ? (nb & 0xFFFF) : na;
        const int len = min(max(ncols - n0, 0), 32) * 27;
        for (int it = 0; it < PER; it += UB) {
            uint32_t v[UB];
            int at[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = tid + (it + u) * 256;
                const int kl = e / (32 * 27), o = e - kl * (32 * 27), k = kbase + kl;
                at[u] = kl * RS1 + o;
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(rw, (o < len && k < klim) ? (uint32_t)((k * na + noff + n0) * 27 + o) * 4u : OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) sm[at[u]] = __builtin_bit_cast(float, v[u]);
        }
    }
    __syncthreads();
    const int lane = tid & 63, nl = lane & 31, k0 = (lane >> 5) * KP;
    T* dst = out + b.vec_start[lo] * KP;
    for (int tap = tid >> 6; tap < 27; tap += 4) {
        float f[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j)
            f[j] = q.mode == 0 ? sm[nl * RS0 + (k0 + j) * 27 + tap] : sm[(k0 + j) * RS1 + nl * 27 + (26 - tap)];
        const size_t idx = ((((size_t)ch * 27 + tap) * 2 + ks) * q.ntiles + ntile) * 64 + lane;
        *(uint4*)(dst + idx * KP) = pack16<T>(f);
    }
}

}  // namespace

int rs_igemm_part_rows(int bn, int pc, int tiles, int n_cols, int N) {
    if (!pc) return tiles;                                            // classic kernel: one row per tile
    const int gy = (n_cols + bn - 1) / bn;                            // producer/consumer: one row per (persistent block, wm)
    return pc_grid_x(tiles, gy, N) * (bn == 32 ? 4 : 2);
}

int rs_launch_igemm(const IgemmParams& p, int dtype, int epi, hipStream_t st) {
    if (p.box > 0 && dtype == RS_BF16) return rs_launch_igemm_box(p, p.box, epi, st);
    if (p.pc == 3) return rs_igemm_kd_supported(p, dtype) ? rs_launch_igemm_kd(p, epi, st) : RS_ERR_UNSUPPORTED;
    if (p.pc == 2 && rs_igemm_ws_supported(p, dtype, epi)) return rs_launch_igemm_ws(p, epi, st);
    if (dtype == RS_BF16 && p.pc) {
        if (p.ntiles % (p.bn / 32)) return RS_ERR_ARG;
        switch (p.bn) {
            case 32: return launch_pc<bf16_t, 4, 2, 1, 1>(p, epi, st);
            case 64: return launch_pc<bf16_t, 2, 4, 2, 1>(p, epi, st);
            case 128: return launch_pc<bf16_t, 2, 4, 2, 2>(p, epi, st);
        }
        return RS_ERR_ARG;
    }
    if (dtype == RS_F32) return launch_dt<float>(p, epi, st);
    if (dtype == RS_BF16) return launch_dt<bf16_t>(p, epi, st);
    return RS_ERR_ARG;
}

size_t rs_packed_elems(int dtype, int ka, int kb, int ntiles) {
    const int KC = dtype == RS_F32 ? 16 : 32, KP = dtype == RS_F32 ? 4 : 8;
    const size_t nch = (size_t)((ka + KC - 1) / KC + (kb + KC - 1) / KC);
    return nch * 27 * 2 * ntiles * 64 * KP;
}

int rs_launch_pack_batch(PackBatch& b, int dtype, void* out, hipStream_t st) {
    const int KC = dtype == RS_F32 ? 16 : 32;
    unsigned blocks = 0;
    for (int i = 0; i < b.n; ++i) {
        b.blk_start[i] = blocks;
        blocks += (unsigned)(((b.q[i].ka + KC - 1) / KC + (b.q[i].kb + KC - 1) / KC) * b.q[i].ntiles * 2);
    }
    b.blk_start[b.n] = blocks;
    if (!blocks) return RS_OK;
    const size_t smem = (size_t)(dtype == RS_F32 ? 8 : 16) * (32 * 27 + 1) * sizeof(float) + 32 * sizeof(float);
    if (dtype == RS_F32) {
        hipLaunchKernelGGL(pack_tile_kernel<float>, dim3(blocks), dim3(256), smem, st, b, (float*)out);
    } else {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)pack_tile_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; }
        hipLaunchKernelGGL(pack_tile_kernel<bf16_t>, dim3(blocks), dim3(256), smem, st, b, (bf16_t*)out);
    }
    return rs_check_launch();
}

int rs_launch_pack(const PackParams& q, int dtype, void* out, hipStream_t st) {
    PackBatch b;
    memset(&b, 0, sizeof(b));
    b.n = 1;
    b.q[0] = q;
    b.vec_start[1] = rs_packed_elems(dtype, q.ka, q.kb, q.ntiles) / (dtype == RS_F32 ? 4 : 8);
    return rs_launch_pack_batch(b, dtype, out, st);
}
