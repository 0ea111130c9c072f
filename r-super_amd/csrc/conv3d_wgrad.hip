// Weight gradient of the 3x3x3 convolution on MFMA (gfx950), channels-last activations.
//
//   dW[co][ci][tap] = sum_v dY[v][co] * x_hat[v + off(tap)][ci]          (GEMM: M = Cout, N = 27*Cin, K = voxels)
//
// Replaces the autograd weight-gradient of every nn.Conv3d(k=3) on the hot path
// (rsuper_train/model/dim3/conv_layers.py:29-38 under loss.backward(), train_ddp.py:349).
// x_hat = relu((x - mean) * rstd) is recomputed from x while the halo tile is staged (never stored).
// The reduction axis (voxels) is the slow axis of both operands in NDHWC, so MFMA fragments need a
// transposed read: bf16 uses ds_read_b64_tr_b16 (TR=1) or eight 16-bit LDS reads (TR=0, reference path);
// f32 MFMA (32x32x2) holds one k per lane and needs no transpose.
//
// Block = 4 waves, output tile = MT*32 output channels x NTAPS taps x 32 input channels, looping over the
// spatial tiles of its split; result accumulated into the f32 dW with atomics (dW pre-zeroed).
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int TD = 4, TH = 4, TW = 16;
constexpr int HH = TH + 2, HW = TW + 2;

typedef short v4s_t __attribute__((__vector_size__(4 * sizeof(short))));

template <typename T> struct WG;
template <> struct WG<bf16_t> { static constexpr int XP = 80; };     // x_hat row pitch: 32 ch * 2 B + 16
template <> struct WG<float> { static constexpr int XP = 144; };     // 32 ch * 4 B + 16

// Fragment = 16 bytes/lane for bf16 (8 k), 8 MFMAs worth of scalars for f32 are loaded on the fly.
template <int TR>
__device__ __forceinline__ uint4 frag_bf16(const char* base, int pitch, int lane) {
    // rows = 16 consecutive voxels starting at `base` (row pitch `pitch`), 32 channels (2 B each) at base.
    // wanted: lane l -> channel l&31, voxels (l>>5)*8 .. +7
    uint4 r;
    if (TR) {
        const int q = lane & 15, g = lane >> 4;
        const char* a = base + ((g >> 1) * 8 + (q >> 2)) * pitch + ((g & 1) * 16 + (q & 3) * 4) * 2;
        v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a));
        v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a + 4 * pitch));
        union { v4s_t v; uint2 u; } ul, uh;
        ul.v = lo; uh.v = hi;
        r = make_uint4(ul.u.x, ul.u.y, uh.u.x, uh.u.y);
    } else {
        const bf16_t* a = (const bf16_t*)(base + ((lane >> 5) * 8) * pitch) + (lane & 31);
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = *(const bf16_t*)((const char*)a + j * pitch);
        r = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
    return r;
}

template <typename T, int MT, int NTAPS, int TR>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = Elem<T>::KP;
    constexpr int XP = WG<T>::XP;
    constexpr int YP = MT * 32 * (int)sizeof(T) + 16;
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;               // halo depth rows
    constexpr int XROWS = HDN * HH * HW;
    constexpr int XV = 32 / KP;                                  // 16-B vectors per x row
    constexpr int YV = MT * 32 / KP;
    constexpr int WT = 4 / MT;                                   // tap stride between a wave's taps
    constexpr int TPW = (NTAPS + WT - 1) / WT;                   // taps per wave (max)
    char* xh = smem;
    char* yt = smem + XROWS * XP;
    float* mr_lds = (float*)(yt + 256 * YP);                     // [32][2] for this chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MT, wt = wave / MT;
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    const int mg = blockIdx.y % mgroups;
    const int kdg = NTAPS == 27 ? 0 : blockIdx.y / mgroups;      // kd handled by this block (9-tap config)
    const int m0 = mg * MT * 32;
    const bool norm = xs.mr != nullptr;

    const int tiles_w = (p.W + TW - 1) / TW, tiles_h = (p.H + TH - 1) / TH, tiles_d = (p.D + TD - 1) / TD;
    const int tiles = tiles_w * tiles_h * tiles_d * p.N;

    f32x16_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int cur_n = -1;
    for (int tile = blockIdx.z; tile < tiles; tile += p.splits) {
        int t = tile;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        const int td = t % tiles_d; t /= tiles_d;
        const int n = t;
        const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
        __syncthreads();                                         // previous tile consumed
        if (norm && n != cur_n) {
            if (tid < 64) mr_lds[tid] = (c0 + (tid >> 1)) < xs.C ? xs.mr[((size_t)n * xs.C + c0) * 2 + tid] : 0.f;
            cur_n = n;
            __syncthreads();
        }
        // ---- stage x_hat halo (norm + relu, zero padded)
        for (int v = tid; v < XROWS * XV; v += 256) {
            const int r = v / XV, s = v % XV;
            const int hd = r / (HH * HW);
            const int rem = r - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            const int d = d0 + hd + (NTAPS == 27 ? -1 : kdg - 1), h = h0 - 1 + hh, w = w0 - 1 + hw;
            const int c = c0 + s * KP;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W && c < xs.C) {
                q = *(const uint4*)((const T*)xs.x + ((((size_t)n * p.D + d) * p.H + h) * p.W + w) * (size_t)xs.ld + c);
                if (norm) {
                    float f[KP];
                    unpack16<T>(q, f);
#pragma unroll
                    for (int j = 0; j < KP; ++j) f[j] = fmaxf((f[j] - mr_lds[2 * (s * KP + j)]) * mr_lds[2 * (s * KP + j) + 1], 0.f);
                    q = pack16<T>(f);
                }
            }
            *(uint4*)(xh + r * XP + s * 16) = q;
        }
        // ---- stage dY tile [256 voxels][MT*32]
        for (int v = tid; v < 256 * YV; v += 256) {
            const int r = v / YV, s = v % YV;
            const int dd = r / (TH * TW), hh = (r / TW) % TH, ww = r % TW;
            const int d = d0 + dd, h = h0 + hh, w = w0 + ww;
            const int m = m0 + s * KP;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (d < p.D && h < p.H && w < p.W && m < Mtot) {
                const size_t vox = (((size_t)n * p.D + d) * p.H + h) * p.W + w;
                if (m < p.ya.C) q = *(const uint4*)((const T*)p.ya.x + vox * p.ya.ld + m);
                else q = *(const uint4*)((const T*)p.yb.x + vox * p.yb.ld + (m - p.ya.C));
            }
            *(uint4*)(yt + r * YP + s * 16) = q;
        }
        __syncthreads();
        // ---- MFMA over the 16 (d,h) rows of the tile; k = 16 voxels along w
        for (int row = 0; row < TD * TH; ++row) {
            const int dd = row / TH, hh = row % TH;
            const char* ybase = yt + (row * TW) * YP + wm * 32 * (int)sizeof(T);
            if constexpr (sizeof(T) == 2) {
                const uint4 afr = frag_bf16<TR>(ybase, YP, lane);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int tl = wt + i * WT;                 // tap index within this block's tap set
                    if (tl < NTAPS) {
                        const int kd = NTAPS == 27 ? tl / 9 : 0, kh = (tl % 9) / 3, kw = tl % 3;
                        const char* xb = xh + (((dd + kd) * HH + hh + kh) * HW + kw) * XP;
                        const uint4 bfr = frag_bf16<TR>(xb, XP, lane);
                        mma32<bf16_t>(acc[i], afr, bfr);
                    }
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {                // 8 MFMAs of K=2 voxels
                    const int wv = k2 * 2 + (lane >> 5);
                    const float a = *(const float*)(ybase + wv * YP + (lane & 31) * 4);
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const int tl = wt + i * WT;
                        if (tl < NTAPS) {
                            const int kd = NTAPS == 27 ? tl / 9 : 0, kh = (tl % 9) / 3, kw = tl % 3;
                            const float b = *(const float*)(xh + (((dd + kd) * HH + hh + kh) * HW + kw + wv) * XP + (lane & 31) * 4);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- accumulate into dW
    const int ci = c0 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tl = wt + i * WT;
        if (tl >= NTAPS) continue;
        const int tap = NTAPS == 27 ? tl : kdg * 9 + tl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + cd_row32(r, lane);
            if (m < Mtot && ci < xs.C) {
                float* dst = m < p.ya.C ? p.dwa + ((size_t)m * cin_total + cin_base + (lane & 31)) * 27 + tap
                                        : p.dwb + ((size_t)(m - p.ya.C) * cin_total + cin_base + (lane & 31)) * 27 + tap;
                atomicAdd(dst, acc[i][r]);
            }
        }
    }
}

template <typename T, int MT, int NTAPS, int TR>
int launch(const WgradParams& p, hipStream_t st) {
    constexpr int XP = WG<T>::XP;
    constexpr int YP = MT * 32 * (int)sizeof(T) + 16;
    constexpr int HDN = NTAPS == 27 ? TD + 2 : TD;
    const size_t smem = (size_t)HDN * HH * HW * XP + 256 * YP + 64 * sizeof(float);
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    const int mgroups = (Mtot + MT * 32 - 1) / (MT * 32);
    dim3 grid(nch, mgroups * (NTAPS == 27 ? 1 : 3), p.splits), block(256);
    auto k = wgrad_kernel<T, MT, NTAPS, TR>;
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    return rs_check_launch();
}

}  // namespace

int rs_wgrad_grid_y(int Mtot) { return Mtot <= 32 ? 1 : 3 * ((Mtot + 63) / 64); }

int rs_launch_wgrad(const WgradParams& p, int dtype, int use_tr, hipStream_t st) {
    const int Mtot = p.ya.C + p.yb.C;
    if (dtype == RS_F32) return Mtot <= 32 ? launch<float, 1, 27, 0>(p, st) : launch<float, 2, 9, 0>(p, st);
    if (dtype == RS_BF16) {
        if (use_tr) return Mtot <= 32 ? launch<bf16_t, 1, 27, 1>(p, st) : launch<bf16_t, 2, 9, 1>(p, st);
        return Mtot <= 32 ? launch<bf16_t, 1, 27, 0>(p, st) : launch<bf16_t, 2, 9, 0>(p, st);
    }
    return RS_ERR_ARG;
}
