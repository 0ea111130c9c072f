// Weight gradient of the 3x3x3 convolution for SMALL volumes (the 12^3 and 6^3 levels of the UNet: 256 - 320 channels, 432 - 3456 voxels per batch),
// gfx950, bf16.
//
//   dW[co][ci][tap] = sum_v dY[v][co] * x_hat[v + off(tap)][ci]          (GEMM: M = Cout, N = 27*Cin, K = voxels)
//
// Replaces the autograd weight-gradient of nn.Conv3d(k=3) (rsuper_train/model/dim3/conv_layers.py:29-38 under loss.backward(), train_ddp.py:349).
// At these levels K is tiny and M x N huge (256 x 6912 outputs for 3456 voxels): the tile-streaming kernels (conv3d_wgrad.hip, conv3d_wgrad2.hip) must
// split K over ~5 blocks per output tile to fill the chip, each block then sweeps 2 - 4 tiles of 4x4x16 voxels (W = 12 / 6 wastes 25 - 62 % of every
// 16-voxel row), writes a slab as large as its whole input, and a second launch sums the slabs: 36 - 40 us per layer at 65 - 300 TFLOP/s.
// Here a block owns 32 output rows x 32 input channels x ALL 27 taps for one slab of `dr` depth planes of one sample, and the slab lives in LDS whole:
//   * x_hat image: (dr + 2) x (H + 2) x (W + 2) rows of 64 B, zero border included (InstanceNorm + ReLU applied while staging, padding written as
//     zeros afterwards as everywhere); dY image: dr x H x W rows of 64 B, contiguous in the flat voxel order.
//   * the reduction runs over the FLAT voxel index of the slab, 16 voxels per MFMA k-step, whatever W is (no padded rows): the A fragment (dY) is a
//     contiguous run of 16 rows; the B fragment of tap (kd, kh, kw) gathers the 16 rows  rowaddr[voxel] + tapoff  -- ds_read_b64_tr_b16 takes one
//     address per lane, and rowaddr (the LDS offset of every slab voxel inside the x_hat image) is a 4-byte table built once per block.
//   * 4 waves, one per SIMD; wave g owns the taps g, g + 4, ... (7 accumulators); per k-step 1 A + 7 B fragments for 7 MFMAs.
//   * one staging phase, one barrier, one MFMA phase per block: at 1 - 2 blocks per CU the phases of different blocks overlap.
// Splits = N x (parts per sample): 2 - 4 slabs per output instead of 5 - 10; the slabs are summed by the reductions of conv3d_wgrad.hip.
#include "common.hpp"
#include "kernels.hpp"
#include "wgrad_frag.hpp"
#include <stdlib.h>

namespace {

constexpr int NT = 256, TPW = 7;
constexpr int SV_LDS_MAX = 158 * 1024;

struct SvGeom { int P, dr, rows_x, rows_y, nks, lds; };

// parts per sample: the fewest whose slab fits LDS, then more while the grid cannot fill the chip (never below two planes per part)
SvGeom sv_geometry(int N, int D, int H, int W, int nch, int Mtot) {
    SvGeom g = {0, 0, 0, 0, 0, 0};
    const int mg = (Mtot + 31) / 32;
    for (int P = 1; P <= D; ++P) {
        const int dr = (D + P - 1) / P;
        if ((P - 1) * dr >= D) continue;                          // a part would be empty
        const long rx = (long)(dr + 2) * (H + 2) * (W + 2), ry = (long)dr * H * W;
        const int nks = (int)((ry + 15) / 16);
        const long lds = 64 * (rx + (long)nks * 16) + 4L * nks * 16 + 1024;
        if (lds > SV_LDS_MAX) continue;
        g = {P, dr, (int)rx, (int)ry, nks, (int)lds};
        if ((long)nch * mg * N * P >= 192 || dr <= 2) break;
    }
    return g;
}

__global__ __launch_bounds__(NT, 2) void wgrad_sv_kernel(WgradParams p, int P, int dr, int rows_x, int nks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xim = smem;                                             // x_hat image
    char* yim = smem + (size_t)rows_x * 64;                       // dY image: nks * 16 rows
    uint32_t* rowaddr = (uint32_t*)(yim + (size_t)nks * 16 * 64); // [nks * 16] LDS byte offset of the voxel's row in the x_hat image (tap 0, 0, 0)
    float* mr_lds = (float*)(rowaddr + nks * 16);                 // [32][2] (sc, nb) of this chunk's channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchA = (p.xa.C + 31) / 32;
    const bool isB = (int)blockIdx.x >= nchA;
    const ConvSrc& xs = isB ? p.xb : p.xa;
    const int c0 = (isB ? blockIdx.x - nchA : blockIdx.x) * 32;
    const int cin_total = p.xa.C + p.xb.C;
    const int cin_base = (isB ? p.xa.C : 0) + c0;
    const int Mtot = p.ya.C + p.yb.C;
    const int m0 = blockIdx.y * 32;
    const int n = blockIdx.z / P, part = blockIdx.z % P;
    const int d0 = part * dr, d1 = min(p.D, d0 + dr);
    const int H = p.H, W = p.W, H2 = H + 2, W2 = W + 2;
    const int nvox = (d1 - d0) * H * W;                           // voxels of this slab (flat, contiguous in global memory)
    const bool norm = xs.mr != nullptr;

    // ---- normalisation constants of the chunk's 32 channels
    if (tid < 32) {
        const int c = c0 + tid;
        float sc = 1.f, nb = 0.f;
        if (norm && c < xs.C) { const float* m = xs.mr + ((size_t)n * xs.C + c) * 2; sc = m[1]; nb = -m[0] * m[1]; }
        mr_lds[2 * tid] = sc; mr_lds[2 * tid + 1] = nb;
    }
    __syncthreads();

    // ---- stage the x_hat image: vector v = (row, 16-byte slot); rows of the zero border and channels past C become zeros
#ifndef WGSV_SKIP_STAGE
    {
        const int slot = tid & 3;
        float sc_[8], nb_[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc_[j] = mr_lds[2 * (slot * 8 + j)]; nb_[j] = mr_lds[2 * (slot * 8 + j) + 1]; }
        const bool cok = c0 + slot * 8 < xs.C;
        const bf16_t* xsrc = (const bf16_t*)xs.x + c0 + slot * 8;
        const int planes = d1 - d0 + 2;
        // U rows per round and thread: all global loads of a round are in flight before the first is normalised (one load per iteration would expose
        // the memory latency 24 times in a row: measured 58 us per launch, 5 of them MFMA time)
        constexpr int U = 8;
        for (int r0 = tid >> 2; r0 < rows_x; r0 += U * (NT / 4)) {
            uint4 q[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * (NT / 4);
                const int pd = r / (H2 * W2), rem = r - pd * (H2 * W2);
                const int ph = rem / W2, pw = rem - ph * W2;
                const int d = d0 - 1 + pd, h = ph - 1, w = pw - 1;
                ok[u] = r < rows_x && cok && pd < planes && d >= 0 && d < p.D && h >= 0 && h < H && w >= 0 && w < W;
                const bf16_t* src = ok[u] ? xsrc + (size_t)(((n * p.D + d) * H + h) * W + w) * xs.ld : (const bf16_t*)xs.x;      // address select, no branch
                q[u] = *(const uint4*)src;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * (NT / 4);
                uint4 v = q[u];
                if (norm) v = norm_relu16<bf16_t>(v, sc_, nb_);
                const uint32_t m = ok[u] ? 0xFFFFFFFFu : 0u;     // border / padding rows stay zero AFTER the activation
                if (r < rows_x) *(uint4*)(xim + (size_t)r * 64 + slot * 16) = make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
            }
        }
        // dY image + row address table
        const int ym = m0 + slot * 8;
        const bf16_t* ysrc = nullptr; int yld = 0;
        if (ym < Mtot) {
            if (ym < p.ya.C) { ysrc = (const bf16_t*)p.ya.x + ym; yld = p.ya.ld; }
            else { ysrc = (const bf16_t*)p.yb.x + (ym - p.ya.C); yld = p.yb.ld; }
        }
        const size_t vbase = (size_t)((n * p.D + d0) * H) * W;
        for (int j0 = tid >> 2; j0 < nks * 16; j0 += U * (NT / 4)) {
            uint4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * (NT / 4);
                const bool ok = j < nvox && ysrc != nullptr;
                q[u] = *(const uint4*)(ok ? ysrc + (vbase + j) * (size_t)yld : (const bf16_t*)p.ya.x);
                if (!ok) q[u] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * (NT / 4);
                if (j >= nks * 16) continue;
                *(uint4*)(yim + (size_t)j * 64 + slot * 16) = q[u];
                if (slot == 0) {
                    const int jj = j < nvox ? j : 0;
                    const int dl = jj / (H * W), rem = jj - dl * (H * W);
                    const int h = rem / W, w = rem - h * W;
                    rowaddr[j] = (uint32_t)(((dl * H2 + h) * W2 + w) * 64);
                }
            }
        }
    }
#endif
    __syncthreads();

    // ---- MFMA phase: wave g owns the taps g, g + 4, ...; per k-step one A fragment (16 consecutive dY rows) and one gathered B fragment per tap
    f32x16_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int tapoff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int tl = wave + 4 * i;
        if (tl >= 27) tl = 26;                                    // wave 3's seventh slot: accumulates into a slot that is never stored
        const int kd = tl / 9, kh = (tl % 9) / 3, kw = tl % 3;
        tapoff[i] = ((kd * H2 + kh) * W2 + kw) * 64;
    }
    // lane part of a transposing fragment read (wgrad_frag.hpp): rows r1 = (g >> 1) * 8 + (q >> 2) and r1 + 4, byte column ((g & 1) * 16 + (q & 3) * 4) * 2
    const int q_ = lane & 15, g_ = lane >> 4;
    const int r1 = (g_ >> 1) * 8 + (q_ >> 2), colb = ((g_ & 1) * 16 + (q_ & 3) * 4) * 2;
    // LDS addresses as 32-bit integers: a generic pointer + per-lane offset costs a 64-bit add and an address-space conversion (null check + select)
    // per read -- 30 VALU instructions per MFMA in the first version of this loop (rocprofv3 SQ_INSTS_VALU), MFMA pipe 10 % busy
    typedef __attribute__((address_space(3))) v4s_t* lds_v4;
    auto tr8 = [&](uint32_t a) {
        union { v4s_t v; uint2 u; } t;
        t.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)a);
        return t.u;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t ya = lds0 + (uint32_t)(yim - smem) + (uint32_t)(r1 * 64 + colb);
    const uint32_t xb = lds0 + (uint32_t)(xim - smem) + (uint32_t)colb;
    const uint32_t* ra = rowaddr + r1;
    struct Ops { uint4 a; uint4 b[TPW]; };
    auto load_ops = [&](int ks, Ops& o) {
        const uint32_t o1 = xb + ra[ks * 16], o2 = xb + ra[ks * 16 + 4];
        const uint2 a_lo = tr8(ya + (uint32_t)ks * 1024u), a_hi = tr8(ya + (uint32_t)ks * 1024u + 256u);
        o.a = make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const uint2 lo = tr8(o1 + (uint32_t)tapoff[i]), hi = tr8(o2 + (uint32_t)tapoff[i]);
            o.b[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    auto mma_ops = [&](const Ops& o) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) mma32<bf16_t>(acc[i], o.a, o.b[i]);
    };
    // two k-steps per iteration on fixed register sets: the operands of step k + 1 are requested before the MFMAs of step k issue
    Ops e, o;
    load_ops(0, e);
#ifdef WGSV_SKIP_MMA
    if (nks > 1000000)
#endif
    for (int ks = 0; ks < nks; ks += 2) {
        if (ks + 1 < nks) load_ops(ks + 1, o);
        __builtin_amdgcn_sched_barrier(0);
        mma_ops(e);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < nks) load_ops(ks + 2, e);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < nks) mma_ops(o);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- this split's partial dW slab: ws[split][tap][m][cin]  (coalesced along cin)
    const int ci = c0 + (lane & 31);
    float* slab = p.ws + (size_t)blockIdx.z * 27 * Mtot * cin_total;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tl = wave + 4 * i;
        if (tl >= 27) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + cd_row32(r, lane);
#ifdef WGSV_SKIP_STORE
            if (acc[i][r] == 12345.678f)
#endif
            if (m < Mtot && ci < xs.C) slab[((size_t)tl * Mtot + m) * cin_total + cin_base + (lane & 31)] = acc[i][r];
        }
    }
}

}  // namespace

// splits (= N x parts per sample) of the small-volume kernel, 0 when it does not apply: bf16, a slab of at least one plane fits LDS, at most 64 tiles of
// 4x4x16 voxels in the batch (the levels where the tile-streaming kernels sweep < 4 tiles per block), 32-row groups of dY inside one source each
int rs_wgrad_sv_splits(int dtype, int Mtot, int Ya, int nch, int N, int D, int H, int W) {
    static const int off = getenv("RSUPER_WGRAD_SV") ? atoi(getenv("RSUPER_WGRAD_SV")) == 0 : 0;
    const int w2 = rs_wgrad2_min_tiles(-1);
    if (off || dtype != RS_BF16 || w2 == 0 || w2 >= (1 << 20)) return 0;      // 0 / "never": the tests force the tile-streaming kernels
    const long tiles = (long)N * ((D + 3) / 4) * ((H + 3) / 4) * ((W + 15) / 16);
    if (tiles > 64 || (Ya < Mtot && (Ya % 32))) return 0;
    // above 6^3 the kernel pays only while the whole launch is small: the wide fused [conv1 | shortcut] of up1.0 (512 rows x 18 input chunks at 12^3 = 288 output
    // tiles x N x depth parts blocks, each re-staging its slab) measured 180 us here against 120 us on the tile-streaming kernel; 256 -> 256 / 128 -> 2 x 256 at
    // 12^3 are equal on both, the 6^3 level is 20-25 % faster here (tools/r06_wgsv_exp.sh, profiles/r06_wgsv_exp.txt)
    static const long max_tiles12 = getenv("RSUPER_WGRAD_SV_MAX") ? atol(getenv("RSUPER_WGRAD_SV_MAX")) : 4096;
    if ((long)D * H * W > 216 && (long)Mtot * nch > max_tiles12) return 0;
    const SvGeom g = sv_geometry(N, D, H, W, nch, Mtot);
    return g.P > 0 ? N * g.P : 0;
}

int rs_launch_wgrad_sv(const WgradParams& p, hipStream_t st) {
    const int nch = (p.xa.C + 31) / 32 + (p.xb.C + 31) / 32;
    const int Mtot = p.ya.C + p.yb.C;
    const SvGeom g = sv_geometry(p.N, p.D, p.H, p.W, nch, Mtot);
    if (g.P <= 0 || p.splits != p.N * g.P) return RS_ERR_ARG;
    dim3 grid(nch, (Mtot + 31) / 32, p.N * g.P), block(NT);
    (void)hipFuncSetAttribute((const void*)wgrad_sv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds);
    hipLaunchKernelGGL(wgrad_sv_kernel, grid, block, g.lds, st, p, g.P, g.dr, g.rows_x, g.nks);
    return RS_OK;
}
