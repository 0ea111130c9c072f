// extern "C" entry points declared in include/rsuper_hip.h: argument checking + kernel launches.
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include "common.hpp"
#include "kernels.hpp"
#include "misc.hpp"
#include "loss.hpp"
#include "morph.hpp"
#include "optim.hpp"
#include "../../include/rsuper_hip.h"

#define ST(s) ((hipStream_t)(s))

static inline bool dt_ok(int dt) { return dt == RS_F32 || dt == RS_BF16; }
static inline bool ch_ok(int C, int ld) { return C >= 0 && (C % 8) == 0 && (ld % 8) == 0 && ld >= C; }
static inline bool bn_ok(int bn) { return bn == 32 || bn == 64 || bn == 96 || bn == 128; }
static inline int ntiles_for(int n_cols, int bn) { const int per = bn / 32; return ((n_cols + bn - 1) / bn) * per; }
// GEMM-N columns of a packing request: mode 0 na + nb; mode 1 (data gradient) na, or the sub-range nb = first column << 16 | columns
static inline int pack_cols(int mode, int na, int nb) { return mode == 1 ? (nb ? (nb & 0xFFFF) : na) : na + nb; }
static inline bool pack_range_ok(int mode, int na, int nb) { return mode != 1 || nb == 0 || ((nb >> 16) % 32 == 0 && (nb & 0xFFFF) > 0 && (nb >> 16) + (nb & 0xFFFF) <= na); }

// ---- optimiser: split the tensor list into kernel-argument sized chunks
template <typename F>
static int for_chunks(int n, void* const* p, void* const* g, void* const* m, void* const* v, void* const* e, const size_t* numel, F f) {
    for (int i0 = 0; i0 < n; i0 += RS_MT_MAX) {
        MTChunk c;
        memset(&c, 0, sizeof(c));
        c.n = n - i0 < RS_MT_MAX ? n - i0 : RS_MT_MAX;
        int blk = 0;
        for (int i = 0; i < c.n; ++i) {
            c.p[i] = p ? p[i0 + i] : nullptr; c.g[i] = g ? g[i0 + i] : nullptr; c.m[i] = m ? m[i0 + i] : nullptr;
            c.v[i] = v ? v[i0 + i] : nullptr; c.ema[i] = e ? e[i0 + i] : nullptr;
            c.numel[i] = numel[i0 + i];
            c.blk_start[i] = blk;
            blk += rs_mt_blocks(numel[i0 + i]);
        }
        c.blk_start[c.n] = blk;
        if (blk == 0) continue;
        const int rc = f(c);
        if (rc) return rc;
    }
    return RS_OK;
}


extern "C" {

const char* rsuper_version(void) { return "rsuper-hip 0.1 (gfx950)"; }

int rsuper_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return RSUPER_ERR_NO_DEVICE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RSUPER_ERR_NO_DEVICE;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return RSUPER_ERR_NO_DEVICE;   // the calling thread's current device
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? RSUPER_OK : RSUPER_ERR_NO_DEVICE;
}

size_t rsuper_conv3_packed_elems(int dtype, int ka, int kb, int n_cols, int bn) {
    if (!dt_ok(dtype) || !bn_ok(bn)) return 0;
    return rs_packed_elems(dtype, ka, kb, ntiles_for(n_cols, bn));
}

int rsuper_conv3_pack_weights(int dtype, int mode, const float* wa, const float* wb, int ka, int kb, int na, int nb,
                              int bn, void* packed, void* stream) {
    if (!dt_ok(dtype) || !wa || !packed || (mode != 0 && mode != 1) || ka <= 0 || kb < 0 || na <= 0 || nb < 0) return RS_ERR_ARG;
    if ((kb > 0 || nb > 0) && mode == 1 && kb > 0 && !wb) return RS_ERR_ARG;
    if (mode == 0 && nb > 0 && !wb) return RS_ERR_ARG;
    if (!bn_ok(bn)) return RS_ERR_ARG;
    PackParams q;
    q.wa = wa; q.wb = wb; q.mode = mode; q.ka = ka; q.kb = kb; q.na = na; q.nb = nb;
    if (!pack_range_ok(mode, na, nb)) return RS_ERR_ARG;
    q.ntiles = ntiles_for(pack_cols(mode, na, nb), bn);
    return rs_launch_pack(q, dtype, packed, ST(stream));
}

int rsuper_conv3_pack_weights_batch(int dtype, int n, const int* host_desc, const float* const* host_wa, const float* const* host_wb,
                                    const size_t* host_out_elems, void* packed, void* stream) {
    // host_desc: n x 6 ints (mode, ka, kb, na, nb, bn); host_out_elems: n element offsets into `packed` (multiples of 8)
    if (!dt_ok(dtype) || n <= 0 || !host_desc || !host_wa || !host_wb || !host_out_elems || !packed) return RS_ERR_ARG;
    const int KP = dtype == RS_F32 ? 4 : 8;
    for (int i0 = 0; i0 < n; i0 += RS_PACK_BATCH_MAX) {
        PackBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - i0 < RS_PACK_BATCH_MAX ? n - i0 : RS_PACK_BATCH_MAX;
        const size_t base = host_out_elems[i0];
        for (int i = 0; i < b.n; ++i) {
            const int* d = host_desc + (size_t)(i0 + i) * 6;
            if ((d[0] != 0 && d[0] != 1) || d[1] <= 0 || d[2] < 0 || d[3] <= 0 || d[4] < 0 || !bn_ok(d[5])) return RS_ERR_ARG;
            PackParams& q = b.q[i];
            q.wa = host_wa[i0 + i]; q.wb = host_wb[i0 + i]; q.mode = d[0]; q.ka = d[1]; q.kb = d[2]; q.na = d[3]; q.nb = d[4];
            if (!pack_range_ok(d[0], d[3], d[4])) return RS_ERR_ARG;
            q.ntiles = ntiles_for(pack_cols(d[0], d[3], d[4]), d[5]);
            if (!q.wa || ((host_out_elems[i0 + i] - base) % KP)) return RS_ERR_ARG;
            b.vec_start[i] = (host_out_elems[i0 + i] - base) / KP;
        }
        // entries are contiguous: the end of the last one closes the table
        const int* dl = host_desc + (size_t)(i0 + b.n - 1) * 6;
        b.vec_start[b.n] = b.vec_start[b.n - 1] + rs_packed_elems(dtype, dl[1], dl[2], ntiles_for(pack_cols(dl[0], dl[3], dl[4]), dl[5])) / KP;
        for (int i = 0; i + 1 < b.n; ++i) {   // contiguity check
            const int* d = host_desc + (size_t)(i0 + i) * 6;
            if (b.vec_start[i] + rs_packed_elems(dtype, d[1], d[2], ntiles_for(pack_cols(d[0], d[3], d[4]), d[5])) / KP != b.vec_start[i + 1]) return RS_ERR_ARG;
        }
        const int rc = rs_launch_pack_batch(b, dtype, (char*)packed + base * (dtype == RS_F32 ? 4 : 2), ST(stream));
        if (rc) return rc;
    }
    return RS_OK;
}

int rsuper_conv3_tiles(int D, int H, int W) { return ((D + 3) / 4) * ((H + 3) / 4) * ((W + 15) / 16); }

static int g_variant = 3;
int rsuper_conv3_wgrad2_min_tiles(int t) { return rs_wgrad2_min_tiles(t); }

int rsuper_conv3_variant(int v) {
    if (v >= 0 && v <= 8 && v != 5) g_variant = v;                   // 5 was the second-generation producer/consumer kernel (measured equal, removed)
    return g_variant;
}
// variant 2 (auto, default): producer/consumer kernel where it measured faster on MI355X -- data-gradient launches with
// bn <= 64 (its epilogue operand prefetch), all 32-column launches, and small grids (persistent blocks fill the chip);
// classic kernel elsewhere.
// variant 3 (default): as variant 2, and 32-column launches over a single 32-channel K chunk (the 32 -> 32 layers at full
// resolution) run the weight-stationary kernel (conv3d_igemm_ws.hip), which writes the same partial rows as the
// producer/consumer kernel.  variant 4: weight-stationary kernel for every bf16 32-column launch (multi-chunk too; tests).
static bool use_pc(int dtype, int epi, int bn, int tiles_total) {
    if (dtype != RS_BF16) return false;
    if (g_variant == 4) return bn == 32 || (bn <= 64 && (epi == 1 || tiles_total <= 1024));
    if (g_variant == 2 || g_variant == 3) return bn <= 64 && (epi == 1 || bn == 32 || tiles_total <= 1024);
    return g_variant == 1;
}

// Volume-fitted K-split kernel (conv3d_igemm_box.hip) for launches that cannot fill the chip with 4x4x16-voxel tiles: bf16, more than
// 32 columns, and fewer than 512 classic work items (the 24^3 / 12^3 / 6^3 levels at batch 2).  Shapes 1 / 2 (4x4x8 / 4x4x4 boxes) take
// 64-column blocks; shape 3 (volumes of at most 6x6x6: one box per sample, the reduction split over blocks through the registered
// workspace, second pass box_splitk_epilogue_kernel) takes 32-column blocks.  Variants 6 / 7 force the kernel for every bf16 launch
// (6: shape chosen per volume, 7: the 4x4x4 box) -- test paths.
// one registered workspace PER DEVICE (indexed by the current HIP device of the calling thread: a host that drives several GPUs from one process
// registers one on each); launches on two streams of the same device share it and must not overlap 6^3-level convolutions
static const int RS_MAX_DEVICES = 64;
static void* g_ws_dev[RS_MAX_DEVICES] = {nullptr};
static size_t g_ws_bytes_dev[RS_MAX_DEVICES] = {0};
static int ws_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= RS_MAX_DEVICES) return 0;
    return d;
}
#define g_ws (g_ws_dev[ws_dev()])
#define g_ws_bytes (g_ws_bytes_dev[ws_dev()])
static const size_t BOXC_WS_BYTES = (size_t)256 * 216 * 32 * 4;    // nsplit x N x 216 x n_cols floats while N x ceil(n_cols / 32) <= 256 (rs_box_nsplit)
// Bytes the split shape writes into the workspace: nsplit x N x voxels x n_cols floats.  rs_box_nsplit deals at most 256 / (N x groups) splits (at least
// one), so beyond N x groups = 256 the requirement grows with N x n_cols without bound -- the shape is only taken while it fits what the caller registered.
static size_t boxc_ws_need(int nsplit, int N, int D, int H, int W, int n_cols) {
    return (size_t)nsplit * (size_t)N * (size_t)(D * H * W) * (size_t)n_cols * sizeof(float);
}
static bool boxc_fits(int N, int D, int H, int W, int n_cols) {
    const int groups = (n_cols + 31) / 32;
    const long ng = (long)N * groups;
    const int smax = ng >= 256 ? 1 : (int)(256 / ng);
    return g_ws && boxc_ws_need(smax, N, D, H, W, n_cols) <= g_ws_bytes;
}
static int box_shape(int dtype, int N, int D, int H, int W, int n_cols) {
    if (dtype != RS_BF16) return 0;
    const bool forced = g_variant == 6 || g_variant == 7;
    static const int off = getenv("RSUPER_NO_BOX") ? atoi(getenv("RSUPER_NO_BOX")) : 0;      // 1: no box kernel, 2: no split shape
    if (!forced && (g_variant != 3 || off == 1 || n_cols <= 32)) return 0;
    if (g_variant != 7 && off != 2 && D <= 6 && H <= 6 && W <= 6 && boxc_fits(N, D, H, W, n_cols)) return 3;
    if (g_variant == 7) return 2;
    if (!forced && (long)N * rsuper_conv3_tiles(D, H, W) * ((n_cols + 127) / 128) >= 512) return 0;
    return rs_box_config(N, D, H, W, n_cols);
}
static int box_for(int dtype, int bn, int N, int D, int H, int W, int n_cols) {
    const int cfg = box_shape(dtype, N, D, H, W, n_cols);
    return (cfg && bn == (cfg == 3 ? 32 : 64)) ? cfg : 0;
}

// Depth-reuse kernel (conv3d_igemm_kd.hip, round 5): bf16 launches with more than 32 columns whose volume gives the persistent blocks enough 4 x 8 x 16-voxel
// tiles (the full-resolution level at batch 2: 3456 tiles; RSUPER_KD_MIN_TILES).  A pure function of (dtype, epi, shape, columns) so that the block width
// (rsuper_conv3_kd_bn -> pick_bn), the partial-row count and the launch agree.  Forward launches (normalised sources, staging with arithmetic in registers)
// take 64-column blocks; data-gradient launches (raw dY) 64 / 96 / 128.  Variant 8 forces it wherever its limits allow (tests).
static int kd_bn_for(int dtype, int epi, int N, int D, int H, int W, int n_cols, int src_flags) {
    if (dtype != RS_BF16 || n_cols <= 32) return 0;
    if (src_flags & 1) return 0;        // one normalised and one raw source: the depth-reuse kernel stages both sources the same way (rs_igemm_kd_supported)
    static const int off = getenv("RSUPER_KD") ? atoi(getenv("RSUPER_KD")) == 0 : 0;
    static const int min_tiles = getenv("RSUPER_KD_MIN_TILES") ? atoi(getenv("RSUPER_KD_MIN_TILES")) : 400;
    static const int dgrad = getenv("RSUPER_KD_DGRAD") ? atoi(getenv("RSUPER_KD_DGRAD")) : 0;
    static const int wide = getenv("RSUPER_KD_WIDE") ? atoi(getenv("RSUPER_KD_WIDE")) : 0;
    if (g_variant != 8 && (g_variant != 3 || off)) return 0;
    if ((long)N * D * H * W >= (1l << 24) || D > 1023 || H > 1023 || W > 1023) return 0;
    if (g_variant != 8) {
        // measured (profiles/r05_conv_layers.txt, same box): forward launches of exactly 33 .. 64 columns at 96^3 / 48^3 gain 10-11 % (up4.0 648 -> 576 us,
        // 64 -> 64 @48^3 60 -> 54 us); 128-column forwards (two column blocks) and the data gradients do not beat the 128-column classic / producer-consumer
        // kernels yet -- they stay selectable (RSUPER_KD_DGRAD=1) and run every parity case under variant 8
        const long tiles = (long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16);
        if (tiles < min_tiles) return 0;
        if (epi == 0 && n_cols > 64) return 0;
        if (epi == 1 && !dgrad) return 0;
    }
    // one column fragment per wave (64-column blocks): two fragments per wave (96 / 128-column blocks, raw sources by LDS-DMA) do not fit the register file
    // without spills yet and measured slower (up4.0 data gradient: 96-column blocks 733 us against 705 us for 2 x 64); RSUPER_KD_WIDE=1 selects them
    if (epi == 0 || !wide) return 64;
    const int r = n_cols % 128;
    return (n_cols > 128 || r == 0) ? 128 : r > 96 ? 128 : r > 64 ? 96 : 64;
}
int rsuper_conv3_kd_bn(int dtype, int epi, int N, int D, int H, int W, int n_cols, int src_flags) {
    if (!dt_ok(dtype) || (epi != 0 && epi != 1) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || n_cols <= 0) return 0;
    return kd_bn_for(dtype, epi, N, D, H, W, n_cols, src_flags);
}

int rsuper_conv3_set_workspace(void* ptr, size_t bytes) {
    g_ws = ptr; g_ws_bytes = ptr ? bytes : 0;
    return RS_OK;
}
size_t rsuper_conv3_workspace_bytes(void) { return BOXC_WS_BYTES; }

int rsuper_conv3_box_bn(int dtype, int N, int D, int H, int W, int n_cols) {
    const int cfg = box_shape(dtype, N, D, H, W, n_cols);
    return cfg == 3 ? 32 : cfg ? 64 : 0;
}

int rsuper_conv3_part_rows(int dtype, int epi, int N, int D, int H, int W, int n_cols, int bn, int src_flags) {
    if (bn == kd_bn_for(dtype, epi, N, D, H, W, n_cols, src_flags)) return rs_igemm_kd_part_rows(bn, N, D, H, W, n_cols);
    if (const int cfg = box_for(dtype, bn, N, D, H, W, n_cols)) return rs_box_part_rows(cfg, D, H, W);
    return rs_igemm_part_rows(bn, use_pc(dtype, epi, bn, N * rsuper_conv3_tiles(D, H, W)) ? 1 : 0, rsuper_conv3_tiles(D, H, W), n_cols, N);
}

// the persistent strided forward (conv3d_igemm_s2k.hip) takes bf16 forward GEMMs with a 16-channel-aligned source; RSUPER_S2K=0: the parity-class kernel everywhere
static bool s2k_on() {                                                       // read per call (tests switch it inside one process)
    const char* e = getenv("RSUPER_S2K");
    return !(e && atoi(e) == 0);
}
static bool s2d_on() {                                                       // RSUPER_S2D=0: the parity-class data gradient
    const char* e = getenv("RSUPER_S2D");
    return !(e && atoi(e) == 0);
}
static IgemmParams s2_shape(int n_cols, int Ca, int Cb, int N, int FD, int FH, int FW) {
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.ntiles = ntiles_for(n_cols, 64); p.bn = 64;
    p.N = N; p.D = (FD + 1) / 2; p.H = (FH + 1) / 2; p.W = (FW + 1) / 2; p.Cout = n_cols;
    p.a.C = Ca; p.b.C = Cb;
    return p;
}
int rsuper_conv3_s2_part_rows(int dtype, int mode, int Ca, int Cb, int n_cols, int N, int FD, int FH, int FW) {
    if ((mode != 1 && mode != 2) || FD <= 0 || FH <= 0 || FW <= 0 || N <= 0 || n_cols <= 0) return 0;
    const int OD = (FD + 1) / 2, OH = (FH + 1) / 2, OW = (FW + 1) / 2, td = mode == 2 ? 2 : 4;
    if (mode == 1 && s2k_on()) {
        IgemmParams p = s2_shape(n_cols, Ca, Cb, N, FD, FH, FW);
        static const float one = 1.f;
        p.a.mr = &one;                                                       // "normalised source" for the support query
        if (rs_igemm_s2k_supported(p, dtype, FD, FH, FW)) return rs_igemm_s2k_part_rows(p.ntiles, n_cols, N, OD, OH, OW);
    }
    if (mode == 2 && s2d_on()) {
        IgemmParams p = s2_shape(n_cols, Ca, Cb, N, FD, FH, FW);
        static const float one = 1.f;
        p.ea.mr = &one; p.ea.x = &one;
        if (rs_igemm_s2d_supported(p, dtype, FD, FH, FW)) return rs_igemm_s2d_part_rows(n_cols, N, OD, OH, OW);
    }
    return ((OD + td - 1) / td) * ((OH + 3) / 4) * ((OW + 15) / 16);
}
// Stride-2 forward (mode 1) / its data gradient (mode 2) on the parity-class kernel (conv3d_igemm_s2.hip); 64-column blocks
int rsuper_conv3_igemm_s2(int dtype, int mode, const void* xa, int lda, int Ca, const float* mra,
                          const void* xb, int ldb, int Cb, const void* packed, int n_cols, int N, int FD, int FH, int FW,
                          void* out, int ldo, float* part, const void* exa, int elda, const float* emra, void* stream) {
    if (!dt_ok(dtype) || (mode != 1 && mode != 2) || !xa || !packed || !out) return RS_ERR_ARG;
    if (!ch_ok(Ca, lda) || Ca == 0 || (Cb > 0 && (!xb || !ch_ok(Cb, ldb))) || n_cols <= 0 || (n_cols % 8) || (ldo % 8) || ldo < n_cols) return RS_ERR_ARG;
    if (N <= 0 || FD <= 0 || FH <= 0 || FW <= 0) return RS_ERR_ARG;
    if (mode == 1 && (Cb > 0 || !mra)) return RS_ERR_ARG;                      // forward: one normalised source
    if (mode == 2 && (mra || !exa || !emra || (elda % 8) || elda < n_cols)) return RS_ERR_ARG;
    {
        const unsigned long long vox = (unsigned long long)N * FD * FH * FW, es = dtype == RS_F32 ? 4 : 2;
        const int lds[4] = {lda, Cb > 0 ? ldb : 0, ldo, mode == 2 ? elda : 0};
        for (int i = 0; i < 4; ++i)
            if (vox * (unsigned long long)lds[i] * es >= (1ull << 32)) return RS_ERR_UNSUPPORTED;
    }
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.a = {xa, lda, Ca, mode == 1 ? mra : nullptr};
    p.b = {xb, ldb, Cb > 0 ? Cb : 0, nullptr};
    p.wp = packed; p.ntiles = ntiles_for(n_cols, 64); p.bn = 64;
    p.N = N; p.D = (FD + 1) / 2; p.H = (FH + 1) / 2; p.W = (FW + 1) / 2; p.Cout = n_cols;
    p.out = out; p.ldo = ldo; p.part = part;
    p.ea = {exa, elda, n_cols, emra};
    if (mode == 1 && s2k_on() && rs_igemm_s2k_supported(p, dtype, FD, FH, FW)) return rs_launch_igemm_s2k(p, FD, FH, FW, ST(stream));
    if (mode == 2 && s2d_on() && rs_igemm_s2d_supported(p, dtype, FD, FH, FW)) return rs_launch_igemm_s2d(p, FD, FH, FW, ST(stream));
    return rs_launch_igemm_s2(p, dtype, mode, FD, FH, FW, ST(stream));
}

static int conv3_igemm_impl(int dtype, int epi, const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* packed, int n_cols, int bn, int N, int D, int H, int W,
                       void* out, int ldo, const void* res, int ldr, float* part,
                       const void* exa, int elda, int eCa, const float* emra,
                       const void* exb, int eldb, int eCb, const float* emrb, void* stream, int out_split, long long out_part) {
    if (!dt_ok(dtype) || (epi != 0 && epi != 1) || !xa || !packed || !out) return RS_ERR_ARG;
    if (out_split) {        // two output tensors: columns [0, out_split) and [out_split, n_cols), row stride ldo each
        if (out_split < 0 || (out_split % 32) || out_split >= n_cols || out_part <= 0 || (out_part % 8) || ldo < out_split || ldo < n_cols - out_split) return RS_ERR_ARG;
        if ((unsigned long long)(out_part + (long long)N * D * H * W * ldo) * 2ull >= (1ull << 32)) return RS_ERR_UNSUPPORTED;
    }
    if (!ch_ok(Ca, lda) || Ca == 0 || (Cb > 0 && (!xb || !ch_ok(Cb, ldb))) || n_cols <= 0 || (n_cols % 8) || (ldo % 8) || (!out_split && ldo < n_cols)) return RS_ERR_ARG;
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return RS_ERR_ARG;
    if (res && ((ldr % 8) || ldr < n_cols)) return RS_ERR_ARG;
    if (epi == 1 && (!exa || !emra || eCa + eCb != n_cols || (eCb > 0 && (!exb || !emrb)))) return RS_ERR_ARG;
    if (!bn_ok(bn)) return RS_ERR_ARG;
    if (dtype == RS_F32 && bn >= 96) return RS_ERR_UNSUPPORTED;    // register budget: f32 parity mode uses bn <= 64
    {   // buffer-addressed staging / epilogue: every activation tensor must stay below 4 GiB (32-bit byte offsets)
        const unsigned long long vox = (unsigned long long)N * D * H * W, es = dtype == RS_F32 ? 4 : 2;
        const int lds[6] = {lda, Cb > 0 ? ldb : 0, ldo, res ? ldr : 0, epi == 1 ? elda : 0, (epi == 1 && eCb > 0) ? eldb : 0};
        for (int i = 0; i < 6; ++i)
            if (vox * (unsigned long long)lds[i] * es >= (1ull << 32)) return RS_ERR_UNSUPPORTED;
    }
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.a = {xa, lda, Ca, mra};
    p.b = {xb, ldb, Cb > 0 ? Cb : 0, Cb > 0 ? mrb : nullptr};
    p.wp = packed; p.ntiles = ntiles_for(n_cols, bn); p.bn = bn;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = n_cols;
    p.out = out; p.ldo = ldo; p.res = res; p.ldr = ldr; p.part = part;
    p.ea = {exa, elda, eCa, emra};
    p.eb = {exb, eldb, eCb, emrb};
    p.box = box_for(dtype, bn, N, D, H, W, n_cols);
    if (p.box == 3) {
        p.ws = (float*)g_ws; p.nsplit = rs_box_nsplit(N, n_cols, (Ca + 31) / 32 + (Cb + 31) / 32);
        if (!g_ws || boxc_ws_need(p.nsplit, N, D, H, W, n_cols) > g_ws_bytes) return RS_ERR_ARG;      // never write past the registered workspace
    }
    p.pc = use_pc(dtype, epi, bn, N * rsuper_conv3_tiles(D, H, W)) ? 1 : 0;
    // the depth-reuse kernel is selected from the shape AND the sources of this launch: a launch it cannot take (one normalised + one raw
    // source) keeps the kernel the same bn gets everywhere else, and rsuper_conv3_part_rows(..., src_flags = 1) sizes `part` for that kernel
    const int src_flags = (Cb > 0 && (mra != nullptr) != (mrb != nullptr)) ? 1 : 0;
    if (bn == kd_bn_for(dtype, epi, N, D, H, W, n_cols, src_flags)) { p.pc = 3; p.box = 0; }
    else if (bn == 96) return RS_ERR_UNSUPPORTED;                  // 96-column blocks exist on the depth-reuse kernel only
    if (p.pc && bn == 32 && (g_variant == 4 || (g_variant == 3 && (Ca + 31) / 32 + (Cb + 31) / 32 == 1))) p.pc = 2;
    if (out_split) {
        if (p.pc != 3 || !rs_igemm_kd_supported(p, dtype)) return RS_ERR_UNSUPPORTED;      // the split store exists in the depth-reuse kernel's epilogue only
        p.out_split = out_split; p.out_part = out_part;
    }
    return rs_launch_igemm(p, dtype, epi, ST(stream));
}
int rsuper_conv3_igemm(int dtype, int epi, const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* packed, int n_cols, int bn, int N, int D, int H, int W,
                       void* out, int ldo, const void* res, int ldr, float* part,
                       const void* exa, int elda, int eCa, const float* emra,
                       const void* exb, int eldb, int eCb, const float* emrb, void* stream) {
    return conv3_igemm_impl(dtype, epi, xa, lda, Ca, mra, xb, ldb, Cb, mrb, packed, n_cols, bn, N, D, H, W, out, ldo, res, ldr, part,
                            exa, elda, eCa, emra, exb, eldb, eCb, emrb, stream, 0, 0);
}
int rsuper_conv3_igemm_split_out(int dtype, int epi, const void* xa, int lda, int Ca, const float* mra,
                                 const void* xb, int ldb, int Cb, const float* mrb,
                                 const void* packed, int n_cols, int bn, int N, int D, int H, int W,
                                 void* out, int ldo, const void* res, int ldr, float* part,
                                 const void* exa, int elda, int eCa, const float* emra,
                                 const void* exb, int eldb, int eCb, const float* emrb, int out_split, long long out_part, void* stream) {
    if (out_split <= 0) return RS_ERR_ARG;
    return conv3_igemm_impl(dtype, epi, xa, lda, Ca, mra, xb, ldb, Cb, mrb, packed, n_cols, bn, N, D, H, W, out, ldo, res, ldr, part,
                            exa, elda, eCa, emra, exb, eldb, eCb, emrb, stream, out_split, out_part);
}

int rsuper_conv3_wgrad_splits(int dtype, int Ca, int Cb, int Ya, int Yb, int N, int D, int H, int W) {
    if (!dt_ok(dtype) || Ca <= 0 || Cb < 0 || Ya <= 0 || Yb < 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return -1;
    const int nch = (Ca + 31) / 32 + (Cb + 31) / 32, Mtot = Ya + Yb;
    // the same predicate the launcher evaluates with the real (Ya, Yb): a fused [conv1 | shortcut] whose first source is not a multiple of 32 rows
    // does not take the small-volume kernel, so its split count must not be sized for it
    if (const int sv = rs_wgrad_sv_splits(dtype, Mtot, Ya, nch, N, D, H, W)) return sv;        // small volumes: N x (depth parts per sample)
    return rs_wgrad_splits(dtype, Mtot, nch, N * rsuper_conv3_tiles(D, H, W));
}

static int conv3_wgrad_impl(int reduce, int dtype, int use_tr, const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                       float* dwa, float* dwb, float* workspace, int N, int D, int H, int W, int splits, void* stream) {
    if (!dt_ok(dtype) || !xa || !ya || !dwa || !workspace || !ch_ok(Ca, lda) || Ca == 0 || !ch_ok(Ya, ldya) || Ya == 0) return RS_ERR_ARG;
    if (Cb > 0 && (!xb || !ch_ok(Cb, ldb))) return RS_ERR_ARG;
    if (Yb > 0 && (!yb || !dwb || !ch_ok(Yb, ldyb))) return RS_ERR_ARG;
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || splits <= 0) return RS_ERR_ARG;
    {   // same 4 GiB limit for the buffer-addressed x_hat staging
        const unsigned long long vox = (unsigned long long)N * D * H * W, es = dtype == RS_F32 ? 4 : 2;
        if (vox * (unsigned long long)lda * es >= (1ull << 32) || (Cb > 0 && vox * (unsigned long long)ldb * es >= (1ull << 32))) return RS_ERR_UNSUPPORTED;
        // ... and for dY: the second-generation kernel reads it through buffer resources too (32-bit byte offsets, num_records = voxels x row bytes)
        if (vox * (unsigned long long)ldya * es >= (1ull << 32) || (Yb > 0 && vox * (unsigned long long)ldyb * es >= (1ull << 32))) return RS_ERR_UNSUPPORTED;
    }
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.xa = {xa, lda, Ca, mra};
    p.xb = {xb, ldb, Cb > 0 ? Cb : 0, Cb > 0 ? mrb : nullptr};
    p.ya = {ya, ldya, Ya, nullptr};
    p.yb = {yb, ldyb, Yb > 0 ? Yb : 0, nullptr};
    p.dwa = dwa; p.dwb = dwb; p.ws = workspace; p.N = N; p.D = D; p.H = H; p.W = W; p.splits = splits;
    return rs_launch_wgrad(p, dtype, use_tr, ST(stream), reduce != 0);
}

int rsuper_conv3_wgrad(int dtype, int use_tr, const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                       float* dwa, float* dwb, float* workspace, int N, int D, int H, int W, int splits, void* stream) {
    return conv3_wgrad_impl(1, dtype, use_tr, xa, lda, Ca, mra, xb, ldb, Cb, mrb, ya, ldya, Ya, yb, ldyb, Yb, dwa, dwb, workspace, N, D, H, W, splits, stream);
}
int rsuper_conv3_wgrad_partial(int dtype, int use_tr, const void* xa, int lda, int Ca, const float* mra,
                               const void* xb, int ldb, int Cb, const float* mrb,
                               const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                               float* workspace, int N, int D, int H, int W, int splits, void* stream) {
    float dummy;                                                   // the partial kernel never touches dW
    return conv3_wgrad_impl(0, dtype, use_tr, xa, lda, Ca, mra, xb, ldb, Cb, mrb, ya, ldya, Ya, yb, ldyb, Yb, &dummy, Yb > 0 ? &dummy : nullptr,
                            workspace, N, D, H, W, splits, stream);
}
int rsuper_conv3_wgrad_reduce(const float* workspace, int splits, int Cin, int Ya, int Yb, float* dwa, float* dwb, void* stream) {
    if (!workspace || splits <= 0 || Cin <= 0 || (Cin % 8) || Ya <= 0 || Yb < 0 || !dwa || (Yb > 0 && !dwb)) return RS_ERR_ARG;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.xa.C = Cin; p.ya.C = Ya; p.yb.C = Yb; p.dwa = dwa; p.dwb = dwb; p.ws = (float*)workspace; p.splits = splits;
    return rs_launch_wgrad_reduce(p, ST(stream));
}

int rsuper_conv3_wgrad_reduce_batch(int n, const void* const* workspaces, const int* splits, const int* Cin, const int* Ya, const int* Yb,
                                     void* const* dwa, void* const* dwb, void* stream) {
    if (n < 0 || (n > 0 && (!workspaces || !splits || !Cin || !Ya || !Yb || !dwa || !dwb))) return RS_ERR_ARG;
    for (int i0 = 0; i0 < n; i0 += RS_REDUCE_BATCH_MAX) {
        ReduceBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - i0 < RS_REDUCE_BATCH_MAX ? n - i0 : RS_REDUCE_BATCH_MAX;
        for (int j = 0; j < b.n; ++j) {
            const int i = i0 + j;
            if (!workspaces[i] || splits[i] <= 0 || Cin[i] <= 0 || (Cin[i] % 8) || Ya[i] <= 0 || Yb[i] < 0 || !dwa[i] || (Yb[i] > 0 && !dwb[i])) return RS_ERR_ARG;
            b.e[j] = {(const float*)workspaces[i], (float*)dwa[i], (float*)dwb[i], splits[i], Ya[i] + Yb[i], Ya[i], Cin[i], 0u};
        }
        const int rc = rs_launch_wgrad_reduce_batch(b, ST(stream));
        if (rc) return rc;
    }
    return RS_OK;
}

int rsuper_conv3_wgrad_reduce_batch_stats(int n, const void* const* workspaces, const int* splits, const int* Cin, const int* Ya, const int* Yb,
                                           void* const* dwa, void* const* dwb, int nstats, const void* const* parts, const int* sN, const int* snblk,
                                           const int* sC, const double* scnt, const int* smode, const int* ssplit, void* const* souts, float eps, void* stream) {
    if (nstats < 0 || nstats > RS_REDUCE_STATS_MAX || n < 1 || n > RS_REDUCE_BATCH_MAX) return RS_ERR_ARG;
    if (!workspaces || !splits || !Cin || !Ya || !Yb || !dwa || !dwb) return RS_ERR_ARG;
    if (nstats > 0 && (!parts || !sN || !snblk || !sC || !scnt || !smode || !ssplit || !souts)) return RS_ERR_ARG;
    ReduceBatch b;
    memset(&b, 0, sizeof(b));
    b.n = n;
    for (int i = 0; i < n; ++i) {
        if (!workspaces[i] || splits[i] <= 0 || Cin[i] <= 0 || (Cin[i] % 8) || Ya[i] <= 0 || Yb[i] < 0 || !dwa[i] || (Yb[i] > 0 && !dwb[i])) return RS_ERR_ARG;
        b.e[i] = {(const float*)workspaces[i], (float*)dwa[i], (float*)dwb[i], splits[i], Ya[i] + Yb[i], Ya[i], Cin[i], 0u};
    }
    b.nstats = nstats;
    for (int i = 0; i < nstats; ++i) {
        if (!parts[i] || !souts[i] || sN[i] <= 0 || snblk[i] <= 0 || sC[i] <= 0 || scnt[i] <= 0.0 || (smode[i] != 0 && smode[i] != 1) || ssplit[i] < 0 || ssplit[i] >= sC[i])
            return RS_ERR_ARG;
        b.sj[i] = {(const float*)parts[i], (float*)souts[i], sN[i], snblk[i], sC[i], smode[i], ssplit[i], eps, scnt[i], 0u};
    }
    return rs_launch_wgrad_reduce_batch(b, ST(stream));
}

int rsuper_conv3_wgrad_s2_splits(int dtype, int Ca, int Mtot, int N, int FD, int FH, int FW) {
    if (!dt_ok(dtype) || Ca <= 0 || Mtot <= 0 || N <= 0 || FD <= 0 || FH <= 0 || FW <= 0) return -1;
    return rs_wgrad_s2_splits(dtype, Ca, Mtot, N, FD, FH, FW);
}
int rsuper_conv3_wgrad_s2(int dtype, const void* xa, int lda, int Ca, const float* mra,
                          const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                          float* dwa, float* dwb, float* workspace, int N, int FD, int FH, int FW, int splits, void* stream) {
    if (!dt_ok(dtype) || !xa || !ya || !dwa || !workspace || !ch_ok(Ca, lda) || Ca == 0 || !ch_ok(Ya, ldya) || Ya == 0) return RS_ERR_ARG;
    if (Yb > 0 && (!yb || !dwb || !ch_ok(Yb, ldyb))) return RS_ERR_ARG;
    if (N <= 0 || FD <= 0 || FH <= 0 || FW <= 0 || splits <= 0) return RS_ERR_ARG;
    if ((unsigned long long)N * FD * FH * FW * (unsigned long long)lda * (dtype == RS_F32 ? 4 : 2) >= (1ull << 32)) return RS_ERR_UNSUPPORTED;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.xa = {xa, lda, Ca, mra};
    p.ya = {ya, ldya, Ya, nullptr};
    p.yb = {yb, ldyb, Yb > 0 ? Yb : 0, nullptr};
    p.dwa = dwa; p.dwb = dwb; p.ws = workspace; p.N = N; p.D = FD; p.H = FH; p.W = FW; p.splits = splits;
    return rs_launch_wgrad_s2(p, dtype, ST(stream));
}

size_t rsuper_pointwise_packed_bytes(int dtype, int K, int N) { return dt_ok(dtype) && K > 0 && N > 0 ? rs_pw_packed_bytes(N, K, dtype) : 0; }
int rsuper_pointwise(int dtype, int mode, const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr,
                     float* y, int ldy, long R, int K, int N, void* packed, void* stream) {
    if (!dt_ok(dtype) || (mode != 0 && mode != 1) || !x || !y || !packed || R <= 0 || K <= 0 || N <= 0) return RS_ERR_ARG;   // w == nullptr: `packed` already holds the fragments
    if ((N % 4) || (ldx % 4) || (ldy % 4) || ldx < K || ldy < N || (mode == 1 && bias)) return RS_ERR_ARG;
    if (res && ((ldr % 4) || ldr < N)) return RS_ERR_ARG;
    if ((unsigned long long)R * ldx * 4ull >= (1ull << 32) || R > 0x7FFFFFFFl) return RS_ERR_UNSUPPORTED;   // buffer-addressed operand loads
    return rs_launch_pointwise(dtype, mode, x, ldx, w, bias, res, ldr, y, ldy, (int)R, K, N, packed, ST(stream));
}

int rsuper_pointwise_pack_batch(int dtype, const long long* table, int n, long total_items, void* arena, void* stream) {
    if (!dt_ok(dtype) || !table || n <= 0 || total_items <= 0 || !arena) return RS_ERR_ARG;
    return rs_launch_pointwise_pack_batch(dtype, table, n, total_items, arena, ST(stream));
}
int rsuper_pointwise_wgrad_splits(long R, int N, int K) { return (R > 0 && R <= 0x7FFFFFFFl && N > 0 && K > 0) ? rs_pw_wgrad_splits((int)R, N, K) : 0; }
int rsuper_pointwise_wgrad(int dtype, const float* dy, int ldy, const float* x, int ldx, long R, int N, int K, float* workspace, int splits,
                           float* dw, float* db, void* stream) {
    if (!dt_ok(dtype) || !dy || !x || !workspace || !dw || R <= 0 || N <= 0 || K <= 0 || splits <= 0) return RS_ERR_ARG;
    if ((N % 4) || (K % 4) || ldy < N || ldx < K) return RS_ERR_ARG;
    if ((unsigned long long)(R + 1) * ldx * 4ull >= (1ull << 32) || (unsigned long long)(R + 1) * ldy * 4ull >= (1ull << 32)) return RS_ERR_UNSUPPORTED;
    return rs_launch_pointwise_wgrad(dtype, dy, ldy, x, ldx, (int)R, N, K, workspace, splits, dw, db, ST(stream));
}

int rsuper_stats_finalize(const float* part, int N, int nblk, int C, double cnt, float eps, int mode, int split, float* out, void* stream) {
    if (!part || !out || N <= 0 || nblk <= 0 || C <= 0 || cnt <= 0 || split < 0 || split >= C || (mode != 0 && mode != 1)) return RS_ERR_ARG;
    return rs_launch_stats_finalize(part, N, nblk, C, cnt, eps, mode, split, out, ST(stream));
}

int rsuper_in_bwd_finalize(int dtype, const void* g, int ldg, const void* x, int ldx, const float* mr, const float* gm,
                           const void* add1, int lda1, const void* add2, int lda2, void* out, int ldo,
                           int N, int vox, int C, void* stream) {
    if (!dt_ok(dtype) || !g || !x || !mr || !gm || !out || !ch_ok(C, ldg) || !ch_ok(C, ldx) || !ch_ok(C, ldo)) return RS_ERR_ARG;
    InBwdParams p = {g, ldg, x, ldx, mr, gm, add1, lda1, add2, lda2, out, ldo, N, vox, C};
    return rs_launch_in_bwd(p, dtype, ST(stream));
}

int rsuper_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                        int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !y || !ch_ok(C, ldx) || !ch_ok(C, ldy) || blocks <= 0 || D < 2 || H < 2 || W < 2) return RS_ERR_ARG;
    PoolParams p = {x, ldx, y, ldy, nullptr, 0, part, N, D, H, W, C};
    return rs_launch_pool(p, dtype, 0, blocks, ST(stream));
}
int rsuper_maxpool2_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx,
                        int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !dy || !dx || !ch_ok(C, ldx) || !ch_ok(C, lddy) || !ch_ok(C, lddx)) return RS_ERR_ARG;
    if ((D & 1) || (H & 1) || (W & 1)) {
        // MaxPool3d(2) floors odd sizes: the trailing plane/row/column is never pooled and receives zero gradient
        if (lddx != C) return RS_ERR_UNSUPPORTED;
        if (const int rc = rs_launch_zero_bytes(dx, (size_t)N * D * H * W * C * (dtype == RS_F32 ? 4 : 2), ST(stream))) return rc;   // a kernel, not a memset node
    }
    PoolParams p = {x, ldx, (void*)dy, lddy, dx, lddx, nullptr, N, D, H, W, C};
    return rs_launch_pool(p, dtype, 1, 1, ST(stream));
}
int rsuper_maxpool2_bwd_add(int dtype, const void* x, int ldx, const void* dy, int lddy, const void* add, int lda, void* dx, int lddx,
                            int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !dy || !add || !dx || !ch_ok(C, ldx) || !ch_ok(C, lddy) || !ch_ok(C, lda) || !ch_ok(C, lddx)) return RS_ERR_ARG;
    if ((D & 1) || (H & 1) || (W & 1)) return RS_ERR_UNSUPPORTED;   // the never-pooled trailing planes would need a copy of `add`: the caller adds the two gradients itself
    PoolParams p = {x, ldx, (void*)dy, lddy, dx, lddx, nullptr, N, D, H, W, C, add, lda};
    return rs_launch_pool(p, dtype, 1, 1, ST(stream));
}

int rsuper_subsample2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                          int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !y || !ch_ok(C, ldx) || !ch_ok(C, ldy) || blocks <= 0 || D < 1 || H < 1 || W < 1 || N <= 0) return RS_ERR_ARG;
    PoolParams p = {x, ldx, y, ldy, nullptr, 0, part, N, D, H, W, C};
    return rs_launch_subsample(p, dtype, 0, blocks, ST(stream));
}
int rsuper_subsample2_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !dy || !dx || !ch_ok(C, lddy) || !ch_ok(C, lddx) || D < 1 || H < 1 || W < 1 || N <= 0) return RS_ERR_ARG;
    PoolParams p = {nullptr, 0, (void*)dy, lddy, dx, lddx, nullptr, N, D, H, W, C};
    return rs_launch_subsample(p, dtype, 1, 1, ST(stream));
}

int rsuper_upsample_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                        int N, int ID, int IH, int IW, int OD, int OH, int OW, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !y || !ch_ok(C, ldx) || !ch_ok(C, ldy) || blocks <= 0) return RS_ERR_ARG;
    UpParams p = {x, ldx, y, ldy, nullptr, 0, part, N, ID, IH, IW, OD, OH, OW, C};
    return rs_launch_upsample(p, dtype, 0, blocks, ST(stream));
}
int rsuper_upsample_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx,
                        int N, int ID, int IH, int IW, int OD, int OH, int OW, int C, void* stream) {
    if (!dt_ok(dtype) || !dy || !dx || !ch_ok(C, lddy) || !ch_ok(C, lddx)) return RS_ERR_ARG;
    UpParams p = {nullptr, 0, (void*)dy, lddy, dx, lddx, nullptr, N, ID, IH, IW, OD, OH, OW, C};
    return rs_launch_upsample(p, dtype, 1, 1, ST(stream));
}
int rsuper_stem_fwd(int dtype, const float* x, const float* w, void* y, int ldy, float* part,
                    int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !w || !y || !part || !ch_ok(C, ldy)) return RS_ERR_ARG;
    StemParams p = {x, w, y, ldy, part, nullptr, N, D, H, W, C};
    return rs_launch_stem(p, dtype, 0, ST(stream));
}
int rsuper_stem_wgrad(int dtype, const float* x, const void* dy, int lddy, float* dw, int N, int D, int H, int W, int C, void* stream) {
    if (!dt_ok(dtype) || !x || !dy || !dw || !ch_ok(C, lddy)) return RS_ERR_ARG;
    StemParams p = {x, nullptr, (void*)dy, lddy, nullptr, dw, N, D, H, W, C};
    return rs_launch_stem(p, dtype, 1, ST(stream));
}

int rsuper_head_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* logits, int N, int vox, int C, int K, void* stream) {
    if (!dt_ok(dtype) || !x || !w || !b || !logits || !ch_ok(C, ldx) || K <= 0) return RS_ERR_ARG;
    HeadParams p = {x, ldx, w, b, logits, nullptr, 0, nullptr, nullptr, N, vox, C, K};
    return rs_launch_head(p, dtype, 0, ST(stream));
}
int rsuper_head_bwd_data(int dtype, const float* dlogits, const float* w, void* dx, int lddx, int N, int vox, int C, int K, void* stream) {
    if (!dt_ok(dtype) || !dlogits || !w || !dx || !ch_ok(C, lddx) || K <= 0) return RS_ERR_ARG;
    HeadParams p = {nullptr, 0, w, nullptr, (float*)dlogits, dx, lddx, nullptr, nullptr, N, vox, C, K};
    return rs_launch_head(p, dtype, 1, ST(stream));
}
int rsuper_head_bwd_weight(int dtype, const void* x, int ldx, const float* dlogits, float* dw, float* db, int N, int vox, int C, int K, void* stream) {
    if (!dt_ok(dtype) || !x || !dlogits || !dw || !db || !ch_ok(C, ldx) || K <= 0) return RS_ERR_ARG;
    HeadParams p = {x, ldx, nullptr, nullptr, (float*)dlogits, nullptr, 0, dw, db, N, vox, C, K};
    return rs_launch_head(p, dtype, 2, ST(stream));
}

int rsuper_head_bwd(int dtype, const void* x, int ldx, const float* dlogits, const float* w, void* dx, int lddx, float* dw, float* db,
                    int N, int vox, int C, int K, void* stream) {
    if (!dt_ok(dtype) || !x || !dlogits || !w || !dx || !dw || !db || !ch_ok(C, ldx) || !ch_ok(C, lddx) || K <= 0) return RS_ERR_ARG;
    if (dtype == RS_BF16 && K <= 32 && C <= 32) {                  // one launch: weight + bias gradient and the data gradient from one pass over dlogits
        HeadParams p = {x, ldx, w, nullptr, (float*)dlogits, dx, lddx, dw, db, N, vox, C, K};
        return rs_launch_head(p, dtype, 2, ST(stream));
    }
    const int rc = rsuper_head_bwd_data(dtype, dlogits, w, dx, lddx, N, vox, C, K, stream);
    return rc ? rc : rsuper_head_bwd_weight(dtype, x, ldx, dlogits, dw, db, N, vox, C, K, stream);
}

int rsuper_plane_partials_fwd(const float* x, size_t xstride, const uint8_t* t, const uint8_t* k, const float* w1, const uint8_t* w2,
                              double* sums, int flags, int planes, size_t V, void* stream) {
    if (!x || !sums || planes <= 0 || V == 0 || (flags & ~2)) return RS_ERR_ARG;
    PlaneParams p = {x, xstride, t, k, w1, w2, sums, nullptr, nullptr, 0, V, k ? (flags >> 1) & 1 : 0, nullptr, 0, 0, nullptr};
    return rs_launch_plane_partials(p, planes, 0, ST(stream));
}
int rsuper_plane_partials_fwd2(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, double* sums, int flags, int planes, size_t V, void* stream) {
    if (!x || !sums || planes <= 0 || V == 0 || (flags & ~2) || (t && tpk)) return RS_ERR_ARG;
    if (tpk && (tC <= 0 || tP * 8 < tC || planes % tC)) return RS_ERR_ARG;
    PlaneParams p = {x, xstride, t, k, w1, w2, sums, nullptr, nullptr, 0, V, k ? (flags >> 1) & 1 : 0, tpk, tP, tC, kflags, nullptr};
    return rs_launch_plane_partials(p, planes, 0, ST(stream));
}
int rsuper_plane_partials_blocks(size_t V) { return V ? rs_plane_partials_blocks(V) : 0; }
int rsuper_plane_partials_fwd3(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, double* pblk, int flags, int planes, size_t V, void* stream) {
    if (!x || !pblk || planes <= 0 || V == 0 || (flags & ~2) || (t && tpk)) return RS_ERR_ARG;
    if (tpk && (tC <= 0 || tP * 8 < tC || planes % tC)) return RS_ERR_ARG;
    PlaneParams p = {x, xstride, t, k, w1, w2, nullptr, nullptr, nullptr, 0, V, k ? (flags >> 1) & 1 : 0, tpk, tP, tC, kflags, pblk};
    return rs_launch_plane_partials(p, planes, 0, ST(stream));
}
int rsuper_plane_sums_reduce(const double* pblk, int rows, int nb, float* out, void* stream) {
    if (!pblk || !out || rows <= 0 || nb <= 0) return RS_ERR_ARG;
    return rs_launch_plane_sums_reduce(pblk, rows, nb, out, ST(stream));
}
int rsuper_plane_partials_bwd(const float* x, size_t xstride, const uint8_t* t, const uint8_t* k, const float* w1, const uint8_t* w2,
                              const float* g, float* dx, int flags, int planes, size_t V, void* stream) {
    if (!x || !g || !dx || planes <= 0 || V == 0 || (flags & ~3)) return RS_ERR_ARG;
    PlaneParams p = {x, xstride, t, k, w1, w2, nullptr, g, dx, flags & 1, V, k ? (flags >> 1) & 1 : 0, nullptr, 0, 0, nullptr, nullptr};
    return rs_launch_plane_partials(p, planes, 1, ST(stream));
}
int rsuper_plane_partials_bwd2(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, const float* g, float* dx, int flags, int planes, size_t V, void* stream) {
    if (!x || !g || !dx || planes <= 0 || V == 0 || (flags & ~3) || (t && tpk)) return RS_ERR_ARG;
    if (tpk && (tC <= 0 || tP * 8 < tC || planes % tC)) return RS_ERR_ARG;
    PlaneParams p = {x, xstride, t, k, w1, w2, nullptr, g, dx, flags & 1, V, k ? (flags >> 1) & 1 : 0, tpk, tP, tC, kflags};
    return rs_launch_plane_partials(p, planes, 1, ST(stream));
}
int rsuper_cnorm_rows(long vox) { return rs_cnorm_rows(vox); }
int rsuper_cnorm_stats(const float* x, const float* dy, const float* mr, float* part, int N, long vox, int C, int relu, int mode, void* stream) {
    if (!x || !part || N <= 0 || vox <= 0 || C <= 0 || (C & 3) || (mode != 0 && mode != 1) || (mode == 1 && (!dy || !mr))) return RS_ERR_ARG;
    return rs_launch_cnorm_stats(x, dy, mr, part, N, vox, C, relu ? 1 : 0, mode, ST(stream));
}
int rsuper_cnorm_apply(const float* x, const float* dy, const float* mr, const float* gm, float* out, int N, long vox, int C, int relu, int mode,
                       void* stream) {
    if (!x || !mr || !out || N <= 0 || vox <= 0 || C <= 0 || (C & 3) || mode < 0 || mode > 2 || (mode == 1 && (!dy || !gm))) return RS_ERR_ARG;
    return rs_launch_cnorm_apply(x, dy, mr, gm, out, N, vox, C, relu ? 1 : 0, mode, ST(stream));
}
/* statistics -> finalize -> apply as ONE host call (three launches): the eager training step of MedFormer is bound by the host's
 * per-call cost (230 norm calls per step), not by these kernels */
int rsuper_cnorm_forward(const float* x, float* part, float* mr, float* y, int N, long vox, int C, int relu, float eps, void* stream) {
    if (!x || !part || !mr || !y || N <= 0 || vox <= 0 || C <= 0 || (C & 3)) return RS_ERR_ARG;
    const int rows = rs_cnorm_rows(vox);
    int rc = rs_launch_cnorm_stats(x, nullptr, nullptr, part, N, vox, C, relu ? 1 : 0, 0, ST(stream));
    if (rc != RS_OK) return rc;
    rc = rs_launch_stats_finalize(part, N, rows, C, (double)vox, eps, 0, 0, mr, ST(stream));
    if (rc != RS_OK) return rc;
    return rs_launch_cnorm_apply(x, nullptr, mr, nullptr, y, N, vox, C, relu ? 1 : 0, 0, ST(stream));
}
int rsuper_cnorm_backward(const float* x, const float* dy, const float* mr, float* part, float* gm, float* dx, int N, long vox, int C, int relu,
                          void* stream) {
    if (!x || !dy || !mr || !part || !gm || !dx || N <= 0 || vox <= 0 || C <= 0 || (C & 3)) return RS_ERR_ARG;
    const int rows = rs_cnorm_rows(vox);
    int rc = rs_launch_cnorm_stats(x, dy, mr, part, N, vox, C, relu ? 1 : 0, 1, ST(stream));
    if (rc != RS_OK) return rc;
    rc = rs_launch_stats_finalize(part, N, rows, C, (double)vox, 0.f, 1, 0, gm, ST(stream));
    if (rc != RS_OK) return rc;
    return rs_launch_cnorm_apply(x, dy, mr, gm, dx, N, vox, C, relu ? 1 : 0, 1, ST(stream));
}
int rsuper_cnorm_small(const float* x, const float* dy, const float* mr, float* out, float* mr_out, int N, long vox, int C, int relu, float eps,
                       int mode, void* stream) {
    if (!x || !out || N <= 0 || vox <= 0 || C <= 0 || (C & 3) || (mode != 0 && mode != 1) || (mode == 0 && !mr_out) || (mode == 1 && (!dy || !mr)))
        return RS_ERR_ARG;
    return rs_launch_cnorm_small(x, dy, mr, out, mr_out, N, vox, C, relu ? 1 : 0, eps, mode, ST(stream));
}
int rsuper_token_attn_supported(int L, int dim_head) { return rs_token_attn_supported(L, dim_head); }
int rsuper_token_attn_fwd(const float* qkv, float* o, float* p, int B, int L, int heads, int dim_head, float scale, void* stream) {
    if (!qkv || !o || !p || B <= 0 || L <= 0 || heads <= 0 || dim_head <= 0) return RS_ERR_ARG;
    return rs_launch_token_attn(qkv, o, p, nullptr, nullptr, B, L, heads, dim_head, scale, ST(stream));
}
int rsuper_token_attn_bwd(const float* qkv, const float* o, const float* p, const float* d_o, float* d_qkv, int B, int L, int heads, int dim_head, float scale,
                          void* stream) {
    if (!qkv || !o || !p || !d_o || !d_qkv || B <= 0 || L <= 0 || heads <= 0 || dim_head <= 0) return RS_ERR_ARG;
    return rs_launch_token_attn(qkv, (float*)o, (float*)p, d_o, d_qkv, B, L, heads, dim_head, scale, ST(stream));
}
int rsuper_battn_supported(int T, int dim_head, int heads) { return rs_battn_supported(T, dim_head, heads); }
int rsuper_battn_chunks(int L, int heads) { return rs_battn_chunks(L, heads); }
int rsuper_battn_fwd(const float* fqv, const float* mqv, float* f_out, float* m_out, float* lse, float* part, float* pms, int B, int L, int T,
                     int heads, int dim_head, float scale, void* stream) {
    if (!fqv || !mqv || !f_out || !m_out || !lse || !part || !pms || B <= 0 || L <= 0 || T <= 0 || heads <= 0 || dim_head <= 0) return RS_ERR_ARG;
    return rs_launch_battn(fqv, mqv, f_out, m_out, lse, nullptr, nullptr, nullptr, nullptr, part, pms, B, L, T, heads, dim_head, scale, 0, ST(stream));
}
int rsuper_battn_bwd(const float* fqv, const float* mqv, const float* m_out, const float* lse, const float* d_f_out, const float* d_m_out,
                     float* d_fqv, float* d_mqv, float* part, int B, int L, int T, int heads, int dim_head, float scale, void* stream) {
    if (!fqv || !mqv || !m_out || !lse || !d_f_out || !d_m_out || !d_fqv || !d_mqv || !part || B <= 0 || L <= 0 || T <= 0 || heads <= 0 || dim_head <= 0)
        return RS_ERR_ARG;
    return rs_launch_battn(fqv, mqv, nullptr, (float*)m_out, (float*)lse, d_f_out, d_m_out, d_fqv, d_mqv, part, nullptr, B, L, T, heads, dim_head,
                           scale, 1, ST(stream));
}
int rsuper_se_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* part, float* ms, float* tab,
                      float* hbuf, float* y, int N, long vox, int C, int r, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !part || !ms || !tab || !hbuf || !y || N <= 0 || vox <= 0 || C <= 0 || (C & 3) || r <= 0) return RS_ERR_ARG;
    return rs_launch_se_forward(x, w1, b1, w2, b2, part, ms, tab, hbuf, y, N, vox, C, r, ST(stream));
}
int rsuper_se_backward(const float* x, const float* dy, const float* ident, const float* w1, const float* w2, const float* ms, const float* tab,
                       const float* hbuf, float* part, float* gm, float* dz1, float* tab2, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                       int N, long vox, int C, int r, void* stream) {
    if (!x || !dy || !ident || !w1 || !w2 || !ms || !tab || !hbuf || !part || !gm || !dz1 || !tab2 || !dx || !dw1 || !db1 || !dw2 || !db2 ||
        N <= 0 || vox <= 0 || C <= 0 || (C & 3) || r <= 0)
        return RS_ERR_ARG;
    return rs_launch_se_backward(x, dy, ident, w1, w2, ms, tab, hbuf, part, gm, dz1, tab2, dx, dw1, db1, dw2, db2, N, vox, C, r, ST(stream));
}
int rsuper_cl_planar(const float* src, float* dst, int N, long vox, int C, int K, int to_channels_last, void* stream) {
    if (!src || !dst || N <= 0 || vox <= 0 || C <= 0 || K <= 0) return RS_ERR_ARG;
    return rs_launch_cl_planar(src, dst, N, vox, C, K, to_channels_last ? 1 : 0, ST(stream));
}
int rsuper_depthwise3_rows(long vox) { return rs_depthwise_rows(vox); }
int rsuper_depthwise3_fwd(const float* x, const float* w, float* y, int N, int D, int H, int W, int C, int flip, void* stream) {
    if (!x || !w || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (flip != 0 && flip != 1)) return RS_ERR_ARG;
    return rs_launch_depthwise(x, w, y, N, D, H, W, C, flip, ST(stream));
}
int rsuper_depthwise3_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int D, int H, int W, int C, void* stream) {
    if (!x || !dy || !part || !dw || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return RS_ERR_ARG;
    return rs_launch_depthwise_wgrad(x, dy, part, dw, N, D, H, W, C, ST(stream));
}
int rsuper_seg_from_sums(const float* sums, const float* cw, int B, int C, size_t V, double scale, float* loss, float* dsums, void* stream) {
    if (!sums || !loss || !dsums || B <= 0 || C <= 0 || V == 0) return RS_ERR_ARG;
    SegSumsParams p = {sums, cw, B, C, 1.0 / ((double)B * C * (double)V), scale, loss, dsums};
    return rs_launch_seg_from_sums(p, ST(stream));
}
int rsuper_report_from_sums(const float* sums, const float* roww, int R, int B, int L, size_t V, int use_vol, const float* flags, const float* rvol,
                            double tol, double E, int nplans, const int* plan, int apply_dice, int standard_ce, float* loss, float* jac, void* stream) {
    if (!sums || !roww || R <= 0 || B <= 0 || L <= 0 || V == 0 || !loss || !jac || nplans < 0 || (nplans > 0 && !plan)) return RS_ERR_ARG;
    if (use_vol && (!flags || !rvol || R < L * B)) return RS_ERR_ARG;
    ReportSumsParams p = {sums, roww, B, L, (double)V, use_vol, flags, rvol, tol, E, nplans, plan, apply_dice, standard_ce, loss, jac};
    return rs_launch_report_from_sums(p, R, ST(stream));
}
int rsuper_sigmoid_mask(const float* x, const uint8_t* m, float* out, size_t V, void* stream) {
    if (!x || !out || V == 0) return RS_ERR_ARG;
    return rs_launch_sigmoid_mask(x, m, out, V, ST(stream));
}

int rsuper_window_accumulate(const float* logits, float* acc, int BK, int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0,
                             int assign, void* stream) {
    if (!logits || !acc || BK <= 0 || wd <= 0 || wh <= 0 || ww <= 0) return RS_ERR_ARG;
    if (d0 < 0 || h0 < 0 || w0 < 0 || d0 + wd > D || h0 + wh > H || w0 + ww > W) return RS_ERR_ARG;
    return rs_launch_window_accumulate(logits, acc, BK, wd, wh, ww, D, H, W, d0, h0, w0, assign, ST(stream));
}
int rsuper_window_normalize(float* acc, const float* cd, const float* ch, const float* cw, long BK, int D, int H, int W, void* stream) {
    if (!acc || !cd || !ch || !cw || BK <= 0 || D <= 0 || H <= 0 || W <= 0) return RS_ERR_ARG;
    return rs_launch_window_normalize(acc, cd, ch, cw, BK, D, H, W, ST(stream));
}

int rsuper_dilate_volume(const uint8_t* in, uint8_t* out, uint8_t* tmp, long nvol, int D, int H, int W, int kernel_size, void* stream) {
    return rsuper_dilate_volume_sparse(in, out, tmp, nullptr, nvol, D, H, W, kernel_size, stream);
}
int rsuper_dilate_volume_sparse(const uint8_t* in, uint8_t* out, uint8_t* tmp, const uint8_t* flags, long nvol, int D, int H, int W, int kernel_size,
                                void* stream) {
    if (!in || !out || nvol <= 0 || kernel_size < 1) return RS_ERR_ARG;
    if (kernel_size % 2 == 0) kernel_size += 1;                  // losses_foundation.py:24-25
    const int full = 3;
    if (kernel_size <= 2 * full + 1) return rs_launch_dilate_pass(in, out, flags, nvol, D, H, W, kernel_size, ST(stream));
    if (!tmp) return RS_ERR_ARG;
    const int radius = (kernel_size - 1) / 2, num_full = radius / full, rem = radius % full;   // :31-44
    const int passes = num_full + (rem > 0 ? 1 : 0);
    const uint8_t* src = in;
    for (int i = 0; i < passes; ++i) {
        uint8_t* dst = ((passes - i) & 1) ? out : tmp;           // last pass lands in `out`
        const int k = i < num_full ? 2 * full + 1 : 2 * rem + 1;
        const int rc = rs_launch_dilate_pass(src, dst, flags, nvol, D, H, W, k, ST(stream));
        if (rc) return rc;
        src = dst;
    }
    return RS_OK;
}

long rsuper_ball_workspace_floats(int D, int H, int W, int d_odd) { return rs_ball_workspace_floats(D, H, W, d_odd); }
int rsuper_ball_conv_argmax(const float* x, int D, int H, int W, int d_odd, float std, unsigned long long* best, float* conv_out, float* workspace,
                            void* stream) {
    if (!x || !best || d_odd < 1 || !(d_odd & 1) || std <= 0) return RS_ERR_ARG;
    return rs_launch_ball_conv_argmax(x, D, H, W, d_odd, std, best, conv_out, workspace, ST(stream));
}
int rsuper_insert_ball(uint8_t* out, int D, int H, int W, int cz, int cy, int cx, int d_odd, int half, unsigned int* count, void* stream) {
    if (!out || !count) return RS_ERR_ARG;
    return rs_launch_insert_ball(out, D, H, W, cz, cy, cx, d_odd, half, count, ST(stream));
}
int rsuper_insert_ball_at(uint8_t* out, int D, int H, int W, const unsigned long long* best, int d_odd, int half, unsigned int* count, void* stream) {
    if (!out || !count || !best) return RS_ERR_ARG;
    return rs_launch_insert_ball_at(out, D, H, W, best, d_odd, half, count, ST(stream));
}
int rsuper_radix_hist(const float* x, const uint8_t* m, long V, uint32_t prefix, int shift, unsigned int* hist256, void* stream) {
    if (!x || !hist256 || (shift != 24 && shift != 16 && shift != 8 && shift != 0)) return RS_ERR_ARG;
    return rs_launch_radix_hist(x, m, V, prefix, shift, hist256, ST(stream));
}
int rsuper_topk_mark(const float* x, const uint8_t* m, long V, uint32_t thr_bits, unsigned int need_eq, uint8_t* out, void* stream) {
    if (!x || !out) return RS_ERR_ARG;
    return rs_launch_topk_mark(x, m, V, thr_bits, need_eq, out, ST(stream));
}
int rsuper_topk_select(const float* x, const uint8_t* m, long V, unsigned int k, uint8_t* out, unsigned int* workspace, void* stream) {
    if (!x || !out || !workspace || V <= 0 || k == 0 || (long)k > V) return RS_ERR_ARG;
    return rs_launch_topk_select(x, m, V, &k, 1, out, workspace, 0, ST(stream));
}
int rsuper_topk_select_multi(const float* x, const uint8_t* m, long V, const unsigned int* k, int nk, uint8_t* out, unsigned int* workspace,
                             int clip_to_mask, void* stream) {
    if (!x || !out || !workspace || !k || V <= 0 || nk < 1 || nk > 4) return RS_ERR_ARG;
    for (int i = 0; i < nk; ++i)
        if (k[i] == 0 || (long)k[i] > V) return RS_ERR_ARG;
    return rs_launch_topk_select(x, m, V, k, nk, out, workspace, clip_to_mask ? 1 : 0, ST(stream));
}
int rsuper_compact(const float* x, const uint8_t* pm, long V, float* vals, uint32_t* idx, unsigned int* n, void* stream) {
    if (!x || !pm || !vals || !idx || !n) return RS_ERR_ARG;
    return rs_launch_compact(x, pm, V, vals, idx, n, ST(stream));
}
int rsuper_rank_weights(const float* vals, const uint32_t* idx, unsigned int n, float log2_d, float scale, float* w, void* stream) {
    if (!vals || !idx || !w) return RS_ERR_ARG;
    return rs_launch_rank_weights(vals, idx, n, log2_d, scale, w, ST(stream));
}
int rsuper_rank_assign(const long long* ids, unsigned int n, float log2_d, float scale, float* w, void* stream) {
    if (!ids || !w) return RS_ERR_ARG;
    return rs_launch_rank_assign(ids, n, log2_d, scale, w, ST(stream));
}
int rsuper_unpack_bits(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, void* stream) {
    if (!packed || !out || B <= 0 || P <= 0 || C <= 0 || C > 8 * P || V <= 0) return RS_ERR_ARG;
    return rs_launch_unpack_bits(packed, out, B, P, C, V, ST(stream));
}
// ---- timing events of the roofline pass (bench.py, ops.KernelTimer).  hipEventDisableSystemFence: an event of the default kind performs a system-scope release when
// it is recorded -- cache write-back + invalidate between every pair of launches it brackets, which both costs time inside the bracket and makes the NEXT kernel start
// on a cold L2 (packed weight fragments, halo rows shared with the previous launch).  hip_runtime_api.h: "can be used for events that are only being used to measure
// timing ... can improve the accuracy of timing measurements by avoiding the cost of cache writeback and invalidation".
int rsuper_timer_event_create(void** ev) {
    if (!ev) return RS_ERR_ARG;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return RS_ERR_LAUNCH;
    *ev = (void*)e;
    return RS_OK;
}
int rsuper_timer_event_record(void* ev, void* stream) {
    if (!ev) return RS_ERR_ARG;
    return hipEventRecord((hipEvent_t)ev, ST(stream)) == hipSuccess ? RS_OK : RS_ERR_LAUNCH;
}
int rsuper_timer_event_elapsed_ms(void* a, void* b, float* ms) {
    if (!a || !b || !ms) return RS_ERR_ARG;
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return RS_ERR_LAUNCH;
    return hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b) == hipSuccess ? RS_OK : RS_ERR_LAUNCH;
}
int rsuper_timer_event_destroy(void* ev) {
    if (!ev) return RS_ERR_ARG;
    return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? RS_OK : RS_ERR_LAUNCH;
}
int rsuper_unpack_bits_sel(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, const uint8_t* flags, const uint8_t* force, void* stream) {
    if (!packed || !out || B <= 0 || P <= 0 || C <= 0 || C > 8 * P || V <= 0) return RS_ERR_ARG;
    return rs_launch_unpack_bits_sel(packed, out, B, P, C, V, flags, force, ST(stream));
}
int rsuper_plane_any_bits(const uint8_t* packed, int B, int P, int C, long V, uint8_t* flags, void* stream) {
    if (!packed || !flags || B <= 0 || P <= 0 || C <= 0 || C > 8 * P || V <= 0 || ((uintptr_t)packed & 15) || (V & 15)) return RS_ERR_ARG;
    return rs_launch_plane_any_bits(packed, B, P, C, V, flags, ST(stream));
}
int rsuper_guard_consistency(const uint8_t* m_any, const uint8_t* u_any, const float* volumes, int B, int T, int* flags, void* stream) {
    if (!m_any || !u_any || !volumes || !flags || B <= 0 || T <= 0) return RS_ERR_ARG;
    return rs_launch_guard_consistency(m_any, u_any, volumes, B, T, flags, ST(stream));
}
int rsuper_guard_range(const float* x, size_t n, float lo, float hi, int* flags, void* stream) {
    if (!x || !flags || ((uintptr_t)x & 3) || (n >= 4 && ((uintptr_t)x & 15))) return RS_ERR_ARG;
    return rs_launch_guard_range(x, n, lo, hi, flags, ST(stream));
}
int rsuper_plane_any(const uint8_t* m, long planes, long V, uint8_t* flags, void* stream) {
    if (!m || !flags || planes <= 0 || V <= 0 || ((uintptr_t)m & 15) || (V & 15 && planes > 1)) return RS_ERR_ARG;
    return rs_launch_plane_any(m, planes, V, flags, ST(stream));
}
int rsuper_mask_op(uint8_t* a, const uint8_t* b, long V, int op, void* stream) {
    if (!a || !b || op < 0 || op > 2) return RS_ERR_ARG;
    return rs_launch_mask_op(a, b, V, op, ST(stream));
}
int rsuper_zero_where(float* x, const uint8_t* m, long V, void* stream) {
    if (!x || !m) return RS_ERR_ARG;
    return rs_launch_zero_where(x, m, V, ST(stream));
}
int rsuper_count(const uint8_t* m, long V, unsigned int* count, void* stream) {
    if (!m || !count) return RS_ERR_ARG;
    return rs_launch_count(m, V, count, ST(stream));
}

int rsuper_grad_sqnorm(int n, void* const* host_g, const size_t* host_numel, double* total_sq, void* stream) {
    if (n <= 0 || !host_g || !host_numel || !total_sq) return RS_ERR_ARG;
    // no hipMemsetAsync of the accumulator: the first chunk's reduce assigns it (a captured memset node wrote 0xC0 bytes after eager launches, optim.hip)
    int first = 1;
    const int rc = for_chunks(n, nullptr, host_g, nullptr, nullptr, nullptr, host_numel, [&](const MTChunk& c) {
        const int r = rs_launch_sqnorm(c, total_sq, first, ST(stream));
        first = 0;
        return r;
    });
    if (rc == RS_OK && first) return rs_launch_zero_bytes(total_sq, sizeof(double), ST(stream));   // no element at all: the norm is 0
    return rc;
}
int rsuper_clip_scale(int n, void* const* host_g, const size_t* host_numel, float max_norm, const double* total_sq, void* stream) {
    if (n <= 0 || !host_g || !host_numel || !total_sq) return RS_ERR_ARG;
    return for_chunks(n, nullptr, host_g, nullptr, nullptr, nullptr, host_numel, [&](const MTChunk& c) { return rs_launch_scale(c, max_norm, total_sq, ST(stream)); });
}
int rsuper_adamw_ema_step(int n, void* const* host_p, void* const* host_g, void* const* host_m, void* const* host_v,
                          void* const* host_ema, const size_t* host_numel, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float ema_alpha, float max_norm, const double* total_sq, void* stream) {
    if (n <= 0 || !host_p || !host_g || !host_m || !host_v || !host_numel || step < 1) return RS_ERR_ARG;
    AdamParams a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    a.sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.ema_alpha = ema_alpha; a.max_norm = max_norm; a.dyn = nullptr;
    return for_chunks(n, host_p, host_g, host_m, host_v, host_ema, host_numel, [&](const MTChunk& c) { return rs_launch_adamw_ema(c, a, total_sq, ST(stream)); });
}
int rsuper_adamw_ema_step_dyn(int n, void* const* host_p, void* const* host_g, void* const* host_m, void* const* host_v,
                              void* const* host_ema, const size_t* host_numel, float beta1, float beta2, float eps, float weight_decay,
                              float max_norm, const double* total_sq, const float* dyn, void* stream) {
    if (n <= 0 || !host_p || !host_g || !host_m || !host_v || !host_numel || !dyn) return RS_ERR_ARG;
    AdamParams a;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.step_size = 0.f; a.sqrt_bc2 = 1.f; a.ema_alpha = 0.f; a.max_norm = max_norm; a.dyn = dyn;
    return for_chunks(n, host_p, host_g, host_m, host_v, host_ema, host_numel, [&](const MTChunk& c) { return rs_launch_adamw_ema(c, a, total_sq, ST(stream)); });
}

}  // extern "C"
