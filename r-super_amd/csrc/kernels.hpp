// Internal launch interfaces shared between the kernel translation units and the C-ABI (api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// One channels-last activation source: element (n,d,h,w,c) at x[(((n*D+d)*H+h)*W+w)*ld + c].
struct ConvSrc {
    const void* x;     // base pointer (already offset to the first channel of the view)
    int ld;            // channel stride of a voxel (elements)
    int C;             // channels in this source (multiple of 8)
    const float* mr;   // [N][C][2] (mean, rstd) -> fused InstanceNorm+ReLU prologue; nullptr -> raw
};

struct IgemmParams {
    ConvSrc a, b;          // GEMM-K sources (b.C == 0 when unused)
    const void* wp;        // packed B fragments (rs_launch_pack)
    int ntiles;            // 32-column tiles in the packed weights (multiple of bn/32)
    int bn;                // block N tile: 32, 64 or 128 (96: depth-reuse kernel only)
    int N, D, H, W;
    int Cout;              // valid GEMM-N columns
    void* out; int ldo;
    const void* res; int ldr;   // EPI 0: optional residual added before store
    float* part;           // per-block partial sums [N][rows][Cout][2] or nullptr (rows = rs_igemm_part_rows)
    int pc;                // 1: producer/consumer persistent kernel (bf16); 2: weight-stationary kernel (bf16, bn 32); 3: depth-reuse kernel (bf16, bn 64 / 96 / 128)
    ConvSrc ea, eb;        // EPI 1: forward inputs (with mr) whose relu mask / x_n the data-gradient needs
    int box;               // > 0: volume-fitted K-split kernel (conv3d_igemm_box.hip), value = rs_box_config (3: one box per sample, reduction split over blocks)
    float* ws; int nsplit; // box == 3: f32 workspace [nsplit][N * D * H * W][Cout] and the number of chunk ranges
    int out_split; long long out_part;   // out_split > 0 (depth-reuse kernel only): columns >= out_split go to a second tensor `out_part` ELEMENTS behind `out`, both with row stride ldo
};

struct PackParams {
    const float* wa; const float* wb;
    int mode;              // 0 forward, 1 dgrad
    int ka, kb;            // GEMM-K channels per source
    int na, nb;            // GEMM-N columns from wa / wb (mode 0); mode 1: na = forward Cin (= row stride), nb = 0 (all columns) or first column << 16 | columns
    int ntiles;
};

struct WgradParams {
    ConvSrc xa, xb;        // forward inputs (x_hat recomputed from mr when non-null)
    ConvSrc ya, yb;        // output-gradient sources: rows [0,ya.C) -> dwa, [ya.C, ya.C+yb.C) -> dwb
    float* dwa; float* dwb;   // (Cout_a, Cin, 27), (Cout_b, Cin, 27) f32, overwritten
    float* ws;                // workspace: splits * 27 * (Ya+Yb) * Cin f32
    int N, D, H, W;
    int splits;            // spatial split factor (grid.z)
};

#define RS_PACK_BATCH_MAX 40
struct PackBatch {
    int n;
    PackParams q[RS_PACK_BATCH_MAX];
    unsigned long long vec_start[RS_PACK_BATCH_MAX + 1];   // prefix sum of 16-byte vectors; entry i writes at out + vec_start[i]*16 B
    unsigned blk_start[RS_PACK_BATCH_MAX + 1];             // prefix sum of blocks (filled by rs_launch_pack_batch)
};
int rs_launch_pack_batch(PackBatch& b, int dtype, void* out, hipStream_t st);
int rs_launch_igemm(const IgemmParams& p, int dtype, int epi, hipStream_t st);
int rs_igemm_part_rows(int bn, int pc, int tiles, int n_cols, int N);
// weight-stationary variant (conv3d_igemm_ws.hip): bf16, bn 32 / 64; same partial-row count as the producer/consumer kernel
bool rs_igemm_ws_supported(const IgemmParams& p, int dtype, int epi);
int rs_launch_igemm_ws(const IgemmParams& p, int epi, hipStream_t st);
// depth-reuse kernel for the wide full-resolution layers (conv3d_igemm_kd.hip): bf16, 64 / 96 / 128-column blocks, 4 x 8 x 16-voxel tiles
bool rs_igemm_kd_supported(const IgemmParams& p, int dtype);
int rs_igemm_kd_part_rows(int bn, int N, int D, int H, int W, int n_cols);
int rs_launch_igemm_kd(const IgemmParams& p, int epi, hipStream_t st);
// volume-fitted in-block K-split kernel for under-filled (low-resolution) launches (conv3d_igemm_box.hip): bf16, bn 64
int rs_box_config(int N, int D, int H, int W, int n_cols);
int rs_box_part_rows(int cfg, int D, int H, int W);
int rs_box_nsplit(int N, int n_cols, int nch);
int rs_launch_igemm_box(const IgemmParams& p, int cfg, int epi, hipStream_t st);
// stride-2 forward (mode 1) / data gradient (mode 2) by parity classes (conv3d_igemm_s2.hip): p.D/H/W = half-resolution grid, bn 64
int rs_launch_igemm_s2(const IgemmParams& p, int dtype, int mode, int FD, int FH, int FW, hipStream_t st);
// persistent strided forward (conv3d_igemm_s2k.hip): bf16, one normalised source; statistics rows = rs_igemm_s2k_part_rows per sample
bool rs_igemm_s2k_supported(const IgemmParams& p, int dtype, int FD, int FH, int FW);
int rs_igemm_s2k_part_rows(int ntiles, int n_cols, int N, int D, int H, int W);
int rs_launch_igemm_s2k(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st);
// persistent strided data gradient (conv3d_igemm_s2d.hip): bf16, raw dy sources; InstanceNorm-backward rows = rs_igemm_s2d_part_rows per sample
bool rs_igemm_s2d_supported(const IgemmParams& p, int dtype, int FD, int FH, int FW);
int rs_igemm_s2d_part_rows(int n_cols, int N, int D, int H, int W);
int rs_launch_igemm_s2d(const IgemmParams& p, int FD, int FH, int FW, hipStream_t st);
size_t rs_packed_elems(int dtype, int ka, int kb, int ntiles);
int rs_launch_pack(const PackParams& q, int dtype, void* out, hipStream_t st);
// softmax(q k^T * scale) v of a short token sequence (token_attn.hip); d_qkv == nullptr: forward (o, p written), else backward (p, d_o read)
int rs_token_attn_supported(int L, int Dh);
int rs_launch_token_attn(const float* qkv, float* o, float* p, const float* d_o, float* d_qkv, int B, int L, int H, int Dh, float scale, hipStream_t st);
int rs_launch_wgrad(const WgradParams& p, int dtype, int use_tr, hipStream_t st, bool reduce = true);
int rs_launch_wgrad_reduce(const WgradParams& p, hipStream_t st);
// slab reductions of several weight gradients in ONE launch (the per-layer reduce is a ~10 us launch at the dependent-launch floor, 34 per step)
#define RS_REDUCE_BATCH_MAX 48
#define RS_REDUCE_STATS_MAX 2
struct ReduceBatch {
    int n;
    struct Entry { const float* ws; float* dwa; float* dwb; int splits, Mtot, Ya, Cin; unsigned blk_start; } e[RS_REDUCE_BATCH_MAX];
    unsigned blocks;
    // statistics finalisations riding in the same launch (round 6): blocks [blocks, blocks + sum of the jobs' blocks) evaluate rs_launch_stats_finalize's
    // arithmetic for job j -- the data-gradient launch of a BasicBlock's conv1 leaves its InstanceNorm-backward rows, the block's weight gradients their slabs,
    // and ONE launch turns both into what in_bwd_finalize / the optimiser read (a dependent 5 us launch less per block and source)
    int nstats;
    struct Stats { const float* part; float* out; int N, nblk, C, mode, split; float eps; double cnt; unsigned blk_start; } sj[RS_REDUCE_STATS_MAX];
};
int rs_launch_wgrad_reduce_batch(ReduceBatch& b, hipStream_t st);
bool rs_wgrad2_mt1(int dtype, int Mtot, int tiles_total);
int rs_wgrad_splits(int dtype, int Mtot, int nch, int tiles_total);
// small-volume weight gradient (conv3d_wgrad_sv.hip): a depth slab of one sample whole in LDS, flat-voxel reduction; splits = 0: does not apply
int rs_wgrad_sv_splits(int dtype, int Mtot, int Ya, int nch, int N, int D, int H, int W);
int rs_launch_wgrad_sv(const WgradParams& p, hipStream_t st);
// second-generation weight gradient (conv3d_wgrad2.hip): bf16, operand re-use across taps + double-buffered tiles; same slabs, the caller reduces
bool rs_wgrad2_supported(const WgradParams& p, int dtype);
int rs_wgrad2_min_tiles(int t);     // tiles per block from which bf16 launches take it (t < 0: query)
int rs_launch_wgrad2(const WgradParams& p, hipStream_t st);
// stride-2 convolution (conv3d_wgrad_s2.hip): p.N/D/H/W = the FULL-resolution grid of x, dY lives on the ((D+1)/2, (H+1)/2, (W+1)/2) grid; xb unused
int rs_wgrad_s2_splits(int dtype, int Ca, int Mtot, int N, int D, int H, int W);
int rs_launch_wgrad_s2(const WgradParams& p, int dtype, hipStream_t st);

// 1x1x1 convolution / linear layer as an MFMA GEMM on f32 channels-last rows (pointwise.hip); packed = workspace of rs_pw_packed_bytes
size_t rs_pw_packed_bytes(int N, int K, int dtype);
int rs_launch_pointwise(int dtype, int mode, const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr,
                        float* y, int ldy, int R, int K, int N, void* packed, hipStream_t st);
// weight (+ bias) gradient of the same layer: dW = dy^T x, db = column sums of dy; part = workspace of S * (N*K + N) floats, S = rs_pw_wgrad_splits
int rs_pw_wgrad_splits(int R, int N, int K);
int rs_launch_pointwise_pack_batch(int dtype, const long long* table, int n, long total_items, void* arena, hipStream_t st);
int rs_launch_pointwise_wgrad(int dtype, const float* dy, int ldy, const float* x, int ldx, int R, int N, int K, float* part, int S,
                              float* dw, float* db, hipStream_t st);
