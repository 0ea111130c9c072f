#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// One launch handles `planes` (sample, class) planes of V voxels.
struct PlaneParams {
    const float* x; size_t xstride;   // logits plane p at x + p*xstride (NCDHW f32)
    const uint8_t* t;                 // target 0/1 [planes][V] or nullptr (=0)
    const uint8_t* k;                 // known / penalise mask [planes][V] or nullptr (=1); with kinv: the UNKNOWN mask (weight = !k)
    const float* w1;                  // foreground weights [planes][V] or nullptr
    const uint8_t* w2;                // dilated pseudo mask [planes][V] or nullptr; background weight = 1 - w2
    double* sums;                     // forward: [planes][6] f64, pre-zeroed
    const float* g;                   // backward: [planes][6] d(loss)/d(sums)
    float* dx;                        // backward: d logits, plane p at dx + p*xstride
    int accumulate;                   // backward: dx += instead of dx =
    size_t V;
    int kinv;
    const uint8_t* tpk; int tP, tC;   // bit-packed target [samples][tP][V] for planes = samples x tC classes (replaces t), or nullptr
    const uint8_t* kflags;            // with kinv: [planes] any-flags of the UNdilated unknown map; 0 = the plane of k is all zero and is not read (nullptr: read every plane)
    double* pblk;                     // forward, round 6: [planes][gridDim.x][6] per-block sums written with plain stores (no atomics, no pre-zeroing); nullptr: atomics into `sums`
};
int rs_plane_partials_blocks(size_t V);
int rs_launch_plane_sums_reduce(const double* pblk, int rows, int nb, float* out, hipStream_t st);

// Segmentation loss from the per-plane sums (losses_foundation.py:945-956 + DiceLossMultiClass :541-607) and its Jacobian.
struct SegSumsParams {
    const float* sums;   // [B*C][6] (S, A, Bs, Cn, -, -)
    const float* cw;     // [B*C] class weights or nullptr
    int B, C;
    double inv_bcv;      // 1 / (B*C*V): BCE mean
    double scale;        // aux weight * seg_loss
    float* loss;         // [1]
    float* dsums;        // [B*C][6] d loss / d sums
};
int rs_launch_seg_from_sums(const SegSumsParams& p, hipStream_t st);

// Report terms (volume loss :250-349 + dice_based_volume_loss :352-395; ball loss tail :1625-1661, :1793-1811 with the Dice of :541-607) from the
// per-plane sums of the report terms, and their Jacobians -- a few dozen scalars per head: on the host through torch's CPU autograd this cost two
// device <-> host round trips and ~0.6 ms of idle GPU per step (config 3), as ATen device ops ~200 launch-bound kernels.
struct ReportSumsParams {
    const float* sums;      // [R][6] rows of the report terms, in the order calculate_loss builds them: volume rows li * B + b, then the plans' rows
    const float* roww;      // [R] class weight of every row (1 without class weights)
    int B, L;               // samples, lesion groups
    double V;               // voxels per plane
    int use_vol;            // rows [0, L*B) are the volume terms
    const float* flags;     // [B][2L]: annotated-tumour flag | segment-present gate  (use_vol)
    const float* rvol;      // [B] report volume (use_vol)
    double tol, E;          // dice_based_volume_loss parameters
    int nplans;             // ball plans (0: no ball loss)
    const int* plan;        // [nplans][2]: kind (0 = no tumour: L rows, 1 = tumour: 1 row), first row
    int apply_dice, standard_ce;
    float* loss;            // [3]: ball_loss_bce, ball_loss_dice, dice_volume_loss
    float* jac;             // [3][R][6] d loss_k / d sums
};
int rs_launch_report_from_sums(const ReportSumsParams& p, int R, hipStream_t st);

int rs_launch_plane_partials(const PlaneParams& p, int planes, int bwd, hipStream_t st);
int rs_launch_sigmoid_mask(const float* x, const uint8_t* m, float* out, size_t V, hipStream_t st);
int rs_launch_window_accumulate(const float* logits, float* acc, int BK, int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0,
                                int assign, hipStream_t st);
int rs_launch_window_normalize(float* acc, const float* cd, const float* ch, const float* cw, long BK, int D, int H, int W, hipStream_t st);
