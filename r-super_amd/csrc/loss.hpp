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
};

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

int rs_launch_plane_partials(const PlaneParams& p, int planes, int bwd, hipStream_t st);
int rs_launch_sigmoid_mask(const float* x, const uint8_t* m, float* out, size_t V, hipStream_t st);
int rs_launch_window_accumulate(const float* logits, float* acc, int BK, int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0,
                                int assign, hipStream_t st);
int rs_launch_window_normalize(float* acc, const float* cd, const float* ch, const float* cw, long BK, int D, int H, int W, hipStream_t st);
