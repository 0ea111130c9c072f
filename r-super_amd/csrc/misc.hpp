// Launch interfaces of the HBM-bound UNet kernels (unet_misc.hip), the loss kernels (loss.hip),
// the morphology kernels (morph.hip) and the optimiser kernels (optim.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct InBwdParams {
    const void* g; int ldg;        // masked upstream gradient g = dX_hat * [x_hat > 0]
    const void* x; int ldx;        // forward input of the norm
    const float* mr;               // [N][C][2] mean, rstd
    const float* gm;               // [N][C][2] mean(g), mean(g * x_n)
    const void* add1; int lda1;    // optional extra gradient terms (residual / skip paths)
    const void* add2; int lda2;
    void* out; int ldo;
    int N, vox, C;
};

struct PoolParams {
    const void* x; int ldx;        // forward input
    void* y; int ldy;              // forward: pooled output; backward: pooled gradient (read)
    void* dx; int lddx;            // backward output
    float* part;                   // forward: partial stats [N][blocks][C][2] or nullptr
    int N, D, H, W, C;             // input dims
    const void* add; int lda;      // max-pool backward: optional second gradient of x (the skip connection's) added to every voxel
};

struct UpParams {
    const void* x; int ldx;        // forward input (low resolution)
    void* y; int ldy;              // forward: output; backward: output gradient (read)
    void* dx; int lddx;
    float* part;
    int N, ID, IH, IW, OD, OH, OW, C;
};

struct StemParams {
    const float* x;                // [N][D][H][W] f32 (single input channel)
    const float* w;                // (C,1,3,3,3)
    void* y; int ldy;              // forward: output; wgrad: dY (read)
    float* part;                   // forward partial stats [N][blocks][C][2]
    float* dw;                     // wgrad output (C,27) f32, pre-zeroed
    int N, D, H, W, C;
};

struct HeadParams {
    const void* x; int ldx;        // features, channels-last
    const float* w; const float* b;   // (K,C), (K)
    float* logits;                 // forward: out [N][K][vox] f32; backward: d logits (read)
    void* dx; int lddx;
    float* dw; float* db;          // pre-zeroed
    int N, vox, C, K;
};

int rs_cnorm_rows(long vox);
int rs_launch_cnorm_small(const float* x, const float* dy, const float* mr, float* out, float* mr_out, int N, long vox, int C, int relu, float eps, int mode, hipStream_t st);
int rs_launch_cnorm_stats(const float* x, const float* dy, const float* mr, float* part, int N, long vox, int C, int relu, int mode, hipStream_t st);
int rs_launch_cnorm_apply(const float* x, const float* dy, const float* mr, const float* gm, float* out, int N, long vox, int C, int relu, int mode, hipStream_t st);
int rs_battn_supported(int T, int dh, int heads);
int rs_battn_chunks(int L, int heads);
int rs_launch_battn(const float* fqv, const float* mqv, float* fout, float* mout, float* lse, const float* dfo, const float* dmo,
                    float* dfqv, float* dmqv, float* part, float* pms, int B, int L, int T, int heads, int dh, float scale, int bwd,
                    hipStream_t st);
int rs_launch_se_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* part, float* ms, float* tab,
                         float* hbuf, float* y, int N, long vox, int C, int r, hipStream_t st);
int rs_launch_se_backward(const float* x, const float* dy, const float* ident, const float* w1, const float* w2, const float* ms, const float* tab,
                          const float* hbuf, float* part, float* gm, float* dz1buf, float* tab2, float* dx, float* dw1, float* db1, float* dw2,
                          float* db2, int N, long vox, int C, int r, hipStream_t st);
int rs_launch_cl_planar(const float* src, float* dst, int N, long vox, int C, int K, int dir, hipStream_t st);
int rs_depthwise_rows(long vox);
int rs_launch_depthwise(const float* x, const float* w, float* y, int N, int D, int H, int W, int C, int flip, hipStream_t st);
int rs_launch_depthwise_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int D, int H, int W, int C, hipStream_t st);
int rs_launch_stats_finalize(const float* part, int N, int nblk, int C, double cnt, float eps, int mode, int split, float* out, hipStream_t st);
int rs_elem_blocks(size_t items);
int rs_launch_in_bwd(const InBwdParams& p, int dtype, hipStream_t st);
int rs_launch_pool(const PoolParams& p, int dtype, int bwd, int blocks, hipStream_t st);
int rs_launch_subsample(const PoolParams& p, int dtype, int bwd, int blocks, hipStream_t st);
int rs_launch_upsample(const UpParams& p, int dtype, int bwd, int blocks, hipStream_t st);
int rs_launch_stem(const StemParams& p, int dtype, int wgrad, hipStream_t st);
int rs_launch_head(const HeadParams& p, int dtype, int which, hipStream_t st);
