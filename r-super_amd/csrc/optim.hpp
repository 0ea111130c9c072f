#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define RS_MT_MAX 36
// Pointer table passed by value as a kernel argument (multi-tensor apply).
struct MTChunk {
    int n;
    void* p[RS_MT_MAX]; void* g[RS_MT_MAX]; void* m[RS_MT_MAX]; void* v[RS_MT_MAX]; void* ema[RS_MT_MAX];
    size_t numel[RS_MT_MAX];
    int blk_start[RS_MT_MAX + 1];   // prefix sum of blocks per tensor
};
struct AdamParams {
    float lr, beta1, beta2, eps, wd;
    float step_size;     // lr / (1 - beta1^t)
    float sqrt_bc2;      // sqrt(1 - beta2^t)
    float ema_alpha;     // min(1 - 1/(step+1), ema_alpha)
    float max_norm;
    const float* dyn;    // optional device array [lr, step_size, sqrt_bc2, ema_alpha] overriding the fields above (hipGraph replay)
};
int rs_mt_blocks(size_t numel);
int rs_launch_sqnorm(const MTChunk& c, double* total, int first, hipStream_t st);   // first: assigns *total (no memset of the accumulator)
int rs_launch_zero_bytes(void* p, size_t bytes, hipStream_t st);                   // zero-fill kernel (16-byte aligned pointer)
int rs_launch_adamw_ema(const MTChunk& c, const AdamParams& a, const double* total_sq, hipStream_t st);
int rs_launch_scale(const MTChunk& c, float max_norm, const double* total_sq, hipStream_t st);
