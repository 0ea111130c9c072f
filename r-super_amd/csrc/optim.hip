// Optimiser side of the training step (gfx950, HBM-bound), multi-tensor:
//   global grad L2 norm  -> clip coefficient (torch.nn.utils.clip_grad_norm_(params, 1.0), train_ddp.py:352)
//   AdamW(eps=1e-5) step (training/utils.py:46-51)  fused with
//   EMA update ema = a*ema + (1-a)*p               (update_ema_variables, training/utils.py:154-161)
// One pass reads g, p, m, v, ema and writes p, m, v, ema; the clip coefficient is read from device memory so
// the host never synchronises on the norm.
#include "common.hpp"
#include "optim.hpp"

namespace {

__device__ __forceinline__ int find_tensor(const MTChunk& c, int blk, int& local) {
    int lo = 0, hi = c.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c.blk_start[mid] <= blk) lo = mid; else hi = mid; }
    local = blk - c.blk_start[lo];
    return lo;
}

constexpr int MT_ELEMS = 256 * 4 * 4;   // elements per block

// Per-block partial sums (no atomics: thousands of blocks adding f64 into ONE address serialised in L2 and made the kernel
// 3x slower than its HBM bound), then one small block adds them to *total in a fixed order -> the norm is deterministic.
__global__ __launch_bounds__(256) void sqnorm_kernel(MTChunk c, double* partial) {
    int local; const int ti = find_tensor(c, blockIdx.x, local);
    const float* g = (const float*)c.g[ti];
    const size_t n = c.numel[ti], base = (size_t)local * MT_ELEMS;
    float s = 0.f;
    if (base + MT_ELEMS <= n && (((uintptr_t)(g + base)) & 15) == 0) {
        const float4* g4 = (const float4*)(g + base);
#pragma unroll
        for (int k = 0; k < MT_ELEMS / 1024; ++k) { const float4 v = g4[threadIdx.x + 256 * k]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    } else {
        // unaligned tensors (views into the gradient buckets of rsuper_amd.reducer) and tails: the SAME element-to-thread map and the same
        // association as the vector path, so the norm -- and with it the clip coefficient -- does not depend on where a gradient lives
#pragma unroll
        for (int k = 0; k < MT_ELEMS / 1024; ++k) {
            const size_t i = base + (size_t)(threadIdx.x + 256 * k) * 4;
            const float x = i < n ? g[i] : 0.f, y = i + 1 < n ? g[i + 1] : 0.f, z = i + 2 < n ? g[i + 2] : 0.f, w = i + 3 < n ? g[i + 3] : 0.f;
            s += x * x + y * y + z * z + w * w;
        }
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

// first: this chunk starts the sum (assigns) -- there is NO hipMemsetAsync of the accumulator any more: captured in a hipGraph, the memset node
// wrote the byte 0xC0 instead of 0 from the first replay that followed eager launches on, so the norm came out as sqrt(-8577.5 + sum) = NaN
// (found with tools/mode_consistency.py: total_sq = 0xC0C0C0C0C0C0C0C0 + the correct sum at every later replay)
__global__ __launch_bounds__(256) void sqnorm_reduce_kernel(const double* partial, int n, double* total, int first) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = first ? red[0] : *total + red[0];
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(MTChunk c, AdamParams a, const double* total_sq) {
    int local; const int ti = find_tensor(c, blockIdx.x, local);
    float* p = (float*)c.p[ti]; const float* g = (const float*)c.g[ti];
    float* m = (float*)c.m[ti]; float* v = (float*)c.v[ti]; float* e = (float*)c.ema[ti];
    const size_t n = c.numel[ti], base = (size_t)local * MT_ELEMS;
    if (a.dyn) { a.lr = a.dyn[0]; a.step_size = a.dyn[1]; a.sqrt_bc2 = a.dyn[2]; a.ema_alpha = a.dyn[3]; }   // step-dependent scalars of a captured step
    float coef = 1.f;
    if (total_sq) {                                              // clip_grad_norm_: max_norm / (norm + 1e-6), clamped to 1
        const float norm = (float)sqrt(*total_sq);
        // A non-finite gradient norm (NaN loss: the reference raises before backward, losses_foundation.py:1070-1071) skips the whole update: with the
        // guards read one step late (train_ddp.StepGuard) the weights, moments and EMA must still be the ones of the last good step when the host raises.
        // (fminf(NaN, 1) = 1 would otherwise apply the NaN gradients.)
        if (!(norm <= 3.0e38f)) return;
        coef = fminf(a.max_norm / (norm + 1e-6f), 1.f);
    }
    for (size_t i = base + threadIdx.x; i < base + MT_ELEMS && i < n; i += 256) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - a.lr * a.wd);                   // decoupled weight decay
        const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gi;
        const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
        const float denom = sqrtf(vi) / a.sqrt_bc2 + a.eps;
        pi -= a.step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (e) e[i] = a.ema_alpha * e[i] + (1.f - a.ema_alpha) * pi;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(MTChunk c, float max_norm, const double* total_sq) {
    int local; const int ti = find_tensor(c, blockIdx.x, local);
    float* g = (float*)c.g[ti];
    const size_t n = c.numel[ti], base = (size_t)local * MT_ELEMS;
    const float coef = fminf(max_norm / ((float)sqrt(*total_sq) + 1e-6f), 1.f);
    for (size_t i = base + threadIdx.x; i < base + MT_ELEMS && i < n; i += 256) g[i] *= coef;
}

// zero-fill as a KERNEL (never a memset node: see sqnorm_reduce_kernel)
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* p, size_t bytes) {
    const size_t nv = bytes / 16, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) ((uint4*)p)[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0) for (size_t i = nv * 16 + threadIdx.x; i < bytes; i += 256) p[i] = 0;
}

}  // namespace

int rs_launch_zero_bytes(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return RS_OK;
    const size_t nv = bytes / 16;
    size_t blocks = (nv + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    const bool aligned = (((uintptr_t)p) & 15) == 0;
    if (!aligned) return RS_ERR_ARG;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)p, bytes);
    return rs_check_launch();
}
int rs_mt_blocks(size_t numel) { return (int)((numel + MT_ELEMS - 1) / MT_ELEMS); }

int rs_launch_sqnorm(const MTChunk& c, double* total, int first, hipStream_t st) {
    static double* ws[16] = {};                                  // per-device partial sums (stream-ordered reuse)
    constexpr int WS_BLOCKS = 1 << 16;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return RS_ERR_LAUNCH;
    if (!ws[dev] && hipMalloc((void**)&ws[dev], (size_t)WS_BLOCKS * sizeof(double)) != hipSuccess) return RS_ERR_LAUNCH;
    const int blocks = c.blk_start[c.n];
    if (blocks > WS_BLOCKS) return RS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, st, c, ws[dev]);
    hipLaunchKernelGGL(sqnorm_reduce_kernel, dim3(1), dim3(256), 0, st, (const double*)ws[dev], blocks, total, first);
    return rs_check_launch();
}
int rs_launch_adamw_ema(const MTChunk& c, const AdamParams& a, const double* total_sq, hipStream_t st) {
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(c.blk_start[c.n]), dim3(256), 0, st, c, a, total_sq);
    return rs_check_launch();
}
int rs_launch_scale(const MTChunk& c, float max_norm, const double* total_sq, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3(c.blk_start[c.n]), dim3(256), 0, st, c, max_norm, total_sq);
    return rs_check_launch();
}
