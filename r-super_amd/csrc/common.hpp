// Common device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the R-Super hot path.
// wave = 64 lanes; LDS 160 KiB/CU; MFMA fragments per cdna_hip_programming.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_F32 0
#define RS_BF16 1

#define RS_OK 0
#define RS_ERR_ARG 1
#define RS_ERR_LAUNCH 2
#define RS_ERR_UNSUPPORTED 3

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two f32 -> packed bf16 pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32; matches torch's RNE)
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    union { bf16x2_t b; uint32_t u; } c;
    c.b = __builtin_convertvector(v, bf16x2_t);
    return c.u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int DT = RS_F32;
    static constexpr int KP = 4;    // elements per 16-byte vector
    static constexpr int KC = 16;   // channels per 64-byte LDS row (one K chunk)
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int DT = RS_BF16;
    static constexpr int KP = 8;
    static constexpr int KC = 32;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// unpack a 16-byte vector into floats / pack floats into a 16-byte vector
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]), f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
}

// x_hat = relu(x * sc + nb) on one 16-byte vector (the InstanceNorm + ReLU prologue of every conv input, sc = rstd,
// nb = -mean * rstd per channel).  bf16: pairs go through v_pk_fma_f32, are rounded by v_cvt_pk_bf16_f32 and the ReLU is a
// packed signed 16-bit max on the bf16 bit patterns (negative bf16 == negative int16; rounding is monotonic and
// round(0) = 0, so max-after-round == round-after-max bit for bit, -0.0 included) -> 20 VALU ops instead of 28.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ uint4 norm_relu16(const uint4& q, const float* sc, const float* nb);
template <> __device__ __forceinline__ uint4 norm_relu16<float>(const uint4& q, const float* sc, const float* nb) {
    float f[4];
    unpack16<float>(q, f);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], nb[j]), 0.f);
    return pack16<float>(f);
}
template <> __device__ __forceinline__ uint4 norm_relu16<bf16_t>(const uint4& q, const float* sc, const float* nb) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2_t x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
        const f32x2_t s2 = {sc[2 * j], sc[2 * j + 1]}, b2 = {nb[2 * j], nb[2 * j + 1]};
        x = __builtin_elementwise_fma(x, s2, b2);
        const uint32_t r = f2bf2(x[0], x[1]);
        i16x2_t v = __builtin_bit_cast(i16x2_t, r);
        const i16x2_t z = {0, 0};
        v = __builtin_elementwise_max(v, z);
        o[j] = __builtin_bit_cast(uint32_t, v);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// One K step of a 32x32 MFMA tile from 16-byte operand vectors.
//   bf16: 8 k per lane half -> one v_mfma_f32_32x32x16_bf16 (K=16)
//   f32 : 4 k per lane half -> four v_mfma_f32_32x32x2_f32 (K=8); any k order works as long as
//         A and B use the same one, so element j of both vectors is paired.
template <typename T> __device__ __forceinline__ void mma32(f32x16_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma32<bf16_t>(f32x16_t& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8_t v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma32<float>(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Tile index inside a sample -> tile coordinates.  Bricks of 2 x 2 x 4 tiles (w, h, d; smaller where the tile grid does not divide) are numbered
// consecutively: the blocks of an XCD that work side by side on consecutive tile indices (XCD-aware orders of the igemm / weight-gradient kernels) then
// hold ONE compact brick -- its interior halo rows are fetched into that L2 once -- instead of a 16-tile strip along w whose d neighbours come round
// only after the strip's rows have left the cache (the d halo is 2 of every 6 staged planes).  A bijection for every tile grid; placement only.
__device__ __forceinline__ void rs_tile_coords(int t, int tiles_w, int tiles_h, int tiles_d, int& tw, int& th, int& td) {
    const int lw = (tiles_w & 1) ? 0 : 1, lh = (tiles_h & 1) ? 0 : 1, ld = (tiles_d & 3) == 0 ? 2 : ((tiles_d & 1) ? 0 : 1);   // log2 of the brick extents
    const int i = t & ((1 << (lw + lh + ld)) - 1), b = t >> (lw + lh + ld);
    const int nbw = tiles_w >> lw, nbh = tiles_h >> lh;
    const int bx = b % nbw, r = b / nbw, by = r % nbh, bz = r / nbh;
    tw = (bx << lw) | (i & ((1 << lw) - 1));
    th = (by << lh) | ((i >> lw) & ((1 << lh) - 1));
    td = (bz << ld) | (i >> (lw + lh));
}

// C/D fragment row of accumulator register `reg` for a 32x32 MFMA (column = lane & 31).
__device__ __forceinline__ int cd_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Fragment row (0..31) -> position inside a [2 h][16 w] voxel patch.  ds_read_b128 is serviced in the
// lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}; mapping each group to one h-row keeps the 16 LDS
// rows of a group distinct mod 16, so a padded 80-byte row pitch is bank-conflict free.
__host__ __device__ constexpr int row_hw_packed(int i) {   // hs * 16 + w, usable in constant expressions
    return i < 4 ? i : i < 12 ? 16 + i - 4 : i < 16 ? i - 8 : i < 20 ? 16 + i - 8 : i < 28 ? i - 12 : 16 + i - 16;
}
__device__ __forceinline__ void row_to_hw(int i, int& hs, int& w) {
    if (i < 4) { hs = 0; w = i; }
    else if (i < 12) { hs = 1; w = i - 4; }
    else if (i < 16) { hs = 0; w = i - 8; }
    else if (i < 20) { hs = 1; w = i - 8; }
    else if (i < 28) { hs = 0; w = i - 12; }
    else { hs = 1; w = i - 16; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int rs_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RS_OK : RS_ERR_LAUNCH;
}
