#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
int rs_launch_dilate_pass(const uint8_t* in, uint8_t* out, const uint8_t* flags, long nvol, int D, int H, int W, int k, hipStream_t st);
long rs_ball_workspace_floats(int D, int H, int W, int d_odd);
int rs_launch_ball_conv_argmax(const float* x, int D, int H, int W, int d_odd, float std, unsigned long long* best, float* conv_out, float* ws,
                               hipStream_t st);
int rs_launch_insert_ball(uint8_t* out, int D, int H, int W, int cz, int cy, int cx, int d_odd, int half, unsigned int* count, hipStream_t st);
int rs_launch_insert_ball_at(uint8_t* out, int D, int H, int W, const unsigned long long* best, int d_odd, int half, unsigned int* count, hipStream_t st);
int rs_launch_radix_hist(const float* x, const uint8_t* m, long V, uint32_t prefix, int shift, unsigned int* hist, hipStream_t st);
int rs_launch_topk_mark(const float* x, const uint8_t* m, long V, uint32_t thr, unsigned int need_eq, uint8_t* out, hipStream_t st);
int rs_launch_topk_select(const float* x, const uint8_t* m, long V, const unsigned int* k, int nk, uint8_t* out, unsigned int* ws, int clip, hipStream_t st);
int rs_launch_compact(const float* x, const uint8_t* pm, long V, float* vals, uint32_t* idx, unsigned int* n, hipStream_t st);
int rs_launch_rank_weights(const float* vals, const uint32_t* idx, unsigned int n, float dlog2, float scale, float* w, hipStream_t st);
int rs_launch_guard_consistency(const uint8_t* m_any, const uint8_t* u_any, const float* vol, int B, int T, int* flags, hipStream_t st);
int rs_launch_guard_range(const float* x, size_t n, float lo, float hi, int* flags, hipStream_t st);
int rs_launch_plane_any(const uint8_t* m, long planes, long V, uint8_t* flags, hipStream_t st);
int rs_launch_mask_op(uint8_t* a, const uint8_t* b, long V, int op, hipStream_t st);
int rs_launch_unpack_bits(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, hipStream_t st);
int rs_launch_unpack_bits_sel(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, const uint8_t* flags, const uint8_t* force, hipStream_t st);
int rs_launch_plane_any_bits(const uint8_t* packed, int B, int P, int C, long V, uint8_t* flags, hipStream_t st);
int rs_launch_zero_where(float* x, const uint8_t* m, long V, hipStream_t st);
int rs_launch_count(const uint8_t* m, long V, unsigned int* count, hipStream_t st);
int rs_launch_rank_assign(const long long* ids, unsigned int n, float dlog2, float scale, float* w, hipStream_t st);
