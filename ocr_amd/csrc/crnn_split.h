// Device helpers shared by crnn_kernels.hip and crnn_kernels_nopk.hip: the F(4x4,3x3) transform rows and the f16x3 split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Winograd F(4x4,3x3) (fp32), interpolation points 0, 1, -1, 2, -1/2, inf.  The textbook set 0, +-1, +-2 was
// measured to give three times the rounding error on conv1's input (tests/exp_winograd_f4_error.py);
// all coefficients below are exact in binary floating point.
//   B^T = [1 3/2 -2 -3/2 1 0; 0 -1 -5/2 -1/2 1 0; 0 1 1/2 -5/2 1 0; 0 -1/2 -1 1/2 1 0; 0 2 -1 -2 1 0; 0 1 3/2 -2 -3/2 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -1/2 0; 0 1 1 4 1/4 0; 0 1 -1 8 -1/8 1]
// (G, applied to the weights on the host in double, is in crnn_api.hip.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wino4_bt(const f32x4 (&d)[6], f32x4 (&u)[6]) {
  u[0] = d[0] + 1.5f * (d[1] - d[3]) - 2.f * d[2] + d[4];
  u[1] = d[4] - d[1] - 2.5f * d[2] - 0.5f * d[3];
  u[2] = d[4] + d[1] + 0.5f * d[2] - 2.5f * d[3];
  u[3] = d[4] - d[2] + 0.5f * (d[3] - d[1]);
  u[4] = d[4] - d[2] + 2.f * (d[1] - d[3]);
  u[5] = d[1] + 1.5f * (d[2] - d[4]) - 2.f * d[3] + d[5];
}
__device__ __forceinline__ void wino4_at(const f32x4 (&m)[6], f32x4 (&r)[4]) {
  const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
  r[0] = m[0] + s12 + m[3] + m[4];
  r[1] = d12 + 2.f * m[3] - 0.5f * m[4];
  r[2] = s12 + 4.f * m[3] + 0.25f * m[4];
  r[3] = d12 + 8.f * m[3] - 0.125f * m[4] + m[5];
}

// ---------------------------------------------------------------------------
// "f16x3" operand format (CRNN_F16X3, default on): fp32-accurate GEMMs on the fp16 MFMA.  Every fp32 operand x of a GEMM
// row is stored as two fp16 numbers hi = fp16(x s), lo = fp16(x s - hi) -- together 22 significant bits, the
// representation error 2^-23 |x| is below fp32's own rounding -- with s a power of two per ROW (per Winograd tile /
// per GEMM row) that puts the row's largest magnitude in [2^14, 2^15), so lo stays a normal fp16 down to 2^-15 of
// the row's maximum (below that its absolute error, 2^-38 of the maximum, is far under the fp32 rounding of the
// row's larger terms).  Layout: per 8 consecutive k, 8 hi then 8 lo (32 bytes: the bytes of 8 floats), so a row
// is as long as its fp32 form and the 16-byte chunks of a K = 32 step are hi(k0-7) lo(k0-7) hi(k8-15) ...
// The GEMM then computes hi_a hi_b + hi_a lo_b + lo_a hi_b (the dropped lo lo term is 2^-24 relative) with three
// v_mfma_f32_32x32x16_f16 per fragment pair, fp32 accumulate, at 16x the f32 MFMA rate; the consumer of the
// product multiplies by 1 / (s_row s_weights) (powers of two: exact).  Measured against float64 the result is as
// accurate as the f32 MFMA's or better (fewer roundings: 16 products are summed per instruction; tools/split_mfma_test.hip).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void split_scale_from_max(float mx, float& s, float& uns) {
  unsigned eb = (__builtin_bit_cast(unsigned, mx) >> 23) & 0xFFu;   // biased exponent of the row's largest magnitude
  eb = eb < 32u ? 32u : (eb > 250u ? 250u : eb);                    // all-zero / tiny rows, infinities: keep s finite
  s = __builtin_bit_cast(float, (268u - eb) << 23);                 // 2^(14 - E): max * s in [2^14, 2^15)
  uns = __builtin_bit_cast(float, (eb - 14u) << 23);                // 1 / s
}
// four scaled values -> 4 hi and 4 lo halves (8 bytes each)
__device__ __forceinline__ void split4(f32x4 v, float s, uint2& hi, uint2& lo) {
  union { _Float16 h[4]; uint2 u; } ph, pl;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float x = v[e] * s;
    ph.h[e] = (_Float16)x;
    pl.h[e] = (_Float16)(x - (float)ph.h[e]);
  }
  hi = ph.u;
  lo = pl.u;
}
// lanes 2i / 2i+1 hold channels [8g, 8g+4) / [8g+4, 8g+8) of one row: the even lane stores the 16-byte hi chunk, the
// odd lane the 16-byte lo chunk (one 16-byte store per lane, as in the fp32 form)
// the 16 bytes the lane stores: lanes trade halves so that the even lane has the hi chunk, the odd lane the lo chunk
__device__ __forceinline__ i32x4 split_pair_chunk(int c4, uint2 hi, uint2 lo) {
  const bool odd = c4 & 1;
  const uint2 send = odd ? hi : lo;
  uint2 recv;
  recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
  recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
  const uint4 out = odd ? uint4{recv.x, recv.y, lo.x, lo.y} : uint4{hi.x, hi.y, recv.x, recv.y};
  return i32x4{(int)out.x, (int)out.y, (int)out.z, (int)out.w};
}
__device__ __forceinline__ void store_split_pair(float* rowbase, int c4, uint2 hi, uint2 lo) {
  const bool odd = c4 & 1;
  const uint2 send = odd ? hi : lo;
  uint2 recv;
  recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
  recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
  const uint4 out = odd ? uint4{recv.x, recv.y, lo.x, lo.y} : uint4{hi.x, hi.y, recv.x, recv.y};
  // V is written once and read by the GEMM much later than a cache could keep it: nontemporal stores (0.453 -> 0.416 ms
  // over the six input transforms of a step)
  __builtin_nontemporal_store(i32x4{(int)out.x, (int)out.y, (int)out.z, (int)out.w}, (i32x4*)(rowbase + (c4 >> 1) * 8 + (odd ? 4 : 0)));
}

}  // namespace crnn
