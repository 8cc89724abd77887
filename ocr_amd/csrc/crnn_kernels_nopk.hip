// Kernels compiled WITHOUT packed-fp32 instructions (Makefile: -target-feature -packed-fp32-ops for this file only).
// Why: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel on src1 (the LOW result taking the HIGH dword of the source)
// were measured on MI355X to return a wrong low result now and then while a wave of ANOTHER kernel executes 16-bit
// MFMAs on the same CU (tools/corun_test.hip: 200-2000 wrong results in 2e9 with op_sel, none with op_sel_hi only or
// no modifier, none without the co-running kernel; wait states after the instruction change nothing).  The compiler
// picks those forms by itself when a scalar it broadcasts sits in an odd register (the float2 patch loads from LDS in
// conv0_win4_h3_kernel; a horizontal sum in cls_select_kernel's bounds), and with several lanes of a handle or several
// callers in flight any kernel can run beside an MFMA kernel -- found as one wrong character in 1 of ~1500 calls when
// two threads shared a handle (tools/stress_threads.py).  The kernels in which it happened live here; for every other
// kernel tools/check_isa.py, run by the Makefile on the linked library, fails the build if such an instruction appears.
#include "crnn_kernels.h"
#include "crnn_split.h"

#include <cmath>

namespace crnn {

// ---------------------------------------------------------------------------
// conv0 + activation + pool0 AND conv1's F(4x4,3x3) input transform (f16x3 form) in one kernel: the pooled conv0
// output (147 MB per 256-line step, written and read back by the two-kernel form) only ever exists as an LDS tile.
// Block = (line, 4 tile columns): phase 1 computes the 16 x 18 pooled pixels x 64 channels those 16 tiles touch
// (one column of halo on each side: 12.5 % recomputed) as conv0_pool_kernel does, into LDS (72 KB, two blocks per
// CU); phase 2 is wino4_in_h3_kernel reading its 6 x 6 patches from LDS: thread = (tile, 4 channels), the tile's
// scale from the 16 lanes that share it.  Rows -1 and 16 and columns outside [0, min(W/2, w1valid)) are zeros.
// ---------------------------------------------------------------------------
// MF (round 3, CRNN_CONV0_MFMA, default on): phase 1 on the fp16 MFMA in the f16x3 form instead of 2592 scalar FMAs per
// thread (this file has no packed fp32).  The conv is a GEMM with K = 9: M = the 1152 conv positions of the block ordered
// (pooled pixel, position in its 2x2 window), N = 64 channels, the nine taps padded to one K = 16 step of
// v_mfma_f32_32x32x16_f16 -- the positions as FIRST operand, so a lane ends with the four window positions of two
// pooled pixels for one channel and the pool is three v_max in registers.  The window sits in LDS as (hi, lo) fp16
// planes (scale 2^14: |x| <= 1), the weights as (hi, lo) fragments in registers (w3: [64][16] in to_f16x3_host's layout,
// us3 = 1 / (2^14 s_weights)); 6 MFMAs per 32 positions.  The kernel is VALU-bound: 0.129 -> 0.114 ms (what remains
// is phase 2, the same transform + split work as a whole wino4_in_h3_kernel launch).
constexpr int C0_TXB = 4, C0_PC = 4 * C0_TXB + 2;
typedef _Float16 f16x8n __attribute__((ext_vector_type(8)));
typedef float f32x16n __attribute__((ext_vector_type(16)));
template <bool MF>
__global__ __launch_bounds__(256, 2) void conv0_win4_h3_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ w9x64,
                                                               const float* __restrict__ bias, float* __restrict__ V,
                                                               float* __restrict__ vs,
                                                               const int32_t* __restrict__ w1valid, int B, int W,
                                                               int act, long ntile_pad,
                                                               const uint16_t* __restrict__ w3, float us3) {
  constexpr int IC = 2 * C0_PC + 4;   // input window: columns 2 x1base - 1 ... 2 (x1base + C0_PC - 1) + 2, padded to 40
  __shared__ __attribute__((aligned(16))) float P[16 * C0_PC * 64];
  __shared__ __attribute__((aligned(16))) float I[MF ? 4 : 34 * IC];
  __shared__ __attribute__((aligned(16))) _Float16 Ih[MF ? 34 * IC : 8], Il[MF ? 34 * IC : 8];
  const int W1 = W >> 1, ntx = (W1 + 3) >> 2, ngrp = (ntx + C0_TXB - 1) / C0_TXB;
  const int cg = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  {   // one block per (line, group of 4 tile columns): a grid-stride loop here makes the compiler hoist the per-thread
      // index arithmetic of all three phases out of it and spill (scratch memory; none is used now)
    const int blk = blockIdx.x;
    const int b = blk / ngrp, grp = blk - b * ngrp;
    const int x1base = 4 * C0_TXB * grp - 1;
    const int wlim = w1valid ? (w1valid[b] < W1 ? w1valid[b] : W1) : W1;
    const float* ip = in + (long)b * 32 * W;
    // ---- phase 0: the zero-padded input window (rows -1 .. 32) into LDS
    for (int i = threadIdx.x; i < 34 * IC; i += 256) {
      const int r = i / IC, c = i - r * IC;
      const int yy = r - 1, xx = 2 * x1base - 1 + c;
      const float v = ((unsigned)yy < 32u && (unsigned)xx < (unsigned)W) ? ip[yy * W + xx] : 0.f;
      if (MF) {
        const float x = v * 16384.f;
        const _Float16 hi = (_Float16)x;
        Ih[i] = hi;
        Il[i] = (_Float16)(x - (float)hi);
      } else {
        I[i] = v;
      }
    }
    __syncthreads();
    // ---- phase 1: pooled conv0 pixels into LDS
    if (MF) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i32 = lane & 31, kg = lane >> 5;
      f16x8n wh[2], wl[2];
      float bsl[2];
#pragma unroll
      for (int nb = 0; nb < 2; nb++) {
        const uint4* wp = (const uint4*)(w3 + ((nb * 32 + i32) * 32 + kg * 16));
        wh[nb] = __builtin_bit_cast(f16x8n, wp[0]);
        wl[nb] = __builtin_bit_cast(f16x8n, wp[1]);
        bsl[nb] = bias[nb * 32 + i32];
      }
      // the lane's eight k: taps (ky, kx) = 0..7 for the low half of the wave, tap 8 and seven zeros for the high half
      int toff[8];
#pragma unroll
      for (int t = 0; t < 8; t++) toff[t] = kg ? 2 * IC + 2 : (t / 3) * IC + t % 3;
      const int e = i32 & 3;
#pragma unroll 1
      for (int mb = wave; mb < 2 * C0_PC; mb += 4) {   // 16 * C0_PC / 8 blocks of 8 pooled pixels = 32 positions
        const int pp = 8 * mb + (i32 >> 2);
        const int y1 = pp / C0_PC, jx = pp - y1 * C0_PC;
        const int base = (2 * y1 + (e >> 1)) * IC + 2 * jx + (e & 1);   // (input row 2 y1 - 1 + dy, column 2 x1 - 1 + dx)
        f16x8n xh, xl;
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const bool on = !kg || t == 0;
          xh[t] = on ? Ih[base + toff[t]] : (_Float16)0;
          xl[t] = on ? Il[base + toff[t]] : (_Float16)0;
        }
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
          f32x16n acc;
#pragma unroll
          for (int r = 0; r < 16; r++) acc[r] = 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[nb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[nb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[nb], acc, 0, 0, 0);
          // acc[4 g + w] = position 8 g + 4 kg + w of the block = window position w of pooled pixel 2 g + kg, channel i32
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int po = 8 * mb + 2 * g + kg;
            const int jo = po % C0_PC, x1 = x1base + jo;
            float v = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3])) * us3 + bsl[nb];
            if (act == 1) v = v > 0.f ? v : 0.f;
            else if (act == 2) v = v > 0.f ? v : 0.2f * v;
            if ((unsigned)x1 >= (unsigned)wlim) v = 0.f;
            P[po * 64 + nb * 32 + i32] = v;
          }
        }
      }
    } else {
      f32x4 wr[9];
#pragma unroll
      for (int t = 0; t < 9; t++) wr[t] = *(const f32x4*)(w9x64 + t * 64 + cg * 4);
      const f32x4 bs = *(const f32x4*)(bias + cg * 4);
      for (int pp = sub; pp < 16 * C0_PC; pp += 16) {
        const int y1 = pp / C0_PC, jx = pp - y1 * C0_PC, x1 = x1base + jx;
        f32x4 best = z;
        if ((unsigned)x1 < (unsigned)wlim) {
          float patch[4][4];
#pragma unroll
          for (int dy = 0; dy < 4; dy++) {
            const float2 lo = *(const float2*)(I + (2 * y1 + dy) * IC + 2 * jx);       // input row 2 y1 - 1 + dy, column 2 x1 - 1
            const float2 hi = *(const float2*)(I + (2 * y1 + dy) * IC + 2 * jx + 2);
            patch[dy][0] = lo.x; patch[dy][1] = lo.y; patch[dy][2] = hi.x; patch[dy][3] = hi.y;
          }
#pragma unroll
          for (int py = 0; py < 2; py++)
#pragma unroll
            for (int px = 0; px < 2; px++) {
              f32x4 a = bs;
#pragma unroll
              for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) a += wr[ky * 3 + kx] * patch[py + ky][px + kx];
              if (py == 0 && px == 0) best = a;
              else {
#pragma unroll
                for (int c = 0; c < 4; c++) best[c] = a[c] > best[c] ? a[c] : best[c];
              }
            }
#pragma unroll
          for (int c = 0; c < 4; c++) {
            float v = best[c];
            if (act == 1) v = v > 0.f ? v : 0.f;
            else if (act == 2) v = v > 0.f ? v : 0.2f * v;
            best[c] = v;
          }
        }
        *(f32x4*)(P + pp * 64 + cg * 4) = best;
      }
    }
    __syncthreads();
    // ---- phase 2: B^T d B of tile (ty, tx), 4 channels per thread.  The tile's scale comes from a BOUND on its largest
    // component, 49 max|d| (the rows of B^T sum to at most 7 in magnitude), so that every row of the result can be
    // split and stored as soon as it is computed (144 instead of ~300 live registers: two blocks per CU).  The bound
    // is loose by 2-3 bits, which only moves the point below which lo parts go subnormal (the fp16 MFMA honours them).
    {
      const int ty = sub >> 2, txl = sub & 3, tx = C0_TXB * grp + txl;
      const bool valid = tx < ntx;   // uniform over the 16 lanes of a tile
      f32x4 u[6][6];
      float mx = 0.f;
      int ro[6];   // six row addresses; the columns are immediate offsets (36 separate addresses get hoisted and spilled)
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int yy = 4 * ty - 1 + k;
        ro[k] = ((yy < 0 ? 0 : (yy > 15 ? 15 : yy)) * C0_PC + 4 * txl) * 64 + cg * 4;
      }
#pragma unroll
      for (int j = 0; j < 6; j++) {
        f32x4 d[6], c[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const int yy = 4 * ty - 1 + k;
          d[k] = *(const f32x4*)(P + ro[k] + j * 64);
          if ((unsigned)yy >= 16u) d[k] = z;
#pragma unroll
          for (int e = 0; e < 4; e++) mx = fmaxf(mx, __builtin_fabsf(d[k][e]));
        }
        wino4_bt(d, c);
#pragma unroll
        for (int k = 0; k < 6; k++) u[k][j] = c[k];
        __builtin_amdgcn_sched_barrier(0);   // one column of loads in flight, not all six (registers: no spills, no scratch)
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      float sc, uns;
      split_scale_from_max(49.f * mx, sc, uns);
      const long t = ((long)b * 4 + ty) * ntx + tx;
      if (valid && cg == 0) vs[t] = uns;
      float* vrow = V + t * 64;   // component after component: one running pointer, not 36 addresses
      const long cstride = ntile_pad * 64;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        f32x4 v[6];
        wino4_bt(u[k], v);
#pragma unroll
        for (int j = 0; j < 6; j++) {
          uint2 hi, lo;
          split4(v[j], sc, hi, lo);
          if (valid) store_split_pair(vrow, cg, hi, lo);
          vrow += cstride;
        }
      }
    }
    __syncthreads();
  }
}

void launch_conv0_win4(const float* in, const float* w9x64, const float* bias, float* V, float* vs,
                       const int32_t* w1valid, int B, int W, int act, hipStream_t s, const uint16_t* w3, float us3) {
  const int W1 = W >> 1, ntx = (W1 + 3) >> 2, ngrp = (ntx + C0_TXB - 1) / C0_TXB;
  const long blocks = (long)B * ngrp;
  if (w3)
    hipLaunchKernelGGL(conv0_win4_h3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, in, w9x64, bias, V, vs, w1valid,
                       B, W, act, wino4_pad(wino4_tiles(B, 16, W1)), w3, us3);
  else
    hipLaunchKernelGGL(conv0_win4_h3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, in, w9x64, bias, V, vs, w1valid,
                       B, W, act, wino4_pad(wino4_tiles(B, 16, W1)), w3, us3);
}

constexpr int CLS_MAXCH = 16;   // chunks of 512 classes a row may have (N <= 8192)
// F16: the approximate logits come from fp16 operands (u = 2^-11 instead of bf16's 2^-8) and are stored as fp16.  The
// bound then is  e_c = 0.00102 ||a|| ||w_c||  ((2u + u^2) = 0.000977 plus 3.1e-5 for the fp32 accumulation of 512 terms)
// + 0.0005 |l~_c| + 1e-7 (fp16 storage rounding, subnormal results) + 7e-7 (||a|| + ||w_c||) (operands below the fp16
// normal range round with an ABSOLUTE error of at most 2^-25 each: sum_k 2^-25 |w_k| <= 2^-25 sqrt(K) ||w||, K <= 512).
// It is 8x tighter than the bf16 one: on the synthetic 5530-class head about one candidate is left per row instead of four,
// so the exact scoring is rarely needed.  Requires |a|, |w| below the fp16 maximum (the caller checks: LSTM head, |h| < 1).
template <bool F16>
__global__ __launch_bounds__(256) void cls_select_kernel(const uint16_t* __restrict__ L, const float* __restrict__ A,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ anorm,
                                                         const float* __restrict__ wnorm, int32_t* __restrict__ idx,
                                                         int M, int N, int K, int ldl, float wnmax) {
  // One wave per row.  The approximate logits are stored as bf16 in rows of ldl (a multiple of 8) elements, so a
  // lane reads eight classes per 16-byte load and the whole row (<= 16 loads per lane, issued together) stays in
  // registers for both steps: best lower bound, then candidates.  The bf16 storage rounding, at most 2^-8 |l|
  // (0.004 |l| used), is added to the bound.
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const uint4* l = (const uint4*)(L + (long)row * ldl);
  // (2u + u^2) with u = 2^-8 is 0.0078278; 0.00785 leaves room for the fp32 accumulation of 512 terms
  const float an = anorm[row] * (F16 ? 0.00102f : 0.00785f);
  const float k2 = F16 ? 0.0005f : 0.004f, aabs = F16 ? 7e-7f * anorm[row] + 1e-7f : 0.f, wabs = F16 ? 7e-7f : 0.f;
  uint4 v[CLS_MAXCH];
#pragma unroll
  for (int ch = 0; ch < CLS_MAXCH; ch++) {
    const int c = ch * 512 + lane * 8;
    v[ch] = c < N ? l[ch * 64 + lane] : uint4{0, 0, 0, 0};
    if (F16) {   // fp16 form: columns from N on (the row's padding, or nothing at all) read as -inf
      if (c >= N) v[ch] = uint4{0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u};
      else if (c + 8 > N) {
        unsigned w[4] = {v[ch].x, v[ch].y, v[ch].z, v[ch].w};
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (c + e >= N) w[e >> 1] = (e & 1) ? ((w[e >> 1] & 0x0000FFFFu) | 0xFC000000u) : ((w[e >> 1] & 0xFFFF0000u) | 0x0000FC00u);
        v[ch] = uint4{w[0], w[1], w[2], w[3]};
      }
    }
  }
  auto val = [&](const uint4& q, int e) {   // element e of the eight bf16 in q
    const unsigned w = e < 2 ? q.x : (e < 4 ? q.y : (e < 6 ? q.z : q.w));
    if (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)((e & 1) ? (w >> 16) : (w & 0xFFFFu)));
    return __builtin_bit_cast(float, (e & 1) ? (w & 0xFFFF0000u) : (w << 16));
  };
  // e_c for this lane's eight classes of chunk ch (wnorm is padded with zeros to a multiple of 8 by the host)
  auto bounds = [&](int ch, float (&lo)[8], float (&hi)[8]) {
    const int c = ch * 512 + lane * 8;
    const bool in = c < N;
    const f32x4 w0 = in ? *(const f32x4*)(wnorm + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 w1 = in ? *(const f32x4*)(wnorm + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float x = val(v[ch], e);
      const float wn = e < 4 ? w0[e & 3] : w1[e & 3];
      const float ec = an * wn + k2 * __builtin_fabsf(x) + (aabs + wabs * wn);
      const bool ok = c + e < N;
      lo[e] = ok ? x - ec : -INFINITY;
      hi[e] = ok ? x + ec : -INFINITY;
    }
  };
  float lb = -INFINITY;
  __shared__ int cand_s[4][64];
  __shared__ float cand_hi[4][64];
  int* cand = cand_s[threadIdx.x >> 6];
  float* chi = cand_hi[threadIdx.x >> 6];
  int count = 0;
  if (!F16) {
#pragma unroll
    for (int ch = 0; ch < CLS_MAXCH; ch++) {
      if (ch * 512 < N) {
        float lo[8], hi[8];
        bounds(ch, lo, hi);
#pragma unroll
        for (int e = 0; e < 8; e++) lb = fmaxf(lb, lo[e]);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lb = fmaxf(lb, __shfl_xor(lb, off));
    // candidate list (wave-uniform), the first 64 kept in LDS
#pragma unroll
    for (int ch = 0; ch < CLS_MAXCH; ch++) {
      if (ch * 512 < N) {
        float lo[8], hi[8];
        bounds(ch, lo, hi);
        unsigned mine = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) mine |= hi[e] >= lb ? 1u << e : 0u;
        unsigned long long lanes = __ballot(mine != 0);   // most chunks hold no candidate at all
        while (lanes) {
          const int src = __builtin_ctzll(lanes);
          lanes &= lanes - 1;
          unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mine, src);
          while (bits) {
            const int e = __builtin_ctz(bits);
            bits &= bits - 1;
            if (count < 64 && lane == 0) cand[count] = ch * 512 + src * 8 + e;
            count++;
          }
        }
      }
    }
  } else {
    // fp16 logits: the same candidate set for a tenth of the arithmetic.  With E = the bound's class-dependent part at
    // the LARGEST weight norm, lo_c >= g(x_c) - E and hi_c <= f(x_c) + E for g(x) = x - k2 |x|, f(x) = x + k2 |x| (both
    // increasing), so  lb' = g(max x) - E  is a lower bound of max lo, and a group of eight logits whose packed maximum m
    // has f(m) + E < lb' holds no candidate: one v_pk_max_f16 tree and one compare per group instead of eight bounds.
    // The few groups that pass get the exact bounds; the class that attains max lo is among them (its hi >= lo >= lb'),
    // so lb = the largest lo seen is the exact one, and filtering the list by hi >= lb gives the set of the bf16 form.
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    float hm[CLS_MAXCH];
    float xm = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < CLS_MAXCH; ch++) {
      hm[ch] = -INFINITY;
      if (ch * 512 < N) {
        hm[ch] = (float)__builtin_reduce_max(__builtin_bit_cast(h8, v[ch]));
        xm = fmaxf(xm, hm[ch]);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) xm = fmaxf(xm, __shfl_xor(xm, off));
    const float E = an * wnmax + (aabs + wabs * wnmax);
    const float lbq = xm - k2 * __builtin_fabsf(xm) - E;
    float lbt = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < CLS_MAXCH; ch++) {
      if (ch * 512 < N) {
        const bool flag = hm[ch] + k2 * __builtin_fabsf(hm[ch]) + E >= lbq;
        unsigned long long lanes = __ballot(flag);
        if (lanes) {
          float lo[8], hi[8];
          bounds(ch, lo, hi);
          unsigned mine = 0;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            if (flag && hi[e] >= lbq) { mine |= 1u << e; lbt = fmaxf(lbt, lo[e]); }
          }
          lanes = __ballot(mine != 0);
          while (lanes) {
            const int src = __builtin_ctzll(lanes);
            lanes &= lanes - 1;
            unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mine, src);
            while (bits) {
              const int e = __builtin_ctz(bits);
              bits &= bits - 1;
              float hv = hi[0];
#pragma unroll
              for (int q = 1; q < 8; q++) hv = e == q ? hi[q] : hv;
              hv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), src));
              if (count < 64 && lane == 0) { cand[count] = ch * 512 + src * 8 + e; chi[count] = hv; }
              count++;
            }
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lbt = fmaxf(lbt, __shfl_xor(lbt, off));
    lb = lbt;
    if (count <= 64) {   // keep the entries with hi >= lb (lane 0 wrote the list; a longer list is rescanned below with lb)
      int kept = 0;
      if (lane == 0) {
        for (int i = 0; i < count; i++)
          if (chi[i] >= lb) { cand[kept] = cand[i]; kept++; }
      }
      count = __builtin_amdgcn_readfirstlane(kept);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // lane 0's list entries before the other lanes read them
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (count <= 1) {   // the bound already separates the best class from all others
    if (lane == 0) idx[row] = count == 1 ? cand[0] : 0;
    return;
  }
  // several candidates: exact fp32 scores, ties to the lowest index.  The weight rows of up to eight candidates
  // are fetched together (the fetch latency, not the arithmetic, is what a candidate costs).
  const float* a = A + (long)row * K;
  float best = -INFINITY;
  int bi = 0x7FFFFFFF;
  auto consider = [&](float part, int cc) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const float s = part + bias[cc];
    if (s > best || (s == best && cc < bi)) { best = s; bi = cc; }
  };
  if (K == 512 && count <= 64) {
    const f32x4 a0 = *(const f32x4*)(a + lane * 4), a1 = *(const f32x4*)(a + 256 + lane * 4);
    for (int base = 0; base < count; base += 8) {
      f32x4 w0[8], w1[8];
      int cc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        cc[u] = base + u < count ? cand[base + u] : -1;
        const float* w = W + (long)(cc[u] < 0 ? 0 : cc[u]) * K + lane * 4;
        w0[u] = *(const f32x4*)w;
        w1[u] = *(const f32x4*)(w + 256);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (cc[u] < 0) continue;
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) part = __builtin_fmaf(a0[e], w0[u][e], part);
#pragma unroll
        for (int e = 0; e < 4; e++) part = __builtin_fmaf(a1[e], w1[u][e], part);
        consider(part, cc[u]);
      }
    }
  } else {   // general K, or more candidates than the list holds: every class of the row that qualifies, one at a time
    const uint16_t* ls = L + (long)row * ldl;
    for (int c0 = 0; c0 < N; c0 += 64) {
      const int c = c0 + lane;
      const float x = c < N ? (F16 ? (float)__builtin_bit_cast(_Float16, ls[c]) : __builtin_bit_cast(float, (unsigned)ls[c] << 16)) : -INFINITY;
      const float wn = wnorm[c < N ? c : 0];
      unsigned long long m = __ballot(c < N && x + (an * wn + k2 * __builtin_fabsf(x) + (aabs + wabs * wn)) >= lb);
      while (m) {
        const int cc = c0 + __builtin_ctzll(m);
        m &= m - 1;
        const float* w = W + (long)cc * K;
        float part = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
          const f32x4 av = *(const f32x4*)(a + k), wv = *(const f32x4*)(w + k);
#pragma unroll
          for (int e = 0; e < 4; e++) part = __builtin_fmaf(av[e], wv[e], part);
        }
        consider(part, cc);
      }
    }
  }
  if (lane == 0) idx[row] = bi == 0x7FFFFFFF ? 0 : bi;
}

void launch_cls_select(const uint16_t* L, const float* A, const float* W, const float* bias, const float* anorm,
                       const float* wnorm, int32_t* idx, int M, int N, int K, int ldl, hipStream_t s, int f16, float wnmax) {
  if (f16) hipLaunchKernelGGL(cls_select_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, L, A, W, bias, anorm, wnorm, idx, M, N, K, ldl, wnmax);
  else hipLaunchKernelGGL(cls_select_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, L, A, W, bias, anorm, wnorm, idx, M, N, K, ldl, wnmax);
}

// One empty launch from this translation unit: the runtime loads a code object (and sets a stream's queue up) at the
// first launch that needs it; crnn_api.hip does these at device set-up so that no later call is the first one.
__global__ void warm_nopk_kernel() {}
void launch_warm_nopk(hipStream_t s) { hipLaunchKernelGGL(warm_nopk_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace crnn
