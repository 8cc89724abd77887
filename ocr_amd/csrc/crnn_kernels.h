// Internal launch interface between crnn_api.hip (host side of the C-ABI)
// and crnn_kernels.hip (gfx950 kernels).  Not part of the public boundary;
// the public boundary is include/crnn_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Kernels that need more than 256 registers per lane (the compiler then keeps part of them in AccVGPRs) claim the WHOLE
// register file of their SIMD: a wave of another kernel that shares the SIMD with such a wave makes its results go wrong
// (round 3, DESIGN.md section 4: found on wino4_out_in_h3_kernel<0> in the pipelined path -- single 4-byte words of V in
// lanes 12-15 of a row of 16 carried a stale register's contents, once in ~50 batches of 256 lines; gone with this line,
// 0 in 60 000 batches against 700-800 in 20 000 in the same box).  Costs nothing: no instruction, no register is touched.
#define CRNN_WHOLE_REGISTER_FILE() asm volatile("" ::: "v255", "a255")

namespace crnn {

// Implicit-GEMM convolution / plain GEMM on the f32 MFMA:
//   C[m][n] = act( sum_k A_im2col[m][k] * Wt[n][k] + bias[n] ),  m = (b, y, x) of the OUTPUT
// A is NHWC [B,Hin,Win,Cin]; K = ks*ks*Cin ordered (ky,kx,ci); a plain GEMM is
// the 1x1 case with Hin=Win=Hout=Wout=1, Cin=K.
// Blocks of a persistent GEMM launch (a multiple of 8 for the XCD-aware unit order): 256, one per CU, when the caller has
// the device to itself.  Launches of the pipelined device path (several lanes in flight) use fewer (192 by default):
// a launch that claims every CU makes the kernels of the other lanes -- the 32-CU recurrences in particular -- and its own
// late blocks queue behind one another; measured 96 k -> 100 k lines/s (bench.py, 4 lanes) for 160-192 blocks, although
// each GEMM alone on the device is 15 % slower with 192.  set_persistent_blocks is per host thread (the API sets it
// before the launches of a network run); CRNN_PERSISTENT_BLOCKS overrides both.
inline thread_local int g_persistent_blocks = 256;
inline void set_persistent_blocks(int g) {
  static const int forced = [] {
    const char* e = getenv("CRNN_PERSISTENT_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v >= 8 && v <= 1024 ? v / 8 * 8 : 0;
  }();
  g_persistent_blocks = forced ? forced : (g >= 8 && g <= 1024 ? g / 8 * 8 : 256);
}
inline int persistent_blocks() { return g_persistent_blocks; }
struct ConvGemm {
  const float* A;
  const float* Wt;    // [N][K]
  const float* bias;  // [N]
  float* C;           // [M][ldc]
  const int32_t* wvalid;  // per-sample valid output width (x >= wvalid[b] -> 0) or nullptr
  int M, N, K, ldc;
  int Hin, Win, Cin, Hout, Wout, ks, pad;
  int act;            // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
  // fused arg-max epilogue (classifier): when pmax != nullptr every block writes, for each of its
  // rows, the best value / lowest best column over ITS 128 columns to pmax/pidx[row][n-tile];
  // C may then be nullptr (no logits written).  launch_argmax_partials finishes the reduction.
  float* pmax;
  int32_t* pidx;
  // batched GEMM (Winograd: 16 independent GEMMs): blockIdx.y selects the batch; element strides
  int nbatch;         // 0 or 1 = not batched
  long bsA, bsW, bsC;
  int gn;             // lin_gemm_kernel: column groups of the XCD split (set by launch_lin_gemm)
  // f16x3 form (lin_gemm_h3_kernel; no arg-max): A and Wt hold (hi, lo) fp16 pairs (crnn_kernels.hip), rs[row / 32] = 1 / the
  // scale of that block of rows (launch_rows_to_f16x3), us = 1 / the weights' scale (to_f16x3_host)
  int h3;
  const float* rs;
  float us;
};

void launch_conv_gemm(const ConvGemm& p, hipStream_t s);
// Plain GEMMs (ks == 1) on the persistent direct-to-LDS kernel; with pmax set the arg-max partials are
// [M][4 * ceil(N/128)] (one per 32-column group).  lin_gemm_ok: shape fits that kernel.
// Plain GEMM with 2-byte operands (fp16 / bf16) on the direct-to-LDS pipeline, 256 x 256 units (crnn_kernels_bf16.hip):
// the classifier pre-filter and the bf16 path's linear layers.  f16: 1 fp16 operands, 0 bf16; out: 0 fp32, 1 bf16, 2 fp16
bool gemm16_dma_ok(const ConvGemm& p);
void launch_gemm16_dma(const ConvGemm& p, int f16, int out, hipStream_t s);
bool lin_gemm_ok(const ConvGemm& p);
void launch_lin_gemm(const ConvGemm& p, hipStream_t s);

#ifdef __HIPCC__
// Arg-max epilogue shared by the fp32 and bf16 GEMM kernels.  acc is the wave's 2x2 set of 32x32
// MFMA accumulators (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  Per row: max
// over the lane's two columns, butterfly max over the 32 lanes of the half-wave, then two ballots
// find the lowest column holding the max (torch's tie rule).  The two N-waves of the block meet
// in LDS (sv/si: 2 x 128 entries each).  Must be called by all 256 threads after a barrier that
// ends the last LDS reads of the main loop.
typedef float crnn_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void argmax_epilogue(const crnn_f32x16 (&acc)[2][2], const ConvGemm& p, float* sv,
                                                int* si, int m0, int n0, int bn, int ntn, int wm, int wn,
                                                int lane, int tid) {
  const int half = lane >> 5;
  const int c0 = n0 + wn * 64 + (lane & 31), c1 = c0 + 32;
  const bool ok0 = c0 < p.N, ok1 = c1 < p.N;
  const float b0 = ok0 ? p.bias[c0] : 0.f, b1 = ok1 ? p.bias[c1] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float v0 = ok0 ? acc[mt][0][r] + b0 : -INFINITY;
      const float v1 = ok1 ? acc[mt][1][r] + b1 : -INFINITY;
      float m = v0 > v1 ? v0 : v1;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = o > m ? o : m;
      }
      const unsigned long long q0 = __ballot(v0 == m), q1 = __ballot(v1 == m);
      const unsigned h0 = half ? (unsigned)(q0 >> 32) : (unsigned)q0;
      const unsigned h1 = half ? (unsigned)(q1 >> 32) : (unsigned)q1;
      const int col = h0 ? __builtin_ctz(h0) : (h1 ? 32 + __builtin_ctz(h1) : 0);
      if ((lane & 31) == 0) {
        const int rl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        sv[wn * 128 + rl] = m;
        si[wn * 128 + rl] = n0 + wn * 64 + col;
      }
    }
  __syncthreads();
  if (tid < 128 && m0 + tid < p.M) {
    const float a = sv[tid], b = sv[128 + tid];
    const bool takeb = b > a;   // tie -> the lower columns (wave wn = 0)
    p.pmax[(long)(m0 + tid) * ntn + bn] = takeb ? b : a;
    p.pidx[(long)(m0 + tid) * ntn + bn] = takeb ? si[128 + tid] : si[tid];
  }
}

// Store epilogue shared by the fp32 and bf16 GEMM kernels: bias + activation (+ per-line width
// mask) for the wave's 2x2 accumulator tiles.  The activation is a pair of selects, the row guard
// and the mask are computed once per row, and the masked variant is a separate (wave-uniform) path.
template <int OUT_BF16>   // 0: fp32 output, 1: bf16, 2: fp16
__device__ __forceinline__ void store_epilogue(const crnn_f32x16 (&acc)[2][2], const ConvGemm& p, float* Cb, int m0,
                                               int n0, int wm, int wn, int lane) {
  const int act = p.act;
  auto activate = [&](float v) {   // same selects as torch: x > 0 ? x : (0 | 0.2 x); act 0 = identity
    const float neg = act == 2 ? 0.2f * v : (act == 1 ? 0.f : v);
    return v > 0.f ? v : neg;
  };
  const int c0 = n0 + wn * 64 + (lane & 31), c1 = c0 + 32;
  const bool ok0 = c0 < p.N, ok1 = c1 < p.N;
  const float b0 = ok0 ? p.bias[c0] : 0.f, b1 = ok1 ? p.bias[c1] : 0.f;
  const int hw = p.Hout * p.Wout;
  const int rbase = m0 + wm * 64 + 4 * (lane >> 5);
  auto put = [&](long off, float v, bool ok) {
    if (!ok) return;
    if (OUT_BF16 == 2) ((_Float16*)Cb)[off] = (_Float16)v;
    else if (OUT_BF16) ((__bf16*)Cb)[off] = (__bf16)v;
    else Cb[off] = v;
  };
  if (!p.wvalid) {
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = rbase + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool rok = row < p.M;
        const long off = (long)row * p.ldc;
        float v0 = acc[mt][0][r] + b0, v1 = acc[mt][1][r] + b1;
        v0 = activate(v0);
        v1 = activate(v1);
        put(off + c0, v0, rok && ok0);
        put(off + c1, v1, rok && ok1);
      }
  } else {
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = rbase + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool rok = row < p.M;
        const int rr = rok ? row : 0;
        const int b = rr / hw;
        const int x = (rr - b * hw) % p.Wout;
        const bool keep = x < p.wvalid[b];
        const long off = (long)row * p.ldc;
        float v0 = acc[mt][0][r] + b0, v1 = acc[mt][1][r] + b1;
        v0 = keep ? activate(v0) : 0.f;
        v1 = keep ? activate(v1) : 0.f;
        put(off + c0, v0, rok && ok0);
        put(off + c1, v1, rok && ok1);
      }
  }
}
#endif

// Winograd F(4x4,3x3) (fp32): 36 components per 4x4 output tile, 2.25 multiplies per output and input
// channel instead of 9 (direct) or 4 (F(2x2)).  Three kernels: input transform x -> V[36][ntile_pad][Cin],
// 36 GEMMs V U^T -> M[36][ntile_pad][Cout] in one persistent launch, output transform (+ bias, activation,
// width mask, optionally MaxPool 2x2/2) M -> NHWC.  H % 4 == 0; ntile_pad = wino4_pad(tiles)
// (the padding rows of V are never written and their products never read).
struct Wino4Gemm {
  const float* V;   // [36][ntile_pad][Cin]
  const float* U;   // [36][N][Cin]
  float* M;         // [36][ntile_pad][N]
  int ntile_pad, N, Cin;
  int ncomp;        // number of component GEMMs: 36 (F(4x4,3x3)); 5 for conv6's 1-D F(4,2)
  int h3;           // operands in the f16x3 format (crnn_kernels.hip): V and U hold (hi, lo) fp16 pairs, M leaves scaled
  // plain-GEMM form of the ping-pong kernel (launch_lin_gemm_pp; ncomp = 1): M = V U^T * lin_rs[row / 32] * lin_us + lin_bias[col]
  const float* lin_bias;
  const float* lin_rs;
  float lin_us;
};
bool wino4_gemm_uses_pp(const Wino4Gemm& p);   // this launch runs wino4_gemm_h3_pp_kernel (else the loader-wave / plain form)
// a plain f16x3 GEMM (A rows from rows_to_f16x3_kernel, W from to_f16x3_host) on wino4_gemm_h3_pp_kernel: rows % 256 == 0,
// rows >= 1024, N % 128 == 0, K % 64 == 0, K >= 256
bool lin_gemm_pp_ok(int rows, int N, int K);
void launch_lin_gemm_pp(const Wino4Gemm& p, hipStream_t s);
// conv0 + activation + pool0 + conv1's F(4x4) input transform (f16x3 V, per-tile scales) in one kernel; in: [B,32,W] fp32
void launch_conv0_win4(const float* in, const float* w9x64, const float* bias, float* V, float* vs,
                       const int32_t* w1valid, int B, int W, int act, hipStream_t s, const uint16_t* w3 = nullptr,
                       float us3 = 1.f);   // w3: conv0's weights as f16x3 [64][16] taps -> conv0 on the fp16 MFMA
// F(4x4,3x3) GEMMs + output transform in one kernel (wino4_fused_h3_kernel; f16x3 operands only)
struct Wino4Fused {
  const float* V;      // [36][ntile_pad][Cin], f16x3
  const float* U;      // [36][N][Cin], f16x3
  const float* vs;     // [ntile]: 1 / tile scale
  const float* bias;   // [N]
  float* y;            // NHWC; pool 0: [B,H,W,N], 1: [B,H/2,W/2,N] (MaxPool 2x2/2 inside the tile)
  const int32_t* wvalid;
  float us;            // 1 / weight scale
  int B, H, W, Cin, N, act, pool, ntile_pad, ntile;
};
void launch_wino4_fused(const Wino4Fused& p, hipStream_t s);
inline long wino4_tiles(int B, int H, int W) { return (long)B * (H / 4) * ((W + 3) / 4); }
// rows of V / M: tiles rounded up to the GEMM's unit height -- 256 where the ping-pong GEMM runs (from 1024 tiles), else 128
inline long wino4_pad(long tiles) { return tiles > 896 ? (tiles + 255) / 256 * 256 : (tiles + 127) / 128 * 128; }
// vs != nullptr: V in the f16x3 format, vs[tile] = 1 / (the tile's scale)
void launch_wino4_in(const float* x, float* V, int B, int H, int W, int C, hipStream_t s, float* vs = nullptr);
// conv6 as 1-D F(4,2): x NHWC [B,2,W,C] -> V[5][pad(B*ceil((W-1)/4))][2*C]; M[5][..][Cout] -> y [B*(W-1)][Cout] + bias, act
void launch_wino42_in(const float* x, float* V, int B, int W, int C, hipStream_t s, float* vs = nullptr);
void launch_wino42_out(const float* M, const float* bias, float* y, int B, int T, int Cout, int act, hipStream_t s,
                       const float* vs = nullptr, float us = 1.f);
void launch_wino4_gemm(const Wino4Gemm& p, hipStream_t s);
// pool: 0 none; 1 MaxPool 2x2/2 (y [B,H/2,W/2,Cout]); 2 MaxPool (2,2)/(2,1)/pad (0,1) (y [B,H/2,W+1,Cout])
// vs / us: unscale factors of f16x3 products (per tile, per layer) or nullptr / 1
void launch_wino4_out(const float* M, const float* bias, float* y, const int32_t* wvalid, int B, int H, int W,
                      int Cout, int act, int pool, hipStream_t s, const float* vs = nullptr, float us = 1.f);
// layer L's output transform + layer L + 1's input transform (f16x3 form) in one kernel: M of layer L -> V and vs_out of
// layer L + 1.  src 0: H = 4, layer L + 1 reads the map as it is; src 1: H = 8, MaxPool (2,2)/(2,1)/pad (0,1) in between
bool wino4_out_in_ok(int B, int H, int W, int C, int src);
void launch_wino4_out_in(const float* M, const float* bias, const float* vs_in, float us, const int32_t* wvalid, float* V,
                         float* vs_out, int B, int H, int W, int C, int act, int src, hipStream_t s);
constexpr int F16X3_ROWS = 32;   // rows per scale of a plain GEMM's f16x3 A operand
// conv6's output transform writing the f16x3 rows (and scales) of the first LSTM projection's A operand directly; y (the
// fp32 rows) only if not null.  Cout = 512.
void launch_wino42_out_split(const float* M, const float* bias, float* y, float* a3, float* rs, int B, int T, int act,
                             hipStream_t s, const float* vs, float us);
// fp32 rows [M][K] (K = 256 or 512; false for any other K) -> f16x3 rows + one unscale factor per block of 32 rows (rs holds 4 per 128-row GEMM unit, whole units)
bool launch_rows_to_f16x3(const float* A, float* out, float* rs, int M, int K, hipStream_t s);
// host: rows of K floats -> the f16x3 format with ONE scale for the whole tensor; returns 1 / scale
float to_f16x3_host(const float* src, size_t n, uint16_t* dst /* 2 n halves */);

// conv0 (1->64, 3x3, pad 1) + ReLU + MaxPool 2x2/2, direct (K = 9, HBM-bound).
// in f32 [B,32,W]; out NHWC [B,16,W/2,64]; w1valid: per-sample valid pooled width or nullptr.
void launch_conv0_pool(const float* in, const float* w9x64, const float* bias, float* out,
                       const int32_t* w1valid, int B, int W, int act, hipStream_t s);

// NHWC max pools.  pool_2x2s2: [B,H,W,C] -> [B,H/2,W/2,C].
// pool_2x2s21p01: kernel (2,2) stride (2,1) pad (0,1): [B,H,W,C] -> [B,H/2,W+1,C].
void launch_pool_2x2s2(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
void launch_pool_2x2s21p01(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);

// LSTM recurrence, both directions.  xp [B*T][2*4*nh] (input projection with
// b_ih+b_hh already added; col = dir*4nh + gate*nh + unit), whh_packed as
// produced by pack_whh(), rec [B*T][2*nh] (col = dir*nh + unit).  nh == 256.
void launch_lstm_rec(const float* xp, const float* whh_packed, float* rec, const int32_t* tlen,
                     int B, int T, hipStream_t s);
// Latency form for small batches: four workgroups per (direction, 16 lines), W_hh slices resident in
// registers, h exchanged every step through rec itself (write-through stores; the rows double as arrival flags).
// tmo: one device word the caller has zeroed, set to 1 if a hand-off timed out (the caller then repeats the
// recurrence with launch_lstm_rec).
// the same on the fp16 MFMA in the f16x3 form (weights from pack_whh_h3, unscale as for launch_lstm_rec_h3): bit-identical to
// lstm_rec_h3_kernel
void launch_lstm_rec_split_h3(const float* xp, const void* whh3, float* rec, const int32_t* tlen, int B, int T,
                              float unscale, unsigned* tmo, hipStream_t s, int nt = 1);
// f16x3 form of the recurrence (crnn_kernels.hip): whh3 from pack_whh_h3 (2 * 1024 * 256 * 2 halves), unscale its return value
// empty launches, one per translation unit of kernels (device set-up: code objects loaded, stream queues created)
void launch_warm_main(hipStream_t s);
void launch_warm_bf16(hipStream_t s);
void launch_warm_nopk(hipStream_t s);
void launch_lstm_rec_h3(const float* xp, const void* whh3, float* rec, const int32_t* tlen, int B, int T, float unscale,
                        hipStream_t s, const unsigned* gate = nullptr);
float pack_whh_h3(const float* whh_fwd, const float* whh_rev, uint16_t* packed);
// host-side repack of W_hh [2][1024][256] into the per-lane fragment order the kernel streams.
void pack_whh(const float* whh_fwd, const float* whh_rev, float* packed /* 2*1024*256 */);

// second stage of the fused classifier arg-max: pmax/pidx [M][ntiles] -> idx [M] (ties -> lowest index)
void launch_argmax_partials(const float* pmax, const int32_t* pidx, int32_t* idx, int M, int ntiles,
                            hipStream_t s);

// row-wise argmax over nclass (ties -> lowest index); logits [M][ld] -> idx [M]
// greedy CTC collapse (util.py:16-22): idx [B][T] -> out [B][T] alphabet positions (class-1), len [B]
void launch_ctc(const int32_t* idx, const int32_t* tlen, int32_t* out, int32_t* len, int B, int T,
                hipStream_t s);

// preprocessing ------------------------------------------------------------
// 256-entry table (x/255-0.5)/0.5 computed on the host in float64 like util.py:12
void upload_norm_lut(const float* lut256, hipStream_t s);
// u8 gray [B,32,W] (already at height 32, width W) -> normalised f32 [B,32,Wp], zero beyond w[b]
void launch_normalize(const uint8_t* gray, float* out, const int32_t* w, int B, int W, int Wp,
                      hipStream_t s);

struct ResizeJob {        // one source image (device resident)
  int64_t src_off;        // byte offset into the packed source buffer
  int32_t sh, sw, stride, ch;   // source height/width/row stride in bytes/channels (1 gray, 3 BGR)
  int32_t dw;             // resized width (util.py:7-9)
  int32_t ksh, ksv;       // taps per output column / row (a pass over an unchanged dimension
                          // has coefficients {1,0}: identical to Pillow skipping it)
  int32_t ybox_first, th; // first source row the vertical pass needs, rows of the temp image
  int64_t tmp_off;        // byte offset of this image's temp (horizontal pass output) [th][dw]
  int64_t coef_off;       // int32 offset of this image's coefficient block:
                          //   bounds_h[2*dw], kk_h[dw*ksh], bounds_v[64], kk_v[32*ksv]
};
// gray + Pillow BILINEAR horizontal pass -> tmp u8
void launch_resize_h(const uint8_t* src, const ResizeJob* jobs, const int32_t* coef, uint8_t* tmp,
                     int n, int max_th, int max_dw, int max_sw, hipStream_t s);
// vertical pass + normalise -> f32 [n,32,Wp] (zero beyond dw)
void launch_resize_v(const uint8_t* src, const ResizeJob* jobs, const int32_t* coef, const uint8_t* tmp,
                     float* out, int n, int Wp, hipStream_t s);

// bf16 path (crnn_kernels_bf16.hip): bf16 activations and weights, fp32 accumulation ----------
// ConvGemm is reused with A / Wt (and C when out_bf16) pointing at bf16 data; K % 64 == 0.
// out_mode: 0 fp32 output, 1 bf16 output; 2: A, Wt and the output are fp16 (the classifier pre-filter)
void launch_conv_gemm_bf16(const ConvGemm& p, int out_mode, hipStream_t s);
// Certified bf16 pre-filter of the classifier arg-max (see crnn_kernels.hip): A fp32 [M][K] -> bf16 copy + row
// norms; then, from the bf16 GEMM's logits L [M][N] (bf16), idx[row] = exact fp32 arg-max over the certified candidates.
// f16 != 0: fp16 instead of bf16 (8x tighter bound; needs |A|, |W| below the fp16 maximum: the LSTM head, |h| < 1)
void launch_rows_to_bf16(const float* A, uint16_t* out, float* norm, int M, int K, hipStream_t s, int f16 = 0);
void launch_cls_select(const uint16_t* L, const float* A, const float* W, const float* bias, const float* anorm,
                       const float* wnorm, int32_t* idx, int M, int N, int K, int ldl, hipStream_t s, int f16 = 0, float wnmax = 0.f);   // ldl % 8 == 0, N <= 8192
uint16_t f32_to_f16_host(float f);
void launch_conv0_pool_bf16(const float* in, const float* w9x64, const float* bias, void* out,
                            const int32_t* w1valid, int B, int W, int act, hipStream_t s);
void launch_pool_2x2s2_bf16(const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
void launch_pool_2x2s21p01_bf16(const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
// xp fp32 [B*T][2048]; whh_packed from pack_whh_bf16; rec bf16 [B*T][512]
void launch_lstm_rec_bf16(const float* xp, const void* whh_packed, void* rec, const int32_t* tlen, int B,
                          int T, hipStream_t s);
uint16_t f32_to_bf16_host(float f);
void pack_whh_bf16(const float* whh_fwd, const float* whh_rev, uint16_t* packed /* 2*1024*256 */);

}  // namespace crnn
