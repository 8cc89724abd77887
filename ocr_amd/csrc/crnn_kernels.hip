// gfx950 (MI355X / CDNA4) kernels of the CRNN text-line recogniser.
//
// Replaces, on the GPU, the torch CPU operators the reference runs in
// crnn/network_torch.py:44-108 (conv/BN/ReLU/pool backbone, two BiLSTMs,
// classifier), :144 (argmax) and crnn/util.py:6-22 (resize/normalise, greedy
// CTC collapse).  fp32 throughout: matrix work goes through the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, a k-ordered fmaf
// chain), byte/integer work is plain coalesced vector code.
#include "crnn_kernels.h"
#include "crnn_split.h"

#include <cmath>
#include <cstring>

namespace crnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution / GEMM.  128x128 output tile per 256-thread block,
// BK = 32, 4 waves as 2(M) x 2(N), each wave 2x2 MFMA tiles of 32x32.
// LDS tiles are [row][k] with rows padded to 36 floats so that the 16-lane
// groups of a ds_read_b128 hit 16 distinct 4-bank slots (conflict free).
// One k-group = 8 consecutive k: lane half h holds k = 8g+4h .. 8g+4h+3 of its
// row as one float4, and MFMA j of the group consumes element j of both
// operands (k pair {8g+j, 8g+4+j}); A and B use the same permutation, so the
// sum over k is complete.
//
// Staging: ONE LDS buffer (36 KB) + register prefetch, so 4 blocks (16 waves)
// share a CU and fill each other's barrier / staging gaps on the MFMA pipe
// (measured: 4 blocks x single buffer beats 2 blocks x double buffer by 10 %).
// Loads are raw buffer loads: the per-lane byte offset of a row is constant
// over the K loop, the K position travels in the scalar offset, and a padding
// tap is an out-of-range offset (the hardware returns zeros), so the loader
// costs ~1 VALU per load and the loop body is one basic block.
// ---------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 3) void conv_gemm_kernel(const ConvGemm p) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bm = blockIdx.x / ntn, bn = blockIdx.x - bm * ntn;
  const int m0 = bm * BM, n0 = bn * BN;

  const int lrow = tid >> 3;
  const int lk = (tid & 7) * 4;
  const int hw = p.Hout * p.Wout;

  // Descriptors.  A's base is moved back by the padding so that every row offset is >= 0;
  // the tap displacement is added through the scalar offset.  Rows beyond M / N are clamped to
  // the last valid row (their results are never stored).
  const long shift = ((long)p.pad * p.Win + p.pad) * p.Cin;
  const float* Ab = p.A + (long)blockIdx.y * p.bsA;      // batched GEMM: blockIdx.y = batch
  const float* Wb = p.Wt + (long)blockIdx.y * p.bsW;
  float* Cb = p.C ? p.C + (long)blockIdx.y * p.bsC : nullptr;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Ab - shift), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7FFFFFFF, 0x00020000);
  unsigned aoff[4], amask[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = m0 + lrow + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int b = m / hw, r = m - b * hw;
    const int y = r / p.Wout, x = r - y * p.Wout;
    aoff[i] = (unsigned)(((((long)b * p.Hin + y) * p.Win + x) * p.Cin + lk) * 4);
    unsigned mk = 0;   // bit (ky*ks+kx): tap inside the image
    for (int ky = 0; ky < p.ks; ky++)
      for (int kx = 0; kx < p.ks; kx++) {
        const int yy = y - p.pad + ky, xx = x - p.pad + kx;
        if ((unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win) mk |= 1u << (ky * p.ks + kx);
      }
    amask[i] = mk;
    int n = n0 + lrow + 32 * i;
    n = n < p.N ? n : p.N - 1;
    boff[i] = (unsigned)(((long)n * p.K + lk) * 4);
  }
  const int nk = p.K / BK;
  f32x4 ra[4], rb[4];
  // wave-uniform position of the NEXT tile to load: tap (ky,kx) and channel offset ci
  int l_tap = 0, l_ky = 0, l_kx = 0, l_ci = 0, l_k0 = 0;

  auto gload = [&]() {
    const int toff = ((l_ky * p.Win + l_kx) * p.Cin + l_ci) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned vo = ((amask[i] >> l_tap) & 1u) ? aoff[i] : 0x80000000u;  // out of range -> zeros
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, vo, toff, 0));
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, boff[i], l_k0 * 4, 0));
    }
    // next K-tile: taps cycle fastest, then the channel chunk (weights are packed in this order)
    l_k0 += BK;
    l_tap++;
    l_kx++;
    const bool wrap_x = l_kx == p.ks;
    l_kx = wrap_x ? 0 : l_kx;
    l_ky += wrap_x ? 1 : 0;
    const bool wrap_t = l_ky == p.ks;
    l_ky = wrap_t ? 0 : l_ky;
    l_tap = wrap_t ? 0 : l_tap;
    l_ci += wrap_t ? BK : 0;
  };
  auto sstore = [&]() {
    float* sa = smem;
    float* sb = sa + BM * LDT;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *(f32x4*)(sa + (lrow + 32 * i) * LDT + lk) = ra[i];
      *(f32x4*)(sb + (lrow + 32 * i) * LDT + lk) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  auto compute = [&]() {
    const float* a = smem + (wm * 64 + (lane & 31)) * LDT + 4 * (lane >> 5);
    const float* b = smem + BM * LDT + (wn * 64 + (lane & 31)) * LDT + 4 * (lane >> 5);
#pragma unroll
    for (int kg = 0; kg < 4; kg++) {
      f32x4 a0 = *(const f32x4*)(a + kg * 8);
      f32x4 a1 = *(const f32x4*)(a + 32 * LDT + kg * 8);
      f32x4 b0 = *(const f32x4*)(b + kg * 8);
      f32x4 b1 = *(const f32x4*)(b + 32 * LDT + kg * 8);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
      }
    }
  };

  gload();
  sstore();
  __syncthreads();
  for (int kt = 0; kt < nk - 1; kt++) {
    gload();            // tile kt+1 -> registers, in flight under the MFMAs of tile kt
    compute();
    __syncthreads();    // every wave has read tile kt
    sstore();
    __syncthreads();    // tile kt+1 visible
  }
  compute();

  if (p.pmax) {   // fused classifier arg-max (wave-uniform branch)
    __syncthreads();
    argmax_epilogue(acc, p, smem, (int*)(smem + 256), m0, n0, bn, ntn, wm, wn, lane, tid);
    if (!Cb) return;
  }

  store_epilogue<false>(acc, p, Cb, m0, n0, wm, wn, lane);
}

void launch_conv_gemm(const ConvGemm& p, hipStream_t s) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(conv_gemm_kernel, dim3(ntm * ntn, p.nbatch > 1 ? p.nbatch : 1), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------
// Operand staging of every GEMM kernel below: direct-to-LDS loads (buffer_load_dwordx4 ... lds): no staging VGPRs and
// no ds_write.  A wave-level load writes 1 KiB contiguously (lane l -> base + 16 l), i.e. eight 128-byte rows back to
// back, so instead of padding rows the 16-byte k-chunks are XOR-swizzled inside 256-byte row pairs: chunk position =
// ((row & 1) * 8 + kchunk) ^ ((row >> 1) & 15).  Each lane simply fetches the global chunk that belongs at its LDS
// position; the 16 lanes of a fragment-read phase (16 consecutive rows, one k-chunk) then hit 16 distinct 16-byte bank
// groups (SQ_LDS_BANK_CONFLICT = 0).  (Round 1's F(2x2,3x3) kernels, where this staging was first built, and the
// register-staged form before it are gone: the F(4x4) kernels below won at every batch size; DESIGN.md, history.)
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

static unsigned ew_blocks_w(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// (wino4_bt / wino4_at and the f16x3 split helpers are in crnn_split.h, shared with crnn_kernels_nopk.hip)

// F(4x4,3x3) input transform writing the f16x3 format: as wino4_in_kernel, plus the tile's scale (the largest |v| over
// its 36 components and all channels: the C/4 threads of a tile are consecutive lanes, or two waves for C = 512) to
// vs[tile] (as 1 / s).  The block-level loop bound is uniform, so every thread reaches the barrier.
__global__ __launch_bounds__(256, 2) void wino4_in_h3_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                             float* __restrict__ vs, int B, int H, int W, int C,
                                                             long ntile_pad) {
  __shared__ float wmax[2][4];
  const int th = H >> 2, tw = (W + 3) >> 2, C4 = C >> 2;
  const long total = (long)B * th * tw * C4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  int par = 0;
  for (long base = (long)blockIdx.x * blockDim.x; base < total; base += (long)gridDim.x * blockDim.x, par ^= 1) {
    const long i = base + threadIdx.x;
    const bool valid = i < total;       // uniform per tile (C4 divides 256 and total)
    const long ii = valid ? i : 0;
    const int c4 = (int)(ii % C4);
    const long t = ii / C4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const int b = (int)(r / th);
    f32x4 u[6][6];
    float mx = 0.f;   // largest |d| of the patch: the tile's scale comes from the bound 49 max|d| on its components
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int xx = 4 * tx - 1 + j;
      const bool okx = (unsigned)xx < (unsigned)W;
      const float* px = x + ((long)b * H * W + (okx ? xx : 0)) * C + c4 * 4;
      f32x4 d[6], c[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int yy = 4 * ty - 1 + k;
        const bool ok = valid && okx && (unsigned)yy < (unsigned)H;
        d[k] = ok ? *(const f32x4*)(px + (long)yy * W * C) : z;
#pragma unroll
        for (int e = 0; e < 4; e++) mx = fmaxf(mx, __builtin_fabsf(d[k][e]));
      }
      wino4_bt(d, c);
#pragma unroll
      for (int k = 0; k < 6; k++) u[k][j] = c[k];
    }
    for (int off = 1; off < C4 && off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (C4 > 64) {   // C = 512: a tile is two waves (wave pairs 0/1 and 2/3 of the block)
      const int wave = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) wmax[par][wave] = mx;
      __syncthreads();
      mx = fmaxf(mx, wmax[par][wave ^ 1]);
    }
    // every component is at most 49 max|d| (the rows of B^T sum to at most 7 in magnitude): with the scale taken from that
    // bound a row of the transform is split and stored as soon as it is computed (no 36-plane live set: two waves per SIMD);
    // the bound is loose by 2-3 bits, which only moves the point below which lo parts go subnormal (the MFMA honours them)
    float sc, uns;
    split_scale_from_max(49.f * mx, sc, uns);
    if (valid && c4 == 0) vs[t] = uns;
    float* vrow = V + t * C;
    const long cstride = ntile_pad * C;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      f32x4 v[6];
      wino4_bt(u[k], v);
#pragma unroll
      for (int j = 0; j < 6; j++) {
        uint2 hi, lo;
        split4(v[j], sc, hi, lo);
        if (valid) store_split_pair(vrow, c4, hi, lo);
        vrow += cstride;
      }
    }
  }
}

// One thread per (tile, 4 channels): 36 float4 loads (the 6x6 patch; neighbours' overlaps are L1/L2 hits),
// columns then rows of B^T d B, 36 float4 stores.
__global__ __launch_bounds__(256) void wino4_in_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                       int B, int H, int W, int C, long ntile_pad) {
  CRNN_WHOLE_REGISTER_FILE();
  const int th = H >> 2, tw = (W + 3) >> 2, C4 = C >> 2;
  const long total = (long)B * th * tw * C4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const int b = (int)(r / th);
    f32x4 u[6][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {   // column j of the patch -> column j of B^T d
      const int xx = 4 * tx - 1 + j;
      const bool okx = (unsigned)xx < (unsigned)W;
      const float* px = x + ((long)b * H * W + (okx ? xx : 0)) * C + c4 * 4;
      f32x4 d[6], c[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int yy = 4 * ty - 1 + k;
        const bool ok = okx && (unsigned)yy < (unsigned)H;
        d[k] = ok ? *(const f32x4*)(px + (long)yy * W * C) : z;
      }
      wino4_bt(d, c);
#pragma unroll
      for (int k = 0; k < 6; k++) u[k][j] = c[k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      f32x4 v[6];
      wino4_bt(u[k], v);
      float* o = V + ((long)(k * 6) * ntile_pad + t) * C + c4 * 4;
#pragma unroll
      for (int j = 0; j < 6; j++) *(f32x4*)(o + (long)j * ntile_pad * C) = v[j];
    }
  }
}

// One thread per (tile, 4 output channels): 36 float4 loads, A^T m A, bias + activation + width mask,
// 16 float4 stores (or the 4 maxima of the 2x2 pool windows inside the tile).
// vs / us (f16x3 GEMMs): the products are scaled by s_tile s_weights; vs[tile] = 1 / s_tile, us = 1 / s_weights (powers of two)
__global__ __launch_bounds__(256) void wino4_out_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                        float* __restrict__ y, const int32_t* __restrict__ wvalid,
                                                        int B, int H, int W, int Co, int act, int pool,
                                                        long ntile_pad, const float* __restrict__ vs, float us) {
  const int th = H >> 2, tw = (W + 3) >> 2, C4 = Co >> 2;
  const long total = (long)B * th * tw * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const int b = (int)(r / th);
    f32x4 s[4][6];   // A^T m
#pragma unroll
    for (int j = 0; j < 6; j++) {
      f32x4 m[6], c[4];
#pragma unroll
      // M is read exactly once: nontemporal loads (0.545 -> 0.481 ms over the six output transforms of a step)
      for (int k = 0; k < 6; k++) m[k] = __builtin_nontemporal_load((const f32x4*)(Mo + ((long)(k * 6 + j) * ntile_pad + t) * Co + c4 * 4));
      wino4_at(m, c);
#pragma unroll
      for (int k = 0; k < 4; k++) s[k][j] = c[k];
    }
    const f32x4 bs = *(const f32x4*)(bias + c4 * 4);
    const int wv = wvalid ? wvalid[b] : W;
    const float rs = vs ? vs[t] * us : 1.f;
    f32x4 o[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wino4_at(s[k], o[k]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int xx = 4 * tx + j;
        f32x4 v = o[k][j] * rs + bs;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float e = v[c];
          if (act == 1) e = e > 0.f ? e : 0.f;
          else if (act == 2) e = e > 0.f ? e : 0.2f * e;
          if (xx >= wv) e = 0.f;
          v[c] = e;
        }
        o[k][j] = v;
      }
    }
    if (pool) {
      const int Wp = W >> 1, Hp = H >> 1;
#pragma unroll
      for (int pi = 0; pi < 2; pi++)
#pragma unroll
        for (int pj = 0; pj < 2; pj++) {
          const int xp = 2 * tx + pj;
          if (xp >= Wp) continue;   // odd W: the last column has no pool window
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; c++)
            v[c] = fmaxf(fmaxf(o[2 * pi][2 * pj][c], o[2 * pi][2 * pj + 1][c]),
                         fmaxf(o[2 * pi + 1][2 * pj][c], o[2 * pi + 1][2 * pj + 1][c]));
          *(f32x4*)(y + (((long)b * Hp + 2 * ty + pi) * Wp + xp) * Co + c4 * 4) = v;
        }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int xx = 4 * tx + j;
          if (xx < W) *(f32x4*)(y + (((long)b * H + 4 * ty + k) * W + xx) * Co + c4 * 4) = o[k][j];
        }
    }
  }
}

// Output transform + MaxPool (2,2)/(2,1)/pad (0,1) (pool2 / pool3) in one kernel: the pool's windows cross tile
// columns, so one thread owns a whole ROW of tiles for 4 channels and walks along x carrying the row-pair maxima of
// the previous tile's last column; the full-resolution conv output is never written.  y: [B, H/2, W+1, Co].
__global__ __launch_bounds__(256) void wino4_out_pool21_kernel(const float* __restrict__ Mo,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               const int32_t* __restrict__ wvalid, int B, int H, int W,
                                                               int Co, int act, long ntile_pad, int nseg,
                                                               const float* __restrict__ vs, float us) {
  CRNN_WHOLE_REGISTER_FILE();
  const int th = H >> 2, tw = (W + 3) >> 2, C4 = Co >> 2, Wp = W + 1, Hp = H >> 1;
  const int seglen = (tw + nseg - 1) / nseg;
  const long total = (long)B * th * nseg * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int seg = (int)(r % nseg);
    r /= nseg;
    const int ty = (int)(r % th);
    const int b = (int)(r / th);
    // a row is cut into nseg segments; a segment after the first recomputes the tile before it for its last column
    const int tx0 = seg * seglen, tx1 = tx0 + seglen < tw ? tx0 + seglen : tw;
    const f32x4 bs = *(const f32x4*)(bias + c4 * 4);
    const int wv = wvalid ? wvalid[b] : W;
    float* yrow = y + ((long)b * Hp + 2 * ty) * Wp * Co + c4 * 4;
    f32x4 prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // column 4 tx - 1, rows (0,1) and (2,3) of the tile row
    for (int tx = tx0 > 0 ? tx0 - 1 : 0; tx < tx1; tx++) {
      const bool emit = tx >= tx0;
      const long t = ((long)b * th + ty) * tw + tx;
      const float rs = vs ? vs[t] * us : 1.f;
      // all 36 loads first (few threads, each walking a row: the memory parallelism has to come from inside a thread)
      f32x4 mm[6][6];
#pragma unroll
      for (int j = 0; j < 6; j++)
#pragma unroll
        for (int k = 0; k < 6; k++)
          mm[j][k] = __builtin_nontemporal_load((const f32x4*)(Mo + ((long)(k * 6 + j) * ntile_pad + t) * Co + c4 * 4));
      f32x4 s[4][6];   // A^T m
#pragma unroll
      for (int j = 0; j < 6; j++) {
        f32x4 c[4];
        wino4_at(mm[j], c);
#pragma unroll
        for (int k = 0; k < 4; k++) s[k][j] = c[k];
      }
      f32x4 cm[2][4];   // max over the two rows of each pool window, per column of the tile
#pragma unroll
      for (int k = 0; k < 4; k++) {
        f32x4 o[4];
        wino4_at(s[k], o);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int xx = 4 * tx + j;
          f32x4 v = o[j] * rs + bs;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            float e = v[c];
            if (act == 1) e = e > 0.f ? e : 0.f;
            else if (act == 2) e = e > 0.f ? e : 0.2f * e;
            if (xx >= wv) e = 0.f;
            v[c] = e;
          }
          if (k & 1) {
#pragma unroll
            for (int c = 0; c < 4; c++) cm[k >> 1][j][c] = fmaxf(cm[k >> 1][j][c], v[c]);
          } else {
            cm[k >> 1][j] = v;
          }
        }
      }
#pragma unroll
      for (int pr = 0; pr < 2; pr++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int xx = 4 * tx + j;            // pooled column xx = max over conv columns xx - 1 and xx
          if (xx > W || !emit) continue;
          f32x4 v;
          const f32x4 left = j == 0 ? prev[pr] : cm[pr][j - 1];
          if (xx == W) v = left;                // right edge: column W - 1 alone
          else if (xx == 0) v = cm[pr][0];      // left edge: column 0 alone
          else {
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = fmaxf(left[c], cm[pr][j][c]);
          }
          *(f32x4*)(yrow + ((long)pr * Wp + xx) * Co) = v;
        }
      prev[0] = cm[0][3];
      prev[1] = cm[1][3];
    }
    if ((W & 3) == 0 && tx1 == tw && tx0 < tw) {   // W a multiple of 4: the last pooled column lies past the last tile
      *(f32x4*)(yrow + (long)W * Co) = prev[0];
      *(f32x4*)(yrow + ((long)Wp + W) * Co) = prev[1];
    }
  }
}

// ---------------------------------------------------------------------------
// Output transform of layer L AND input transform of layer L + 1 in one kernel (f16x3 path: conv3 -> pool2 -> conv4 and
// conv4 -> conv5): the activation between the two layers, written and read back by the two-kernel form, only exists
// in registers.  The next layer's tile scale is a maximum over ALL channels of the tile, so a workgroup is the C / 4
// threads of all channels (one or two waves); it owns a segment of one line's row of tiles and walks along x.  Step tx
// computes the 4 x 4 tile tx of the next layer's input map -- SRC 0: A^T m A of tile tx (the map is 4 rows high);
// SRC 1: of the two tiles above each other, max-pooled (2,2)/(2,1)/pad (0,1) as wino4_out_pool21_kernel does (8 rows ->
// 4, W -> W + 1 columns) -- after which the 6 x 6 patch of the next layer's tile tx - 1 is complete: column 3 of tile
// tx - 2, tile tx - 1, column 0 of tile tx, rows -1 and 4 the zero padding.  It goes through the arithmetic of
// wino4_in_h3_kernel: the same operations on the same values as the kernels this one replaces, so V and the scales are
// bit-identical to theirs (tests/test_hip_chain.py).  A segment computes the tile before and the tile after its own for
// their edge columns (two segments: 11 % more M reads).  Few threads, but each has 36 independent 16-byte loads in
// flight per step, and with one wave per SIMD (400-480 registers) the chip still has ~35 MB on the way.
// What bounds it (conv4 -> conv5, 256 lines): 0.162 ms; 0.10 with the stores ablated, 0.10 with the loads ablated, 0.06
// with both -- compute, load latency and store acknowledgement add up instead of overlapping.  The compiler waits for
// a step's loads with vmcnt(0): with loads AND stores pending it assumes they may complete out of order, so the wait
// also covers the 36 stores issued just before.  Hence plain stores, not the nontemporal ones of the separate input
// transform: the L2 acknowledges them sooner (0.156 -> 0.134 ms; conv3 -> conv4 0.123 -> 0.114).  Tried and dropped:
// loads as inline asm with hand-written vmcnt(36) waits (the register allocator moves the "loaded" values before
// they arrive: memory faults), the tile staged through LDS by direct-to-LDS loads with exact waits (correct, 0.152 ms).
// SRC 2 is conv5 -> pool3 -> conv6: the map is 4 rows high, the pool leaves 2, and the next layer is the 1-D F(4,2)
// of wino42_in_h3_kernel (tile tx - 1 = pooled columns 4 (tx - 1) .. 4 tx of both rows; scale = the exact maximum).
// ---------------------------------------------------------------------------
template <int SRC, int ACT>   // ACT 1: ReLU, 2: LeakyReLU(0.2) -- a compile-time switch: as a run-time one it was compiled to branches per element
__global__ __launch_bounds__(128) void wino4_out_in_h3_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                              const float* __restrict__ vs_in, float us,
                                                              const int32_t* __restrict__ wvalid, float* __restrict__ V,
                                                              float* __restrict__ vs_out, int W, int C,
                                                              long ntile_pad_in, long ntile_pad_out, int nseg) {
  __shared__ float wmax[2][2];
  CRNN_WHOLE_REGISTER_FILE();
  const int C4 = C >> 2, c4 = threadIdx.x, wave = threadIdx.x >> 6;
  const int tw = (W + 3) >> 2;                              // tiles per tile row of layer L
  // width of layer L + 1's input map and its tiles per line (SRC 2: conv6 has Wn - 1 output columns in tiles of 4)
  const int Wn = SRC ? W + 1 : W, twn = SRC == 2 ? (Wn - 1 + 3) >> 2 : (Wn + 3) >> 2;
  const int b = blockIdx.x / nseg, seg = blockIdx.x - b * nseg;
  const int seglen = (twn + nseg - 1) / nseg;
  const int tx0 = seg * seglen, tx1 = tx0 + seglen < twn ? tx0 + seglen : twn;
  if (tx0 >= tx1) return;                                   // (uniform over the workgroup)
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bs = *(const f32x4*)(bias + c4 * 4);
  const int wv = wvalid ? (wvalid[b] < W ? wvalid[b] : W) : W;
  // Buffer addressing (both tensors are below 2 GB: wino4_out_in_ok): one VGPR offset per tile, the component stride in
  // an SGPR offset -- with 64-bit pointers every one of the 72 accesses of a step cost a 64-bit multiply on the scalar
  // unit, which a wave that is alone on its SIMD pays in full
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Mo, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned cs_in = (unsigned)(ntile_pad_in * C * 4);                        // bytes per component
  const unsigned cs_out = (unsigned)(ntile_pad_out * C * 4) * (SRC == 2 ? 2u : 1u);   // (conv6's V: two rows of C per tile)
  // the 36 products of layer L's tile t (read once: nontemporal); all loads first: the memory parallelism has to come
  // from inside a thread
  auto load_tile = [&](long t, f32x4 (&mm)[6][6]) __attribute__((always_inline)) {
    const unsigned vo = (unsigned)((t * C + c4 * 4) * 4);
    unsigned so = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
      for (int j = 0; j < 6; j++, so += cs_in)
        mm[j][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, vo, so, 2));
  };
  // A^T m A of that tile (column index tx), scaled, bias, activation, columns from wv on zero
  auto out_tile = [&](const f32x4 (&mm)[6][6], long t, int tx, f32x4 (&o)[4][4]) __attribute__((always_inline)) {
    const float rs = vs_in[t] * us;
    f32x4 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      f32x4 c[4];
      wino4_at(mm[j], c);
#pragma unroll
      for (int k = 0; k < 4; k++) s[k][j] = c[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      wino4_at(s[k], o[k]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int xx = 4 * tx + j;
        f32x4 v = o[k][j] * rs + bs;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float e = v[c];
          if (ACT == 1) e = e > 0.f ? e : 0.f;
          else if (ACT == 2) e = e > 0.f ? e : 0.2f * e;
          if (xx >= wv) e = 0.f;
          v[c] = e;
        }
        o[k][j] = v;
      }
    }
  };
  f32x4 prev[4][4], pp3[4], o[4][4];   // tile tx - 1, column 3 of tile tx - 2, tile tx of the next layer's input map
  f32x4 carry[4];                      // SRC 1: conv column 4 tx - 1, maxima of the four row pairs
#pragma unroll
  for (int r = 0; r < 4; r++) {
    pp3[r] = z;
    carry[r] = z;
#pragma unroll
    for (int j = 0; j < 4; j++) prev[r][j] = z;
  }
  int par = 0;
  const int txs = tx0 > 0 ? tx0 - 1 : 0;
  const long trow = SRC == 1 ? (long)b * 2 * tw : (long)b * tw;   // layer L's first tile of this line (SRC 1: of its upper tile row)
  // The products of a step's tile are requested one step ahead: they are in flight under the input transform of the
  // step before (a wave has nobody to hide its load latency behind: one wave per SIMD).
  f32x4 mm[6][6];
  if (txs < tw) load_tile(trow + txs, mm);
#pragma unroll 1
  for (int tx = txs; tx <= tx1; tx++) {
    const bool more = tx + 1 <= tx1 && tx + 1 < tw;
    // ---- tile tx of the map (zeros past its last tile)
    if (SRC == 0) {
      if (tx < tw) out_tile(mm, trow + tx, tx, o);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) o[r][j] = z;
      }
      if (more) load_tile(trow + tx + 1, mm);
    } else {
      constexpr int PR = SRC == 1 ? 4 : 2;   // pooled rows
      f32x4 cm[PR][4];   // [pooled row][column of the tile]: maximum over the two conv rows of the pool window
      if (SRC == 2) {
        if (tx < tw) {
          f32x4 c[4][4];
          out_tile(mm, trow + tx, tx, c);
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              cm[0][j][e] = fmaxf(c[0][j][e], c[1][j][e]);
              cm[1][j][e] = fmaxf(c[2][j][e], c[3][j][e]);
            }
          if (more) load_tile(trow + tx + 1, mm);
        } else {
#pragma unroll
          for (int r = 0; r < PR; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) cm[r][j] = z;
        }
      } else if (tx < tw) {
        f32x4 m1[6][6], c[4][4];
        load_tile(trow + tw + tx, m1);     // the lower tile row: in flight under the upper tile's transform
        out_tile(mm, trow + tx, tx, c);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            cm[0][j][e] = fmaxf(c[0][j][e], c[1][j][e]);
            cm[1][j][e] = fmaxf(c[2][j][e], c[3][j][e]);
          }
        if (more) load_tile(trow + tx + 1, mm);
        out_tile(m1, trow + tw + tx, tx, c);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            cm[PR - 2][j][e] = fmaxf(c[0][j][e], c[1][j][e]);
            cm[PR - 1][j][e] = fmaxf(c[2][j][e], c[3][j][e]);
          }
      } else {
#pragma unroll
        for (int r = 0; r < PR; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) cm[r][j] = z;
      }
#pragma unroll
      for (int r = 0; r < PR; r++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int xx = 4 * tx + j;            // pooled column xx = max over conv columns xx - 1 and xx
          const f32x4 left = j == 0 ? carry[r] : cm[r][j - 1];
          f32x4 v;
          if (xx > W) v = z;                    // past the pooled map
          else if (xx == W) v = left;           // right edge: column W - 1 alone
          else if (xx == 0) v = cm[r][0];       // left edge: column 0 alone
          else {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(left[e], cm[r][j][e]);
          }
          o[r][j] = v;
        }
        carry[r] = cm[r][3];
      }
    }
    // ---- conv6's tile tx - 1: pooled columns prev[.][0..3] | o[.][0] of both rows (wino42_in_h3_kernel's arithmetic)
    if (SRC == 2 && tx > tx0) {
      f32x4 v[2][5];
      float mx = 0.f;
#pragma unroll
      for (int ky = 0; ky < 2; ky++) {
        const f32x4 d0 = prev[ky][0], d1 = prev[ky][1], d2 = prev[ky][2], d3 = prev[ky][3], d4 = o[ky][0];
        const f32x4 d31 = d3 - d1;
        v[ky][0] = 2.f * (d0 - d2) + d31;
        v[ky][1] = d3 - d2 - 2.f * d1;
        v[ky][2] = 2.f * d1 - 3.f * d2 + d3;
        v[ky][3] = d31;
        v[ky][4] = 2.f * (d1 - d3) - d2 + d4;
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
          for (int e = 0; e < 4; e++) mx = fmaxf(mx, __builtin_fabsf(v[ky][k][e]));
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      if ((threadIdx.x & 63) == 0) wmax[par][wave] = mx;
      __syncthreads();
      mx = fmaxf(mx, wmax[par][wave ^ 1]);
      par ^= 1;
      float sc, uns;
      split_scale_from_max(mx, sc, uns);
      const long tn = (long)b * twn + tx - 1;
      if (c4 == 0) vs_out[tn] = uns;
#pragma unroll
      for (int ky = 0; ky < 2; ky++) {
        const unsigned vo = (unsigned)(((tn * 2 + ky) * C + (c4 >> 1) * 8 + (c4 & 1) * 4) * 4);
        unsigned so = 0;
#pragma unroll
        for (int k = 0; k < 5; k++, so += cs_out) {
          uint2 hi, lo;
          split4(v[ky][k], sc, hi, lo);
          __builtin_amdgcn_raw_buffer_store_b128(split_pair_chunk(c4, hi, lo), v_rsrc, vo, so, 0);
        }
      }
    }
    // ---- the next layer's tile tx - 1: patch columns pp3 | prev[.][0..3] | o[.][0]
    if (SRC != 2 && tx > tx0) {
      f32x4 u[6][6];
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 6; j++) {
        f32x4 d[6], c[6];
        d[0] = z;
        d[5] = z;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          d[r + 1] = j == 0 ? pp3[r] : (j == 5 ? o[r][0] : prev[r][j == 0 || j == 5 ? 0 : j - 1]);
#pragma unroll
          for (int e = 0; e < 4; e++) mx = fmaxf(mx, __builtin_fabsf(d[r + 1][e]));
        }
        wino4_bt(d, c);
#pragma unroll
        for (int k = 0; k < 6; k++) u[k][j] = c[k];
      }
      for (int off = 1; off < C4 && off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      if (C4 > 64) {
        if ((threadIdx.x & 63) == 0) wmax[par][wave] = mx;
        __syncthreads();
        mx = fmaxf(mx, wmax[par][wave ^ 1]);
        par ^= 1;
      }
      float sc, uns;
      split_scale_from_max(49.f * mx, sc, uns);   // (the bound wino4_in_h3_kernel uses)
      const long tn = (long)b * twn + tx - 1;
      if (c4 == 0) vs_out[tn] = uns;
      // (store_split_pair's layout: the even lane of a pair stores the 16-byte hi chunk of 8 channels, the odd lane the lo chunk)
      const unsigned vo = (unsigned)((tn * C + (c4 >> 1) * 8 + (c4 & 1) * 4) * 4);
      unsigned so = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        f32x4 v[6];
        wino4_bt(u[k], v);
#pragma unroll
        for (int j = 0; j < 6; j++, so += cs_out) {
          uint2 hi, lo;
          split4(v[j], sc, hi, lo);
          __builtin_amdgcn_raw_buffer_store_b128(split_pair_chunk(c4, hi, lo), v_rsrc, vo, so, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < (SRC == 2 ? 2 : 4); r++) {
      pp3[r] = prev[r][3];
#pragma unroll
      for (int j = 0; j < 4; j++) prev[r][j] = o[r][j];
    }
  }
}

// src 0: layer L's map is 4 rows high and feeds layer L + 1 directly; 1: 8 rows high, MaxPool (2,2)/(2,1)/pad (0,1) between;
// 2: 4 rows high, that pool, then conv6's F(4,2) (C = 512 only, as wino42_in_h3_kernel)
bool wino4_out_in_ok(int B, int H, int W, int C, int src) {
  if (src == 2)
    return C == 512 && H == 4 && W >= 4 && B >= 1 && 36.0 * (double)wino4_pad(wino4_tiles(B, H, W)) * C * 4.0 < 2.0e9 &&
           5.0 * (double)wino4_pad((long)B * ((W + 3) / 4)) * 2 * C * 4.0 < 2.0e9;
  return (C == 256 || C == 512) && H == (src ? 8 : 4) && W >= 4 && B >= 1 &&
         36.0 * (double)wino4_pad(wino4_tiles(B, H, W)) * C * 4.0 < 2.0e9 &&
         36.0 * (double)wino4_pad(wino4_tiles(B, 4, src ? W + 1 : W)) * C * 4.0 < 2.0e9;   // (32-bit buffer offsets)
}
void launch_wino4_out_in(const float* M, const float* bias, const float* vs_in, float us, const int32_t* wvalid, float* V,
                         float* vs_out, int B, int H, int W, int C, int act, int src, hipStream_t s) {
  const int Wn = src ? W + 1 : W, twn = src == 2 ? (Wn - 1 + 3) / 4 : (Wn + 3) / 4;
  static const int nseg_env = getenv("CRNN_WINO4_CHAIN_NSEG") ? atoi(getenv("CRNN_WINO4_CHAIN_NSEG")) : 0;
  // 256 lines: two segments of two-wave workgroups (C = 512) are one wave per SIMD over the chip; for the one-wave
  // workgroups of C = 256 three segments measured best (conv3 -> conv4: 0.123 ms against 0.133 with four, 0.139 with two)
  int nseg = nseg_env > 0 ? nseg_env : (C == 512 ? 2 : 3);
  if (nseg > twn) nseg = twn;
  const long tp_in = wino4_pad(wino4_tiles(B, H, W));
  const long tp_out = src == 2 ? wino4_pad((long)B * twn) : wino4_pad(wino4_tiles(B, 4, Wn));
  const dim3 g(B * nseg), bl(C / 4);
  if (src == 2 && act == 1) hipLaunchKernelGGL((wino4_out_in_h3_kernel<2, 1>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
  else if (src == 2) hipLaunchKernelGGL((wino4_out_in_h3_kernel<2, 2>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
  else if (src && act == 1) hipLaunchKernelGGL((wino4_out_in_h3_kernel<1, 1>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
  else if (src) hipLaunchKernelGGL((wino4_out_in_h3_kernel<1, 2>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
  else if (act == 1) hipLaunchKernelGGL((wino4_out_in_h3_kernel<0, 1>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
  else hipLaunchKernelGGL((wino4_out_in_h3_kernel<0, 2>), g, bl, 0, s, M, bias, vs_in, us, wvalid, V, vs_out, W, C, tp_in, tp_out, nseg);
}

// The 36 GEMMs M[c] = V[c] U[c]^T as ONE persistent launch.  Work unit = (component, 128 tiles, TBN channels),
// K = Cin; unit u = (component * ntm + m-tile) * ntn + n-tile.  Block L takes units L, L + G, L + 2G, ...:
// the launch is G = one block per CU slot, every block gets the same number of units (+-1) -- no partly filled
// last round -- and at any moment the blocks work on G consecutive units, i.e. on one or two components, so
// the U slice and the V tiles that several units share are L2 hits (the XCD-aware block order keeps the units
// that share a V tile on one XCD).  Inside a block the K steps of all its units form one flat loop over the
// same three-stage direct-to-LDS pipeline: the loader walks into the next unit while
// the MFMAs finish the current one, whose product is stored straight from the accumulators.
// The kernel (f32 MFMA form; the version in which every wave issued its share of the loads is gone) has a dedicated
// LOADER WAVE: a ninth wave issues all direct-to-LDS operand loads of the
// block (32 per K step) and is the only one that waits on vmcnt; the eight MFMA waves only read fragments,
// issue MFMAs and store finished products -- their stores no longer sit in the counter the operand pipeline
// waits on, and they issue no load instructions at all.  One barrier per K step, executed by all nine waves:
// before it the loader has seen the next step's operands land and every MFMA wave has read the current
// stage into registers; after it the loader refills that stage for step + 3.
__global__ __launch_bounds__(576, 1) void wino4_gemm_lw_kernel(const Wino4Gemm p) {
  constexpr int NWN = 4, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];   // [B0 B1 B2 | A0 A1 A2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.Cin / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.Cin * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.Cin * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };

  if (wave == 8) {
    // ---------------- loader ----------------
    // chunk position P = q * 64 + lane (q = 0..15) of a stage half holds (row, k-chunk) given by the swizzle
    unsigned off[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int P = q * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      off[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.Cin + (c16 & 7) * 4) * 4);
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
      unsigned m0_saved;   // M0 is restored: the compiler may keep a value of its own there (clobbering it is not allowed)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    };
    int l_u = L, l_k0 = 0;
    unsigned long long pa, pb, pm;
    bases(l_u, pa, pb, pm);
    auto gload = [&](int stage) __attribute__((always_inline)) {
      const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const unsigned sa = lds0 + (3 * SB + stage * SA) * 4, sb = lds0 + stage * SB * 4;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        dma(a_rsrc, sa + q * 1024, off[q], l_k0 * 4);
        dma(b_rsrc, sb + q * 1024, off[q], l_k0 * 4);
      }
      l_k0 += BK;
      if (l_k0 == p.Cin) {
        l_k0 = 0;
        if (l_u + G < NU) l_u += G;
        bases(l_u, pa, pb, pm);
      }
    };
    gload(0);
    gload(1);
    gload(2);
    asm volatile("s_waitcnt vmcnt(63)" ::: "memory");   // 96 issued: at most 63 left => step 0 (the first 32) landed
    __builtin_amdgcn_s_barrier();
    int st = 0;
#pragma unroll 1
    for (int it = 0; it < nit; it++) {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // step it + 1 landed (step it + 2 may be in flight)
      __builtin_amdgcn_s_barrier();
      if (st == 0) gload(0);
      else if (st == 1) gload(1);
      else gload(2);
      st = st == 2 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS write may be in flight when the block ends
    return;
  }

  // ---------------- MFMA waves ----------------
  const int wm = wave / NWN, wn = wave % NWN;
  int aaddr[4], baddr[4];
  {
    const int Ra = wm * 64 + (lane & 31), Rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int kg = 0; kg < 4; kg++) {
      aaddr[kg] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (2 * kg + h)) ^ ((Ra >> 1) & 15)) << 2);
      baddr[kg] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (2 * kg + h)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
  int kc = 0, c_u = L;
  const unsigned moff = (unsigned)((wm * 64 + (lane & 31)) * p.N + wn * 32 + 4 * (lane >> 5));
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
        *(gf32x4*)(dst + mt * 32 * p.N + 8 * g) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) acc[mt][4 * g + e] = 0.f;
      }
    kc = 0;
    c_u += G;
  };
  f32x4 a0[2], a1[2], b0[2];
  auto frag = [&](int stage, int kg, int set) __attribute__((always_inline)) {
    const float* sa = smem + 3 * SB + stage * SA;
    const float* sb = smem + stage * SB;
    a0[set] = *(const f32x4*)(sa + aaddr[kg]);
    a1[set] = *(const f32x4*)(sa + aaddr[kg] + 1024);
    b0[set] = *(const f32x4*)(sb + baddr[kg]);
  };
  auto mfma8 = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[set][j], a0[set][j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[set][j], a1[set][j], acc[1], 0, 0, 0);
    }
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, nx = (st + 1) % 3;
    frag(st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(0);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: stage st is in registers (vmcnt / expcnt not waited)
    __builtin_amdgcn_s_barrier();
    frag(nx, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
  };
  __builtin_amdgcn_s_barrier();
  frag(0, 0, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// ---------------------------------------------------------------------------
// The Winograd component GEMMs on operands in the f16x3 format (see wino4_in_h3_kernel): same persistent balanced-unit
// structure, unit order, three-stage direct-to-LDS pipeline, LDS layout and swizzle as wino4_gemm_lw_kernel -- a row's
// K = 32 step is still 128 bytes = 8 chunks of 16 bytes, now hi(k0-7) lo(k0-7) hi(k8-15) lo(k8-15) ... -- but a K step
// is two k16 MFMA steps of three products each (lo hi, hi lo, hi hi into one accumulator): 12 MFMAs of 32 cycles per wave
// and K step instead of 32 of 64.  The products leave scaled by s_tile s_weights; the output transform undoes that.
// 128 x 128 units, 8 waves as 2 (M) x 4 (N), wave tile 64 x 32.
// ---------------------------------------------------------------------------
// NLW dedicated loader waves (waves 8 .. 8 + NLW - 1, as in wino4_gemm_lw_kernel): they issue all
// direct-to-LDS loads of the block (32 per K step) and are the only waves that wait on vmcnt; the eight MFMA waves read
// fragments, issue MFMAs and store finished products.  One barrier per K step, executed by all waves.  With the MFMA
// phase of a step only 384 cycles per wave, one loader wave cannot issue 32 loads in time: four do (8 each).  Measured
// against the form in which every wave loads its share (tools/h3_gemm_bench.hip): -9 % at K = 512, -4..6 % at K = 256 and
// K = 1024, +5 % at K = 64 (conv1, which is HBM-bound on its product tensor and runs on the fused kernel below anyway).
template <int NLW>
__global__ __launch_bounds__(512 + 64 * NLW, 1) void wino4_gemm_h3_lw_kernel(const Wino4Gemm p) {
  constexpr int NWN = 4, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32, QW = 16 / NLW;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];
  __shared__ __attribute__((aligned(16))) float wtile[8 * 32 * 36];   // 36 KB: the product-store transpose tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.Cin / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.Cin * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.Cin * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  if (wave >= 8) {
    const int lw = wave - 8;
    unsigned off[QW];
#pragma unroll
    for (int q = 0; q < QW; q++) {
      const int P = (lw * QW + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      off[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.Cin + (c16 & 7) * 4) * 4);
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)(lw * QW) * 1024u;
    auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
      unsigned m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    };
    int l_u = L, l_k0 = 0;
    unsigned long long pa, pb, pm;
    bases(l_u, pa, pb, pm);
    auto gload = [&](int stage) __attribute__((always_inline)) {
      const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const unsigned sa = lds0 + (3 * SB + stage * SA) * 4, sb = lds0 + stage * SB * 4;
#pragma unroll
      for (int q = 0; q < QW; q++) {
        dma(a_rsrc, sa + q * 1024, off[q], l_k0 * 4);
        dma(b_rsrc, sb + q * 1024, off[q], l_k0 * 4);
      }
      l_k0 += BK;
      if (l_k0 == p.Cin) {
        l_k0 = 0;
        if (l_u + G < NU) l_u += G;
        bases(l_u, pa, pb, pm);
      }
    };
    gload(0);
    gload(1);
    gload(2);
    if (QW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (QW == 8) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
#pragma unroll 1
    for (int it = 0; it < nit; it++) {
      if (QW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // step it + 1 landed (step it + 2 may be in flight)
      else if (QW == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (st == 0) gload(0);
      else if (st == 1) gload(1);
      else gload(2);
      st = st == 2 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int wm = wave / NWN, wn = wave % NWN;
  int ahi[2], alo[2], bhi[2], blo[2];
  {
    const int Ra = wm * 64 + (lane & 31), Rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = 4 * j + 2 * h;
      ahi[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | c) ^ ((Ra >> 1) & 15)) << 2);
      alo[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (c + 1)) ^ ((Ra >> 1) & 15)) << 2);
      bhi[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | c) ^ ((Rb >> 1) & 15)) << 2);
      blo[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (c + 1)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
  int kc = 0, c_u = L;
  // Product stores through LDS: from the accumulators a lane would write 16 bytes of 32 different rows (32-byte pieces of
  // M's 128-byte lines); transposed through a per-wave LDS tile (32 rows x 144 bytes) eight lanes write one full line.
  // Measured with the stores ablated they cost 0.07 ms of a layer's 0.19-0.30 ms (tools/fused_bench.hip, H3_NOSTORE).
  float* wst = wtile + wave * (32 * 36);
  const int wr_off = (lane & 31) * 36 + 4 * (lane >> 5), rd_off = (lane >> 3) * 36 + (lane & 7) * 4;
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    __attribute__((address_space(1))) float* dst =
        (__attribute__((address_space(1))) float*)pm + (unsigned)((wm * 64 + (lane >> 3)) * p.N + wn * 32 + (lane & 7) * 4);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v = {acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]};
        *(f32x4*)(wst + wr_off + 8 * g) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) acc[mt][4 * g + e] = 0.f;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave's own tile is in LDS
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const f32x4 v = *(const f32x4*)(wst + rd_off + i * 8 * 36);
#ifdef H3_NOSTORE
        if (v[0] == 12345.f)
#endif
        *(gf32x4*)(dst + (mt * 32 + i * 8) * p.N) = v;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // the reads are done before the next pass overwrites the tile
      __builtin_amdgcn_wave_barrier();
    }
    kc = 0;
    c_u += G;
  };
  f16x8 a0h[2], a0l[2], a1h[2], a1l[2], bh[2], bl[2];
  auto frag = [&](int stage, int j, int set) __attribute__((always_inline)) {
    const float* sa = smem + 3 * SB + stage * SA;
    const float* sb = smem + stage * SB;
    a0h[set] = *(const f16x8*)(sa + ahi[j]);
    a0l[set] = *(const f16x8*)(sa + alo[j]);
    a1h[set] = *(const f16x8*)(sa + ahi[j] + 1024);
    a1l[set] = *(const f16x8*)(sa + alo[j] + 1024);
    bh[set] = *(const f16x8*)(sb + bhi[j]);
    bl[set] = *(const f16x8*)(sb + blo[j]);
  };
  auto mfma6 = [&](int set) __attribute__((always_inline)) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[set], a0h[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[set], a1h[set], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a0l[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a1l[set], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a0h[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a1h[set], acc[1], 0, 0, 0);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, nx = (st + 1) % 3;
    frag(st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
    frag(nx, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
  };
  __builtin_amdgcn_s_barrier();
  frag(0, 0, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
}



// ---------------------------------------------------------------------------------------------------------------------
// The f16x3 component GEMMs, ping-pong form (round 3; CRNN_GEMM_PP=0 keeps the form above).  Units of 256 tiles x 128
// channels; eight MFMA waves as 4 (tiles) x 2 (channels), wave tile 64 x 64 = four accumulators (8 fragment reads per 12
// MFMAs instead of 6 per 6), in TWO GROUPS of four -- one wave of each group on every SIMD -- that run one slot apart: while
// group 0 issues the 12 MFMAs of a k16 half-step, group 1 reads the fragments of its next one, and the other way round, a
// barrier between slots.  A SIMD's matrix pipe always has one wave's MFMAs, the other wave's fragment-read latency is under
// them, and one fragment register set is enough (134 registers).  Four loader waves issue the direct-to-LDS loads: 48 pieces
// of 1 KiB per K step into three stages of 48 KB (A rows first, then B rows; the row-pair / XOR-swizzled layout of the
// kernels above).  Barrier B_t opens slot t:
//   group 0   reads (K step s, half j) in slot 4 s + 2 j, waits for them and issues the MFMAs in the slot after;
//   group 1   the same one slot later (it passes one extra barrier at the start, group 0 one at the end);
//   loaders   the batch of K step s + 2 -- stage (s + 2) % 3, last read in K step s - 1, and those reads were waited for by
//             slot 4 s at the latest -- goes out in slots 4 s + 1 .. 4 s + 3; then vmcnt(12): the batch of K step s + 1 has
//             landed before B_{4 s + 4}, and group 0 first reads it in slot 4 s + 4 (a read comes one slot after the wait
//             that retires its stage, never in the same slot).
// The MFMAs take the V fragment first, so a lane ends with one channel of rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a
// 32 x 32 product: a dword store instruction writes two full 128-byte lines of M, no transpose through LDS.
// Measured in tools/h3_gemm_bench.hip against the loader-wave form in the same process (conv5 / conv3 / conv4 / conv6
// shapes at 256 lines): 0.305 / 0.181 / 0.182 / 0.306 ms against 0.348 / 0.221 / 0.219 / 0.338.  In core cycles the kernel
// is within 30 % of the MFMA issue time; what it lacks in wall time is clock: s_memtime against s_memrealtime inside the
// kernel gives 1.0-1.3 GHz ("the power wall").
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768, 1) void wino4_gemm_h3_pp_kernel(const Wino4Gemm p) {
  constexpr int TBM = 256, TBN = 128, SA = TBM * 32, SB = TBN * 32, STG = SA + SB;
  __shared__ __attribute__((aligned(1024))) float smem[3 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.Cin / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.Cin * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.Cin * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  if (wave >= 8) {
    const int lw = wave - 8;
    unsigned offa[8], offb[4];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int P = (lw * 8 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offa[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.Cin + (c16 & 7) * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int P = (lw * 4 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offb[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.Cin + (c16 & 7) * 4) * 4);
    }
    const unsigned ldsA = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)(lw * 8) * 1024u;
    const unsigned ldsB = (unsigned)(size_t)(lds_ptr_t)smem + SA * 4 + (unsigned)(lw * 4) * 1024u;
    auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
      unsigned m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    };
    int l_u = L, l_k0 = 0, lst = 0;
    unsigned long long pa, pb, pm;
    bases(l_u, pa, pb, pm);
    auto part = [&](int which) __attribute__((always_inline)) {
      const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
      const unsigned so = (unsigned)lst * (STG * 4);
      if (which == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
      } else if (which == 1) {
#pragma unroll
        for (int q = 4; q < 8; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) dma(b_rsrc, ldsB + so + q * 1024, offb[q], l_k0 * 4);
        l_k0 += BK;
        lst = lst == 2 ? 0 : lst + 1;
        if (l_k0 == p.Cin) {
          l_k0 = 0;
          if (l_u + G < NU) l_u += G;
          bases(l_u, pa, pb, pm);
        }
      }
    };
    part(0); part(1); part(2);
    if (nit > 1) { part(0); part(1); part(2); }
    if (nit > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // B_0
#pragma unroll 1
    for (int s = 0; s < nit; s++) {
      const bool more = s + 2 < nit;
      __builtin_amdgcn_s_barrier();
      if (more) part(0);
      __builtin_amdgcn_s_barrier();
      if (more) part(1);
      __builtin_amdgcn_s_barrier();
      if (more) {
        part(2);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
    return;
  }
  const int grp = wave >> 2, wm = wave & 3, wn = grp;
  int ahi[2], alo[2], bhi[2], blo[2];
  {
    const int Ra = wm * 64 + (lane & 31), Rb = wn * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = 4 * j + 2 * h;
      ahi[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | c) ^ ((Ra >> 1) & 15)) << 2);
      alo[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (c + 1)) ^ ((Ra >> 1) & 15)) << 2);
      bhi[j] = SA + (Rb >> 1) * 64 + (((((Rb & 1) << 3) | c) ^ ((Rb >> 1) & 15)) << 2);
      blo[j] = SA + (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (c + 1)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
  int kc = 0, c_u = L;
  const unsigned moff = (unsigned)((wm * 64 + 4 * (lane >> 5)) * p.N + wn * 64 + (lane & 31));
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
    // plain-GEMM form (lstm0_ih, emb0, lstm1_ih): unscale per 32 rows, bias per column (the lane's two columns)
    float rsc[2] = {1.f, 1.f}, bsv[2] = {0.f, 0.f};
    if (p.lin_bias) {
      const int ntu = c_u % ntn, mtu = (c_u / ntn) % ntm;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        rsc[i] = p.lin_rs[mtu * (TBM / 32) + wm * 2 + i] * p.lin_us;
        bsv[i] = p.lin_bias[ntu * TBN + wn * 64 + i * 32 + (lane & 31)];
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
#ifdef H3_NOSTORE
          if (acc[mt][nt][r] == 12345.f)
#endif
          dst[(mt * 32 + (r & 3) + 8 * (r >> 2)) * p.N + nt * 32] = p.lin_bias ? acc[mt][nt][r] * rsc[mt] + bsv[nt] : acc[mt][nt][r];
          acc[mt][nt][r] = 0.f;
        }
    kc = 0;
    c_u += G;
  };
  f16x8 ah[2], al[2], bh[2], bl[2];
  auto half = [&](int stage, int j) __attribute__((always_inline)) {
    const float* sb = smem + stage * STG;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      ah[t] = *(const f16x8*)(sb + ahi[j] + 1024 * t);
      al[t] = *(const f16x8*)(sb + alo[j] + 1024 * t);
      bh[t] = *(const f16x8*)(sb + bhi[j] + 1024 * t);
      bl[t] = *(const f16x8*)(sb + blo[j] + 1024 * t);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    // small terms first (lo hi, hi lo, then hi hi); four accumulators, so equal accumulators are four MFMAs apart
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    half(st, 0);
    __builtin_amdgcn_s_barrier();
    half(st, 1);
    if (++kc == nk) flush();
    __builtin_amdgcn_s_barrier();
  };
  __builtin_amdgcn_s_barrier();                            // B_0
  if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one slot behind
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------
// F(4x4,3x3) component GEMMs AND output transform in one kernel (f16x3 operands): the product tensor M (2.25x the layer's
// output, written and read back by the three-kernel form) never exists.  Keeping all 36 products of an output block in
// registers caps the wave tile at 32 x 16 with nothing left (wino4_c64_fused_kernel); instead the output transform is
// ACCUMULATED while the components go by: with the components taken column by column (j outer, i inner),
//   T[a]    += A^T[a][i] m_ij        after the K loop of component (i, j)    (4 planes, compile-time coefficients)
//   Y[a][b] += A^T[b][j] T[a]        after the sixth component of column j   (16 planes)
// so a wave holds 16 + 4 + 1 planes of its tile: 32 tiles x 16 channels = 168 registers, two waves per SIMD, and the
// VALU work of one wave (6 operations per product element) runs under the MFMAs of the other.
// Block = 8 waves = 64 tiles x 64 channels, as 2 (tiles) x 4 (channels); v_mfma_f32_16x16x32_f16 with the U fragment
// first, so a lane ends with 4 consecutive channels of one tile.  Unit = (64 tiles, 64 channels), all 36 components,
// K = Cin in steps of 64: 36 Cin / 64 steps through a four-stage direct-to-LDS pipeline (stage = 64 + 64 rows of 256
// bytes, chunk c of row R at position c ^ (R & 15): conflict-free 16-byte fragment reads), every wave issuing four of
// the 32 loads of a step.  Persistent: block L takes units L, L + G, ... (XCD-aware order: the channel blocks of one
// tile block run on one XCD), the loads walk into the next unit while the epilogue of the last one runs.
// Epilogue: unscale (vs[tile] us), bias, activation, width mask; POOL 0: 16 stores of 16 bytes per tile, 1: MaxPool
// 2x2/2 inside the tile (pool1).
// Measured (256 lines; tools/fused_bench.hip): conv1 0.20-0.23 ms against 0.32 for GEMM + output transform, conv2 0.22
// against 0.22, conv3 / conv4 0.40 against 0.27-0.28, conv5 0.71 against 0.37.  The 64 x 64 unit (all the registers
// allow) draws twice the operand bytes per FLOP through L1 and LDS of the 128 x 128 GEMM: with the loads ablated conv5
// takes 0.36 ms, with MFMAs and fragment reads ablated 0.48 (5.4 GB from L2 at 11 TB/s, the rate the 128 x 128 GEMM also
// tops out at), so it pays only where the product tensor dominates (K <= 128).
// ---------------------------------------------------------------------------
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b);
#ifndef FUSED_ABL
#define FUSED_ABL 0   // tools/fused_bench.hip: 1 no MFMAs, 2 fragments read once, 4 no loads, 8 no barrier, 16 no transform
#endif
template <int POOL>
__global__ __launch_bounds__(512, 1) void wino4_fused_h3_kernel(const Wino4Fused p) {
  constexpr int NST = 4, STG = 8192, SVF = 4096;   // floats per stage; the U rows start SVF floats into a stage
  __shared__ __attribute__((aligned(1024))) float smem[NST * STG];   // 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, r16 = lane & 15, g = lane >> 4;
  const int ntn = p.N >> 6, ntm = p.ntile_pad >> 6;
  const int NU = ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.Cin >> 6;
  // ---- loads: instruction q of this wave brings rows 4 I .. 4 I + 3 (I = 2 wave + q) of the V block and of the U block
  unsigned voff[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int row = 4 * (2 * wave + q) + g, c = r16 ^ (row & 15);
    voff[q] = (unsigned)(row * p.Cin * 4 + c * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 2048u;
  auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned vo, int soff) __attribute__((always_inline)) {
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(vo), "s"(rsrc), "s"(soff) : "memory");
  };
  int l_u = L, l_c = 0, l_ks = 0, l_st = 0;   // next (unit, component in issue order, K step) to load and its stage
  auto gload = [&]() __attribute__((always_inline)) {
    const int comp = (l_c % 6) * 6 + l_c / 6;   // column by column: i = l_c % 6, j = l_c / 6
    const int tb = l_u / ntn, cb = l_u - tb * ntn;
    const unsigned long long pa = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)tb * 64) * p.Cin * 4ull;
    const unsigned long long pb = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)cb * 64) * p.Cin * 4ull;
    const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
    const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
    const unsigned sv = lds0 + (unsigned)l_st * (STG * 4), su = sv + SVF * 4;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      dma(a_rsrc, sv + q * 1024, voff[q], l_ks * 256);
      dma(b_rsrc, su + q * 1024, voff[q], l_ks * 256);
    }
    l_st = (l_st + 1) & (NST - 1);
    if (++l_ks == nk) {
      l_ks = 0;
      if (++l_c == 36) {
        l_c = 0;
        if (l_u + G < NU) l_u += G;   // past the last unit: its loads are repeated (valid memory, never read)
      }
    }
  };
  // ---- fragments: k32 sub-step s of a stage, k-group g: hi chunk 8 s + 2 g, lo chunk the next one
  int ov[2][2], ou[2][2];   // [sub-step][hi / lo], float offsets inside a stage
  {
    const int Rv = wm * 32 + r16, Ru = wn * 16 + r16;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
      for (int hl = 0; hl < 2; hl++) {
        const int c = 8 * s2 + 2 * g + hl;
        ov[s2][hl] = Rv * 64 + ((c ^ (Rv & 15)) << 2);
        ou[s2][hl] = SVF + Ru * 64 + ((c ^ (Ru & 15)) << 2);
      }
  }
  f32x4 Y[4][4][2], T[4][2], acc[2];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int th = 0; th < 2; th++) {
      T[a][th] = z4;
#pragma unroll
      for (int b = 0; b < 4; b++) Y[a][b][th] = z4;
    }
  acc[0] = z4;
  acc[1] = z4;
  f16x8 uh[2], ul[2], vh[2][2], vl[2][2];   // [set]([tile half])
  int c_st = 0;
  auto frag = [&](int stage, int s2, int set) __attribute__((always_inline)) {
    const float* sb = smem + stage * STG;
    uh[set] = *(const f16x8*)(sb + ou[s2][0]);
    ul[set] = *(const f16x8*)(sb + ou[s2][1]);
    vh[set][0] = *(const f16x8*)(sb + ov[s2][0]);
    vl[set][0] = *(const f16x8*)(sb + ov[s2][1]);
    vh[set][1] = *(const f16x8*)(sb + ov[s2][0] + 1024);   // tile + 16: same swizzle
    vl[set][1] = *(const f16x8*)(sb + ov[s2][1] + 1024);
  };
  auto mfma6 = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int th = 0; th < 2; th++) acc[th] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul[set], vh[set][th], acc[th], 0, 0, 0);
#pragma unroll
    for (int th = 0; th < 2; th++) acc[th] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[set], vl[set][th], acc[th], 0, 0, 0);
#pragma unroll
    for (int th = 0; th < 2; th++) acc[th] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[set], vh[set][th], acc[th], 0, 0, 0);
  };
  auto step = [&]() __attribute__((always_inline)) {
    if (!(FUSED_ABL & 2)) frag(c_st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(FUSED_ABL & 1)) mfma6(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // the next step's loads have landed ((NST - 2) LPS may fly)
    if (!(FUSED_ABL & 8)) __syncthreads();
    if (!(FUSED_ABL & 4)) gload();                                 // refills the stage just read
    c_st = (c_st + 1) & (NST - 1);
    if (!(FUSED_ABL & 2)) frag(c_st, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(FUSED_ABL & 1)) mfma6(1);
    else if (!(FUSED_ABL & 2)) asm volatile("" ::"v"(uh[0]), "v"(ul[0]), "v"(vh[0][0]), "v"(vl[0][0]), "v"(vh[0][1]), "v"(vl[0][1]), "v"(uh[1]), "v"(ul[1]), "v"(vh[1][0]), "v"(vl[1][0]), "v"(vh[1][1]), "v"(vl[1][1]));
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f},
                              {0.f, 1.f, 1.f, 4.f, 0.25f, 0.f}, {0.f, 1.f, -1.f, 8.f, -0.125f, 1.f}};
  auto comp = [&](auto IC) __attribute__((always_inline)) {
    constexpr int I = decltype(IC)::value;
#pragma unroll 1
    for (int ks = 0; ks < nk; ks++) step();
#pragma unroll
    for (int th = 0; th < 2; th++) {
#pragma unroll
      for (int a = 0; a < 4; a++)
        if (AT[a][I] != 0.f && !(FUSED_ABL & 16)) T[a][th] += AT[a][I] * acc[th];
      if (!(FUSED_ABL & 16)) acc[th] = z4;
    }
  };
  const int nty = p.H >> 2, ntx = (p.W + 3) >> 2;
  auto unit_out = [&](int u) __attribute__((always_inline)) {
    const int tb = u / ntn, cb = u - tb * ntn;
    const int c = cb * 64 + wn * 16 + 4 * g;
    const f32x4 bs = *(const f32x4*)(p.bias + c);
#pragma unroll
    for (int th = 0; th < 2; th++) {
      const int t = tb * 64 + wm * 32 + th * 16 + r16;
      if (t < p.ntile) {
        const int tx = t % ntx, r = t / ntx, ty = r % nty, b = r / nty;
        const float rs = p.vs[t] * p.us;
        const int wv = p.wvalid ? p.wvalid[b] : p.W;
        auto val = [&](int a, int bb) __attribute__((always_inline)) {
          f32x4 v = Y[a][bb][th] * rs + bs;
          const int xx = 4 * tx + bb;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float x = v[e];
            if (p.act == 1) x = x > 0.f ? x : 0.f;
            else if (p.act == 2) x = x > 0.f ? x : 0.2f * x;
            if (xx >= wv) x = 0.f;
            v[e] = x;
          }
          return v;
        };
        if (POOL == 0) {
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int bb = 0; bb < 4; bb++)
              if (4 * tx + bb < p.W) *(f32x4*)(p.y + (((long)b * p.H + 4 * ty + a) * p.W + 4 * tx + bb) * p.N + c) = val(a, bb);
        } else if (POOL == 1) {
          const int Wp = p.W >> 1, Hp = p.H >> 1;
#pragma unroll
          for (int pi = 0; pi < 2; pi++)
#pragma unroll
            for (int pj = 0; pj < 2; pj++) {
              const int xp = 2 * tx + pj;
              if (xp >= Wp) continue;
              const f32x4 v = max4(max4(val(2 * pi, 2 * pj), val(2 * pi, 2 * pj + 1)), max4(val(2 * pi + 1, 2 * pj), val(2 * pi + 1, 2 * pj + 1)));
              *(f32x4*)(p.y + (((long)b * Hp + 2 * ty + pi) * Wp + xp) * p.N + c) = v;
            }
        }
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int bb = 0; bb < 4; bb++) Y[a][bb][th] = z4;
    }
  };
  // ---- pipeline start: NST steps in flight, the first one landed
#pragma unroll
  for (int i = 0; i < NST; i++) gload();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // (NST - 1) LPS
  __syncthreads();
  frag(0, 0, 0);
#pragma unroll 1
  for (int un = 0; un < my_units; un++) {
#pragma unroll 1
    for (int j = 0; j < 6; j++) {
      comp(std::integral_constant<int, 0>{});
      comp(std::integral_constant<int, 1>{});
      comp(std::integral_constant<int, 2>{});
      comp(std::integral_constant<int, 3>{});
      comp(std::integral_constant<int, 4>{});
      comp(std::integral_constant<int, 5>{});
      const float c0 = j < 5 ? 1.f : 0.f;
      const float c1 = j == 0 || j == 5 ? 0.f : (j == 1 ? 1.f : (j == 2 ? -1.f : (j == 3 ? 2.f : -0.5f)));
      const float c2 = j == 0 || j == 5 ? 0.f : c1 * c1;
      const float c3 = j == 0 ? 0.f : (j == 5 ? 1.f : c2 * c1);
#pragma unroll
      for (int th = 0; th < 2; th++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          Y[a][0][th] += c0 * T[a][th];
          Y[a][1][th] += c1 * T[a][th];
          Y[a][2][th] += c2 * T[a][th];
          Y[a][3][th] += c3 * T[a][th];
          T[a][th] = z4;
        }
    }
    unit_out(L + un * G);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void launch_wino4_fused(const Wino4Fused& p, hipStream_t s) {
  const int nu = (p.ntile_pad / 64) * (p.N / 64);
  const int g = nu < persistent_blocks() ? 8 * ((nu + 7) / 8) : persistent_blocks();
  if (p.pool == 1) hipLaunchKernelGGL((wino4_fused_h3_kernel<1>), dim3(g), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((wino4_fused_h3_kernel<0>), dim3(g), dim3(512), 0, s, p);
}

// ---------------------------------------------------------------------------
// Plain GEMM C = act(A W^T + bias) on the same persistent direct-to-LDS pipeline (LSTM input projections and
// the classifier: M = lines x T rows, K = 256 / 512).  Units of 128 x 128, unit u = m-tile * ntn + n-tile,
// block L takes units L, L + G, ...; rows past M / N are clamped in the loader and dropped in the epilogue.
// The bias sits in LDS (a global load in the epilogue would make the compiler wait for vmcnt(0), i.e. for
// every operand load in flight).  ARGMAX: instead of (or besides) C, every wave reduces its 64 x 32 piece to
// (max, lowest column) per row and writes one partial per row and 32-column group,
// pmax / pidx[row][n-tile * 4 + wave column]; launch_argmax_partials finishes the reduction.
// ---------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte store at a 4-byte aligned address
constexpr int LIN_MAXN = 8192;   // bias columns held in LDS (32 KB beside the 96 KB of stages)
template <bool ARGMAX>
__global__ __launch_bounds__(512, 1) void lin_gemm_kernel(const ConvGemm p) {
  constexpr int NWN = 4, NT = 512, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32, NLA = 2;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];   // [B0 B1 B2 | A0 A1 A2]
  __shared__ float sbias[LIN_MAXN];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int ntn = (p.N + TBN - 1) / TBN, ntm = (p.M + TBM - 1) / TBM;
  const int NU = ntm * ntn, G = gridDim.x;
  // Which units this block takes.  Small launches (G < 256): unit L + k G, L the XCD-aware block order.
  // Full launches: the 8 XCDs split the unit grid into gn column groups x 8/gn row groups, and the G/8 blocks
  // of an XCD stride through their rectangle row-major -- the rectangle's slice of W stays in that XCD's L2
  // and an A tile is fetched once per column group instead of once per XCD and n-tile.
  int nb, ne, gw, mb, me, first, stride, my_units;
  if (p.gn > 0) {   // (set by the launcher only for a full launch: G a multiple of 8, every XCD's rectangle >= G / 8 units)
    const int x = blockIdx.x & 7, gm = 8 / p.gn, cg = x % p.gn, rg = x / p.gn;
    nb = cg * ntn / p.gn; ne = (cg + 1) * ntn / p.gn;
    mb = rg * ntm / gm; me = (rg + 1) * ntm / gm;
    first = blockIdx.x >> 3; stride = G >> 3;
  } else {
    const int chunk = G >> 3;
    nb = 0; ne = ntn; mb = 0; me = ntm;
    first = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3); stride = G;
  }
  gw = ne - nb;
  const int NUx = gw * (me - mb);
  if (first >= NUx) return;
  my_units = (NUx - first + stride - 1) / stride;
  (void)NU;
  for (int i = tid; i < ntn * TBN; i += NT) sbias[i] = i < p.N ? p.bias[i] : 0.f;
  __syncthreads();

  // loader: local row and k-chunk byte offset of this lane's LDS chunk positions
  int lrow[2];
  unsigned lkb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int P = i * NT + tid, S = P >> 4, c16 = (P & 15) ^ (S & 15);
    lrow[i] = 2 * S + (c16 >> 3);
    lkb[i] = (unsigned)((c16 & 7) * 16);
  }
  unsigned aoff[2], boff[2];
  auto offsets = [&](int q) __attribute__((always_inline)) {
    const int qm = q / gw, mt = mb + qm, nt = nb + q - qm * gw;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int m = mt * TBM + lrow[i], n = nt * TBN + lrow[i];
      m = m < p.M ? m : p.M - 1;
      n = n < p.N ? n : p.N - 1;
      aoff[i] = (unsigned)m * (unsigned)(p.K * 4) + lkb[i];
      boff[i] = (unsigned)n * (unsigned)(p.K * 4) + lkb[i];
    }
  };
  const int nk = p.K / BK;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 1024u;
  auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
    unsigned m0_saved;   // M0 is restored: the compiler may keep a value of its own there (clobbering it is not allowed)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  const unsigned long long pa = (unsigned long long)p.A, pb = (unsigned long long)p.Wt;
  int l_u = first, l_k0 = 0;
  offsets(l_u);
  auto gload = [&](int stage) __attribute__((always_inline)) {
    const i32x4 a_rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                          __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xFFFF)), 0x7FFFFFFF, 0x00020000};
    const i32x4 b_rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pb),
                          __builtin_amdgcn_readfirstlane((int)((pb >> 32) & 0xFFFF)), 0x7FFFFFFF, 0x00020000};
    const unsigned sa = lds0 + (3 * SB + stage * SA) * 4, sb = lds0 + stage * SB * 4;
    const int soff = __builtin_amdgcn_readfirstlane(l_k0 * 4);
#pragma unroll
    for (int i = 0; i < NLA; i++) dma(a_rsrc, sa + i * NT * 16, aoff[i], soff);
#pragma unroll
    for (int i = 0; i < 2; i++) dma(b_rsrc, sb + i * NT * 16, boff[i], soff);
    l_k0 += BK;
    if (l_k0 == p.K) {   // next unit of this block (past the last one: re-read it into a stage nobody reads)
      l_k0 = 0;
      if (l_u + stride < NUx) l_u += stride;
      offsets(l_u);
    }
  };

  int aaddr[4], baddr[4];
  {
    const int Ra = wm * 64 + (lane & 31), Rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int kg = 0; kg < 4; kg++) {
      aaddr[kg] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (2 * kg + h)) ^ ((Ra >> 1) & 15)) << 2);
      baddr[kg] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (2 * kg + h)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
  int kc = 0, c_u = first;
  const int act = p.act, half = lane >> 5;
  const int ntp = ntn * NWN;
  // The MFMAs take the weight fragment as their first operand, so the product sits transposed in the
  // accumulators: lane & 31 = row (m) of the 32-row sub-tile, register r = column 8 (r >> 2) + 4 half + (r & 3).
  // A lane then owns four consecutive columns of one row per register quad (16-byte stores), and the row's
  // arg-max over the wave's 32 columns is an in-lane scan plus one exchange between the half-waves.
  auto flush = [&]() __attribute__((always_inline)) {
    const int qm = c_u / gw, mt = mb + qm, nt = nb + c_u - qm * gw;
    const int col0 = nt * TBN + wn * 32 + 4 * half;
#pragma unroll
    for (int m2 = 0; m2 < 2; m2++) {
      const int row = mt * TBM + wm * 64 + m2 * 32 + (lane & 31);
      float bestv = -INFINITY;
      int besti = 0;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int col = col0 + 8 * g;
        const f32x4 bias = *(const f32x4*)(sbias + col);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v[e] = acc[m2][4 * g + e] + bias[e];
          acc[m2][4 * g + e] = 0.f;
          if (ARGMAX) {
            const float vv = col + e < p.N ? v[e] : -INFINITY;
            const bool up = vv > bestv;          // ascending columns, strict: the lowest column keeps a tie
            bestv = up ? vv : bestv;
            besti = up ? col + e : besti;
          }
        }
        if (!ARGMAX || p.C) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float neg = act == 2 ? 0.2f * v[e] : (act == 1 ? 0.f : v[e]);
            v[e] = v[e] > 0.f ? v[e] : neg;
          }
          if (row < p.M) {
            float* dst = p.C + (long)row * p.ldc + col;
            if (col + 3 < p.N) *(f32x4u*)dst = v;
            else {
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (col + e < p.N) dst[e] = v[e];
            }
          }
        }
      }
      if (ARGMAX) {
        const float ov = __shfl_xor(bestv, 32);
        const int oi = __shfl_xor(besti, 32);
        const bool take = ov > bestv || (ov == bestv && oi < besti);
        bestv = take ? ov : bestv;
        besti = take ? oi : besti;
        if (half == 0 && row < p.M) {
          p.pmax[(long)row * ntp + nt * NWN + wn] = bestv;
          p.pidx[(long)row * ntp + nt * NWN + wn] = besti;
        }
      }
    }
    kc = 0;
    c_u += stride;
  };

  f32x4 a0[2], a1[2], b0[2];
  auto frag = [&](int stage, int kg, int set) __attribute__((always_inline)) {
    const float* sa = smem + 3 * SB + stage * SA;
    const float* sb = smem + stage * SB;
    a0[set] = *(const f32x4*)(sa + aaddr[kg]);
    a1[set] = *(const f32x4*)(sa + aaddr[kg] + 1024);
    b0[set] = *(const f32x4*)(sb + baddr[kg]);
  };
  auto mfma8 = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[set][j], a0[set][j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[set][j], a1[set][j], acc[1], 0, 0, 0);
    }
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, nx = (st + 1) % 3;
    frag(st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(0);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1);
    __builtin_amdgcn_sched_barrier(0);
    frag(st, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070 | (NLA + 2));   // vmcnt(one step's loads), lgkmcnt(0)
    __syncthreads();
    gload(st);
    frag(nx, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
  };

  const int nit = my_units * nk;
  gload(0);
  gload(1);
  gload(2);
  constexpr int LPS = NLA + 2;
  __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * LPS));
  __syncthreads();
  frag(0, 0, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  __builtin_amdgcn_s_waitcnt(0x0F70);     // no LDS write may be in flight when the block ends
}

// The same kernel on operands in the f16x3 format (A converted by rows_to_f16x3_kernel with one scale per 32-row block
// = per unit, rs[m-tile] = 1 / scale; W converted at load time with one scale, us = 1 / scale):
// C = act((hi hi + hi lo + lo hi) rs us + bias).
__global__ __launch_bounds__(512, 1) void lin_gemm_h3_kernel(const ConvGemm p) {
  constexpr int NWN = 4, NT = 512, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32, NLA = 2;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];   // [B0 B1 B2 | A0 A1 A2]
  __shared__ float sbias[LIN_MAXN];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int ntn = (p.N + TBN - 1) / TBN, ntm = (p.M + TBM - 1) / TBM;
  const int NU = ntm * ntn, G = gridDim.x;
  // Which units this block takes.  Small launches (G < 256): unit L + k G, L the XCD-aware block order.
  // Full launches: the 8 XCDs split the unit grid into gn column groups x 8/gn row groups, and the G/8 blocks
  // of an XCD stride through their rectangle row-major -- the rectangle's slice of W stays in that XCD's L2
  // and an A tile is fetched once per column group instead of once per XCD and n-tile.
  int nb, ne, gw, mb, me, first, stride, my_units;
  if (p.gn > 0) {   // (set by the launcher only for a full launch: G a multiple of 8, every XCD's rectangle >= G / 8 units)
    const int x = blockIdx.x & 7, gm = 8 / p.gn, cg = x % p.gn, rg = x / p.gn;
    nb = cg * ntn / p.gn; ne = (cg + 1) * ntn / p.gn;
    mb = rg * ntm / gm; me = (rg + 1) * ntm / gm;
    first = blockIdx.x >> 3; stride = G >> 3;
  } else {
    const int chunk = G >> 3;
    nb = 0; ne = ntn; mb = 0; me = ntm;
    first = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3); stride = G;
  }
  gw = ne - nb;
  const int NUx = gw * (me - mb);
  if (first >= NUx) return;
  my_units = (NUx - first + stride - 1) / stride;
  (void)NU;
  for (int i = tid; i < ntn * TBN; i += NT) sbias[i] = i < p.N ? p.bias[i] : 0.f;
  __syncthreads();

  // loader: local row and k-chunk byte offset of this lane's LDS chunk positions
  int lrow[2];
  unsigned lkb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int P = i * NT + tid, S = P >> 4, c16 = (P & 15) ^ (S & 15);
    lrow[i] = 2 * S + (c16 >> 3);
    lkb[i] = (unsigned)((c16 & 7) * 16);
  }
  unsigned aoff[2], boff[2];
  auto offsets = [&](int q) __attribute__((always_inline)) {
    const int qm = q / gw, mt = mb + qm, nt = nb + q - qm * gw;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int m = mt * TBM + lrow[i], n = nt * TBN + lrow[i];
      m = m < p.M ? m : p.M - 1;
      n = n < p.N ? n : p.N - 1;
      aoff[i] = (unsigned)m * (unsigned)(p.K * 4) + lkb[i];
      boff[i] = (unsigned)n * (unsigned)(p.K * 4) + lkb[i];
    }
  };
  const int nk = p.K / BK;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 1024u;
  auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
    unsigned m0_saved;   // M0 is restored: the compiler may keep a value of its own there (clobbering it is not allowed)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  const unsigned long long pa = (unsigned long long)p.A, pb = (unsigned long long)p.Wt;
  int l_u = first, l_k0 = 0;
  offsets(l_u);
  auto gload = [&](int stage) __attribute__((always_inline)) {
    const i32x4 a_rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                          __builtin_amdgcn_readfirstlane((int)((pa >> 32) & 0xFFFF)), 0x7FFFFFFF, 0x00020000};
    const i32x4 b_rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pb),
                          __builtin_amdgcn_readfirstlane((int)((pb >> 32) & 0xFFFF)), 0x7FFFFFFF, 0x00020000};
    const unsigned sa = lds0 + (3 * SB + stage * SA) * 4, sb = lds0 + stage * SB * 4;
    const int soff = __builtin_amdgcn_readfirstlane(l_k0 * 4);
#pragma unroll
    for (int i = 0; i < NLA; i++) dma(a_rsrc, sa + i * NT * 16, aoff[i], soff);
#pragma unroll
    for (int i = 0; i < 2; i++) dma(b_rsrc, sb + i * NT * 16, boff[i], soff);
    l_k0 += BK;
    if (l_k0 == p.K) {   // next unit of this block (past the last one: re-read it into a stage nobody reads)
      l_k0 = 0;
      if (l_u + stride < NUx) l_u += stride;
      offsets(l_u);
    }
  };

  int ahi[2], alo[2], bhi[2], blo[2];
  {
    const int Ra = wm * 64 + (lane & 31), Rb = wn * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = 4 * j + 2 * h;
      ahi[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | c) ^ ((Ra >> 1) & 15)) << 2);
      alo[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (c + 1)) ^ ((Ra >> 1) & 15)) << 2);
      bhi[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | c) ^ ((Rb >> 1) & 15)) << 2);
      blo[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (c + 1)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
  int kc = 0, c_u = first;
  const int act = p.act, half = lane >> 5;
  // The MFMAs take the weight fragment as their first operand, so the product sits transposed in the
  // accumulators: lane & 31 = row (m) of the 32-row sub-tile, register r = column 8 (r >> 2) + 4 half + (r & 3).
  // A lane then owns four consecutive columns of one row per register quad (16-byte stores), and the row's
  // arg-max over the wave's 32 columns is an in-lane scan plus one exchange between the half-waves.
  const float us = p.us;
  auto flush = [&]() __attribute__((always_inline)) {
    const int qm = c_u / gw, mt = mb + qm, nt = nb + c_u - qm * gw;
    const int col0 = nt * TBN + wn * 32 + 4 * half;
#pragma unroll
    for (int m2 = 0; m2 < 2; m2++) {
      const float r = p.rs[(mt * TBM + wm * 64 + m2 * 32) >> 5] * us;   // one scale per 32 rows: a scalar load (not in vmcnt)
      const int row = mt * TBM + wm * 64 + m2 * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int col = col0 + 8 * g;
        const f32x4 bias = *(const f32x4*)(sbias + col);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v[e] = acc[m2][4 * g + e] * r + bias[e];
          acc[m2][4 * g + e] = 0.f;
          const float neg = act == 2 ? 0.2f * v[e] : (act == 1 ? 0.f : v[e]);
          v[e] = v[e] > 0.f ? v[e] : neg;
        }
        if (row < p.M) {
          float* dst = p.C + (long)row * p.ldc + col;
          if (col + 3 < p.N) *(f32x4u*)dst = v;
          else {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (col + e < p.N) dst[e] = v[e];
          }
        }
      }
    }
    kc = 0;
    c_u += stride;
  };

  f16x8 a0h[2], a0l[2], a1h[2], a1l[2], bh[2], bl[2];
  auto frag = [&](int stage, int j, int set) __attribute__((always_inline)) {
    const float* sa = smem + 3 * SB + stage * SA;
    const float* sb = smem + stage * SB;
    a0h[set] = *(const f16x8*)(sa + ahi[j]);
    a0l[set] = *(const f16x8*)(sa + alo[j]);
    a1h[set] = *(const f16x8*)(sa + ahi[j] + 1024);
    a1l[set] = *(const f16x8*)(sa + alo[j] + 1024);
    bh[set] = *(const f16x8*)(sb + bhi[j]);
    bl[set] = *(const f16x8*)(sb + blo[j]);
  };
  auto mfma6 = [&](int set) __attribute__((always_inline)) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[set], a0h[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[set], a1h[set], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a0l[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a1l[set], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a0h[set], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[set], a1h[set], acc[1], 0, 0, 0);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, nx = (st + 1) % 3;
    frag(st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070 | (NLA + 2));   // vmcnt(one step's loads), lgkmcnt(0)
    __syncthreads();
    gload(st);
    frag(nx, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
  };

  const int nit = my_units * nk;
  gload(0);
  gload(1);
  gload(2);
  constexpr int LPS = NLA + 2;
  __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * LPS));
  __syncthreads();
  frag(0, 0, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  __builtin_amdgcn_s_waitcnt(0x0F70);     // no LDS write may be in flight when the block ends
}

bool lin_gemm_ok(const ConvGemm& p) {
  // plain GEMM, K a multiple of 32, byte offsets below 2 GiB, bias of all n-tiles fits in LDS
  const long ntn = (p.N + 127) / 128;
  return p.ks == 1 && p.nbatch <= 1 && !p.wvalid && p.K % 32 == 0 && p.M > 0 && p.N > 0 &&
         (double)p.M * p.K * 4 < 2147000000.0 && (double)p.N * p.K * 4 < 2147000000.0 && ntn * 128 <= LIN_MAXN;
}

void launch_lin_gemm(const ConvGemm& p0, hipStream_t s) {
  ConvGemm p = p0;
  const int ntn = (p.N + 127) / 128, ntm = (p.M + 127) / 128, units = ntm * ntn;
  const int g = units < persistent_blocks() ? 8 * ((units + 7) / 8) : persistent_blocks();
  // column groups: the fewest whose slice of W (N/gn x K floats) fits an XCD's 4 MB L2 with room for A tiles,
  // provided every XCD's rectangle still has at least one unit per block
  p.gn = 0;
  static const int force = getenv("CRNN_LIN_GN") ? atoi(getenv("CRNN_LIN_GN")) : -1;
  if (g == persistent_blocks() && g % 8 == 0) {
    for (int gn = 1; gn <= 8; gn *= 2) {
      const double slice = (double)p.N * p.K * 4 / gn;
      if ((slice <= 3.0e6 || gn == 8) && ntn >= gn && ntm >= 8 / gn && (ntn / gn) * (ntm / (8 / gn)) >= g / 8) { p.gn = gn; break; }
    }
    if (force >= 0) p.gn = force;
  }
  if (p.h3) hipLaunchKernelGGL(lin_gemm_h3_kernel, dim3(g), dim3(512), 0, s, p);   // no arg-max form (checked by the caller)
  else if (p.pmax) hipLaunchKernelGGL((lin_gemm_kernel<true>), dim3(g), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((lin_gemm_kernel<false>), dim3(g), dim3(512), 0, s, p);
}


// Which launches take the ping-pong form: 256-tile units (wino4_pad rounds to 256 from 1024 tiles up).  Measured alone on
// the device at 256 lines against the loader-wave form: conv4 0.162 against 0.181 ms, conv5 0.250 against 0.277, but conv3
// (N = 256: a unit is only 8 K steps before its 64 store instructions per wave) 0.193 against 0.169 and conv6 (K = 1024, 5
// components) 0.090 against 0.083 -- so it takes the layers with N >= 512 and K <= 512.  In the pipeline the two forms are
// within 0.5 % of each other (the chip is power-bound there).  CRNN_GEMM_PP=0: never; 2: every layer it can take.
// (Read per launch: bench.py switches it between handles for its A/B line.)
bool wino4_gemm_uses_pp(const Wino4Gemm& p) {
  const char* ppe = getenv("CRNN_GEMM_PP");
  const int ppm = ppe ? atoi(ppe) : 1;
  return p.h3 && ppm != 0 && p.Cin >= 256 && p.ntile_pad >= 1024 && p.ntile_pad % 256 == 0 && p.N % 128 == 0 &&
         (ppm == 2 || (p.N >= 512 && p.Cin <= 512));
}

bool lin_gemm_pp_ok(int rows, int N, int K) {
  const char* e = getenv("CRNN_LIN_PP");   // 0: the plain GEMMs stay on lin_gemm_h3_kernel (read per launch)
  if (e && atoi(e) == 0) return false;
  return rows >= 1024 && rows % 256 == 0 && N % 128 == 0 && K % 64 == 0 && K >= 256 &&
         (double)rows * K * 4.0 < 2.0e9 && (double)rows * N * 4.0 < 2.0e9;
}
void launch_lin_gemm_pp(const Wino4Gemm& p, hipStream_t s) {
  const int u2 = (p.ntile_pad / 256) * (p.N / 128);
  hipLaunchKernelGGL(wino4_gemm_h3_pp_kernel, dim3(u2 < persistent_blocks() ? 8 * ((u2 + 7) / 8) : persistent_blocks()), dim3(768), 0, s, p);
}

void launch_wino4_gemm(const Wino4Gemm& p, hipStream_t s) {
  // units of 128 tiles x 128 channels (N % 128 == 0 checked by the caller), one persistent block per CU slot
  const int u = p.ncomp * (p.ntile_pad / 128) * (p.N / 128);
  const int g = u < persistent_blocks() ? 8 * ((u + 7) / 8) : persistent_blocks();
  if (!p.h3) {   // f32 MFMA (CRNN_F16X3=0)
    hipLaunchKernelGGL(wino4_gemm_lw_kernel, dim3(g), dim3(576), 0, s, p);
  } else if (wino4_gemm_uses_pp(p)) {
    const int u2 = p.ncomp * (p.ntile_pad / 256) * (p.N / 128);
    hipLaunchKernelGGL(wino4_gemm_h3_pp_kernel, dim3(u2 < persistent_blocks() ? 8 * ((u2 + 7) / 8) : persistent_blocks()), dim3(768), 0, s, p);
  } else {
    hipLaunchKernelGGL((wino4_gemm_h3_lw_kernel<4>), dim3(g), dim3(768), 0, s, p);
  }
}
void launch_wino4_in(const float* x, float* V, int B, int H, int W, int C, hipStream_t s, float* vs) {
  const long total = wino4_tiles(B, H, W) * (C / 4);
  if (vs)   // f16x3 format + per-tile scales; C / 4 must divide 256 (C = 64 ... 512, a power of two)
    hipLaunchKernelGGL(wino4_in_h3_kernel, dim3(ew_blocks_w(total)), dim3(256), 0, s, x, V, vs, B, H, W, C,
                       wino4_pad(wino4_tiles(B, H, W)));
  else
    hipLaunchKernelGGL(wino4_in_kernel, dim3(ew_blocks_w(total)), dim3(256), 0, s, x, V, B, H, W, C,
                       wino4_pad(wino4_tiles(B, H, W)));
}
void launch_wino4_out(const float* M, const float* bias, float* y, const int32_t* wvalid, int B, int H, int W,
                      int Cout, int act, int pool, hipStream_t s, const float* vs, float us) {
  if (pool == 2) {
    const int tw = (W + 3) / 4, nseg = tw >= 16 ? tw / 8 : 1;   // segments of 8-9 tiles (one extra tile of work each)
    const long rows = (long)B * (H / 4) * nseg * (Cout / 4);
    hipLaunchKernelGGL(wino4_out_pool21_kernel, dim3(ew_blocks_w(rows)), dim3(256), 0, s, M, bias, y, wvalid, B, H, W,
                       Cout, act, wino4_pad(wino4_tiles(B, H, W)), nseg, vs, us);
    return;
  }
  const long total = wino4_tiles(B, H, W) * (Cout / 4);
  hipLaunchKernelGGL(wino4_out_kernel, dim3(ew_blocks_w(total)), dim3(256), 0, s, M, bias, y, wvalid, B, H, W, Cout,
                     act, pool, wino4_pad(wino4_tiles(B, H, W)), vs, us);
}

// ---------------------------------------------------------------------------
// conv6 (2x2, no padding, input height 2 -> output height 1) as 1-D Winograd F(4,2) along the width: 5 products
// per 4 outputs instead of 8 (Cook-Toom, points 0, 1, -1, 2, inf; derived and checked numerically):
//   B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
//   G   = [1/2 0; -1/2 -1/2; -1/6 1/6; 1/6 1/3; 0 1]          A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 0; 0 1 -1 8 1]
// The two kernel rows are folded into K: V6[5][tiles][ky * C + c], U6[5][Cout][ky * C + c]; the 5 GEMMs run on
// wino4_gemm_lw_kernel (ncomp = 5, K = 2 C).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino42_in_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                        int W, int C, long ntile_pad) {
  const int T = W - 1, tw = (T + 3) >> 2, C4 = C >> 2;
  const long total = (long)B * tw * 2 * C4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ky = (int)(r & 1);
    r >>= 1;
    const int tx = (int)(r % tw);
    const int b = (int)(r / tw);
    const float* px = x + (((long)b * 2 + ky) * W) * C + c4 * 4;
    f32x4 d[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int xx = 4 * tx + j;
      d[j] = xx < W ? *(const f32x4*)(px + (long)xx * C) : z;
    }
    const f32x4 d31 = d[3] - d[1];
    f32x4 v[5];
    v[0] = 2.f * (d[0] - d[2]) + d31;
    v[1] = d[3] - d[2] - 2.f * d[1];
    v[2] = 2.f * d[1] - 3.f * d[2] + d[3];
    v[3] = d31;
    v[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    const long t = (long)b * tw + tx;
    float* o = V + (t * 2 + ky) * C + c4 * 4;
#pragma unroll
    for (int k = 0; k < 5; k++) *(f32x4*)(o + (long)k * ntile_pad * 2 * C) = v[k];
  }
}

// f16x3 form of the input transform: a row of V6 is one tile (both kernel rows, all channels): with C = 512 that is
// exactly one 256-thread block iteration, so the row's scale is a block-wide maximum.
__global__ __launch_bounds__(256) void wino42_in_h3_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                           float* __restrict__ vs, int B, int W, int C, long ntile_pad) {
  __shared__ float wmax[2][4];
  const int T = W - 1, tw = (T + 3) >> 2, C4 = C >> 2;   // 2 * C4 == 256
  const long total = (long)B * tw * 2 * C4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  int par = 0;
  for (long base = (long)blockIdx.x * blockDim.x; base < total; base += (long)gridDim.x * blockDim.x, par ^= 1) {
    const long i = base + threadIdx.x;     // total is a multiple of 256: every thread of the block is valid
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ky = (int)(r & 1);
    r >>= 1;
    const int tx = (int)(r % tw);
    const int b = (int)(r / tw);
    const float* px = x + (((long)b * 2 + ky) * W) * C + c4 * 4;
    f32x4 d[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int xx = 4 * tx + j;
      d[j] = xx < W ? *(const f32x4*)(px + (long)xx * C) : z;
    }
    const f32x4 d31 = d[3] - d[1];
    f32x4 v[5];
    v[0] = 2.f * (d[0] - d[2]) + d31;
    v[1] = d[3] - d[2] - 2.f * d[1];
    v[2] = 2.f * d[1] - 3.f * d[2] + d[3];
    v[3] = d31;
    v[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
      for (int e = 0; e < 4; e++) mx = fmaxf(mx, __builtin_fabsf(v[k][e]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) wmax[par][threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wmax[par][0], wmax[par][1]), fmaxf(wmax[par][2], wmax[par][3]));
    float sc, uns;
    split_scale_from_max(mx, sc, uns);
    const long t = (long)b * tw + tx;
    if (threadIdx.x == 0) vs[t] = uns;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      uint2 hi, lo;
      split4(v[k], sc, hi, lo);
      store_split_pair(V + (long)k * ntile_pad * 2 * C + (t * 2 + ky) * C, c4, hi, lo);
    }
  }
}

__global__ __launch_bounds__(256) void wino42_out_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                         float* __restrict__ y, int B, int T, int Co, int act,
                                                         long ntile_pad, const float* __restrict__ vs, float us) {
  const int tw = (T + 3) >> 2, C4 = Co >> 2;
  const long total = (long)B * tw * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const int b = (int)(t / tw);
    f32x4 m[5];
#pragma unroll
    for (int k = 0; k < 5; k++) m[k] = __builtin_nontemporal_load((const f32x4*)(Mo + ((long)k * ntile_pad + t) * Co + c4 * 4));
    const f32x4 bs = *(const f32x4*)(bias + c4 * 4);
    const float rs = vs ? vs[t] * us : 1.f;
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
    f32x4 o[4];
    o[0] = m[0] + s12 + m[3];
    o[1] = d12 + 2.f * m[3];
    o[2] = s12 + 4.f * m[3];
    o[3] = d12 + 8.f * m[3] + m[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int xx = 4 * tx + j;
      if (xx >= T) continue;
      f32x4 v = o[j] * rs + bs;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float e = v[c];
        if (act == 1) e = e > 0.f ? e : 0.f;
        else if (act == 2) e = e > 0.f ? e : 0.2f * e;
        v[c] = e;
      }
      *(f32x4*)(y + ((long)b * T + xx) * Co + c4 * 4) = v;
    }
  }
}

// conv6's output transform AND the f16x3 conversion of its rows (the A operand of the first LSTM input projection) in
// one kernel: the fp32 sequence tensor is written only on request (y), and rows_to_f16x3_kernel's launch and its read
// of that tensor go away.  Workgroup = one 32-row block of the GEMM operand (its scale is the block's maximum):
// 128 threads x 4 channels for each of two rows per pass, 16 passes; a thread computes the one output column of the
// tile its row needs (the tile's five products are read by the four rows that share it: L1 hits).  The same
// expressions on the same values as wino42_out_kernel + rows_to_f16x3_kernel<512>: identical bits (tests/test_hip_chain.py).
__global__ __launch_bounds__(256) void wino42_out_split_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                               float* __restrict__ y, float* __restrict__ a3,
                                                               float* __restrict__ rs, int B, int T, int act,
                                                               long ntile_pad, const float* __restrict__ vs, float us) {
  constexpr int Co = 512;
  __shared__ float wmax[4];
  const int c4 = threadIdx.x & 127, rsub = threadIdx.x >> 7, tw = (T + 3) >> 2;
  const long rows = (long)B * T, r0 = (long)blockIdx.x * F16X3_ROWS;
  const f32x4 bs = *(const f32x4*)(bias + c4 * 4);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 val[F16X3_ROWS / 2];
  float mx = 0.f;
#pragma unroll
  for (int it = 0; it < F16X3_ROWS / 2; it++) {
    const long row = r0 + it * 2 + rsub;
    val[it] = z;
    if (row < rows) {
      const int b = (int)(row / T), xx = (int)(row - (long)b * T), j = xx & 3;   // (uniform over a wave)
      const long t = (long)b * tw + (xx >> 2);
      f32x4 m[5];
#pragma unroll
      for (int k = 0; k < 5; k++) m[k] = *(const f32x4*)(Mo + ((long)k * ntile_pad + t) * Co + c4 * 4);
      const float r = vs[t] * us;
      const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
      f32x4 o;
      if (j == 0) o = m[0] + s12 + m[3];
      else if (j == 1) o = d12 + 2.f * m[3];
      else if (j == 2) o = s12 + 4.f * m[3];
      else o = d12 + 8.f * m[3] + m[4];
      f32x4 v = o * r + bs;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float e = v[c];
        if (act == 1) e = e > 0.f ? e : 0.f;
        else if (act == 2) e = e > 0.f ? e : 0.2f * e;
        v[c] = e;
        mx = fmaxf(mx, __builtin_fabsf(e));
      }
      val[it] = v;
      if (y) *(f32x4*)(y + row * Co + c4 * 4) = v;
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  float sc, uns;
  split_scale_from_max(mx, sc, uns);
  if (threadIdx.x == 0) rs[blockIdx.x] = uns;
#pragma unroll
  for (int it = 0; it < F16X3_ROWS / 2; it++) {
    const long row = r0 + it * 2 + rsub;
    if (row >= rows) continue;
    uint2 hi, lo;
    split4(val[it], sc, hi, lo);
    *(i32x4*)(a3 + row * Co + (c4 >> 1) * 8 + (c4 & 1) * 4) = split_pair_chunk(c4, hi, lo);
  }
}
void launch_wino42_out_split(const float* M, const float* bias, float* y, float* a3, float* rs, int B, int T, int act,
                             hipStream_t s, const float* vs, float us) {
  const long tiles = (long)B * ((T + 3) / 4), rows = (long)B * T;
  const unsigned g = (unsigned)((rows + 127) / 128 * (128 / F16X3_ROWS));   // whole 128-row GEMM units, as rows_to_f16x3_kernel
  hipLaunchKernelGGL(wino42_out_split_kernel, dim3(g), dim3(256), 0, s, M, bias, y, a3, rs, B, T, act, wino4_pad(tiles), vs, us);
}

void launch_wino42_in(const float* x, float* V, int B, int W, int C, hipStream_t s, float* vs) {
  const long tiles = (long)B * ((W - 1 + 3) / 4);
  if (vs && C == 512)
    hipLaunchKernelGGL(wino42_in_h3_kernel, dim3(ew_blocks_w(tiles * 2 * (C / 4))), dim3(256), 0, s, x, V, vs, B, W, C,
                       wino4_pad(tiles));
  else
    hipLaunchKernelGGL(wino42_in_kernel, dim3(ew_blocks_w(tiles * 2 * (C / 4))), dim3(256), 0, s, x, V, B, W, C,
                       wino4_pad(tiles));
}
void launch_wino42_out(const float* M, const float* bias, float* y, int B, int T, int Cout, int act, hipStream_t s,
                       const float* vs, float us) {
  const long tiles = (long)B * ((T + 3) / 4);
  hipLaunchKernelGGL(wino42_out_kernel, dim3(ew_blocks_w(tiles * (Cout / 4))), dim3(256), 0, s, M, bias, y, B, T, Cout,
                     act, wino4_pad(tiles), vs, us);
}

// ---------------------------------------------------------------------------
// conv0 + ReLU + pool0, direct.  16 pooled pixels x 16 channel groups (4 ch)
// per 256-thread block; each thread keeps its 9x4 weights in registers, reads a
// 4x4 input patch and writes one float4 of the NHWC output (256 B per pixel).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_pool_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ w9x64,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out,
                                                         const int32_t* __restrict__ w1valid, int B,
                                                         int W, int act) {
  const int W1 = W >> 1;
  const int cg = threadIdx.x & 15;
  const long npix = (long)B * 16 * W1;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; t++) wr[t] = *(const f32x4*)(w9x64 + t * 64 + cg * 4);
  const f32x4 bs = *(const f32x4*)(bias + cg * 4);
  for (long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < npix; pix += (long)gridDim.x * 16) {
    const int b = (int)(pix / (16 * W1));
    const int r = (int)(pix - (long)b * 16 * W1);
    const int y1 = r / W1, x1 = r - y1 * W1;
    const float* ip = in + (long)b * 32 * W;
    float patch[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; dy++) {
      const int yy = 2 * y1 - 1 + dy;
#pragma unroll
      for (int dx = 0; dx < 4; dx++) {
        const int xx = 2 * x1 - 1 + dx;
        const bool ok = (unsigned)yy < 32u && (unsigned)xx < (unsigned)W;
        patch[dy][dx] = ok ? ip[yy * W + xx] : 0.f;
      }
    }
    f32x4 best;
#pragma unroll
    for (int py = 0; py < 2; py++)
#pragma unroll
      for (int px = 0; px < 2; px++) {
        f32x4 a = bs;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const float v = patch[py + ky][px + kx];
            a += wr[ky * 3 + kx] * v;
          }
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
    if (w1valid && x1 >= w1valid[b]) best = f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(out + pix * 64 + cg * 4) = best;
  }
}

void launch_conv0_pool(const float* in, const float* w9x64, const float* bias, float* out,
                       const int32_t* w1valid, int B, int W, int act, hipStream_t s) {
  long npix = (long)B * 16 * (W >> 1);
  long blocks = (npix + 15) / 16;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(conv0_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, w9x64, bias, out,
                     w1valid, B, W, act);
}

// ---------------------------------------------------------------------------
// NHWC max pools (float4 over channels)
// ---------------------------------------------------------------------------
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
  f32x4 r;
#pragma unroll
  for (int c = 0; c < 4; c++) r[c] = a[c] > b[c] ? a[c] : b[c];
  return r;
}

__global__ __launch_bounds__(256) void pool_2x2s2_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out, int B, int H, int W,
                                                         int C) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long pix = i / C4;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float* p = in + (((long)b * H + 2 * y) * W + 2 * x) * C + c4 * 4;
    f32x4 v = max4(max4(*(const f32x4*)p, *(const f32x4*)(p + C)),
                   max4(*(const f32x4*)(p + (long)W * C), *(const f32x4*)(p + (long)W * C + C)));
    *(f32x4*)(out + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void pool_2x2s21p01_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out, int B, int H,
                                                             int W, int C) {
  const int Ho = H >> 1, Wo = W + 1, C4 = C >> 2;
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long pix = i / C4;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float* r0 = in + (((long)b * H + 2 * y) * W) * C + c4 * 4;
    const float* r1 = r0 + (long)W * C;
    f32x4 v;
    if (x < W) {
      v = max4(*(const f32x4*)(r0 + (long)x * C), *(const f32x4*)(r1 + (long)x * C));
      if (x > 0) v = max4(v, max4(*(const f32x4*)(r0 + (long)(x - 1) * C), *(const f32x4*)(r1 + (long)(x - 1) * C)));
    } else {
      v = max4(*(const f32x4*)(r0 + (long)(x - 1) * C), *(const f32x4*)(r1 + (long)(x - 1) * C));
    }
    *(f32x4*)(out + i * 4) = v;
  }
}

static unsigned ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

void launch_pool_2x2s2(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  long total = (long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(pool_2x2s2_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, in, out, B, H, W, C);
}
void launch_pool_2x2s21p01(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  long total = (long)B * (H / 2) * (W + 1) * (C / 4);
  hipLaunchKernelGGL(pool_2x2s21p01_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, in, out, B, H, W, C);
}

// ---------------------------------------------------------------------------
// LSTM recurrence.  One 256-thread workgroup per (direction, 16 samples); no
// cross-workgroup communication.  Per step gates^T[1024 x 16] = W_hh[1024x256]
// . h^T[256x16] on v_mfma_f32_16x16x4_f32: wave w owns hidden units
// [64w, 64w+64) of all four gates (16 MFMA row tiles), so i,f,g,o of one
// (unit, sample) land in the same lane and the cell update needs no exchange.
// W_hh streams from L2 in a pre-packed per-lane fragment order (1 KiB per
// wave-instruction); h lives in LDS (double buffered), c in registers.
// ---------------------------------------------------------------------------
constexpr int LB = 16, HS = 260;

// Gate nonlinearities on the hardware exp2 / rcp units (about 1 ulp each): the libm versions
// cost ~10x more VALU work, and with one wave per SIMD that work is serial with the MFMAs
// (measured: 25 us -> 15 us per step).  Absolute error < 3e-7, far inside the 1e-3 logit bound.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  const float t = __builtin_amdgcn_exp2f(-2.885390082f * __builtin_fabsf(x));   // exp(-2|x|)
  const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return __builtin_copysignf(r, x);
}

__global__ __launch_bounds__(256) void lstm_rec_kernel(const float* __restrict__ xp,
                                                       const float* __restrict__ whhp,
                                                       float* __restrict__ rec,
                                                       const int32_t* __restrict__ tlen, int B, int T) {
  CRNN_WHOLE_REGISTER_FILE();
  __shared__ __attribute__((aligned(16))) float hs[2 * LB * HS];
  const int dir = blockIdx.y, b0 = blockIdx.x * LB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, bl = lane & 15;
  const int b = b0 + bl;
  const int Ti = (b < B) ? (tlen ? tlen[b] : T) : 0;
  int Tmax = 0;
  for (int j = 0; j < LB; j++) {
    int bb = b0 + j;
    int tj = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    Tmax = tj > Tmax ? tj : Tmax;
  }
  for (int i = tid; i < 2 * LB * HS; i += 256) hs[i] = 0.f;
  float c[4][4], h[4][4];
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int r = 0; r < 4; r++) { c[u][r] = 0.f; h[u][r] = 0.f; }
  const f32x4* wp = (const f32x4*)whhp + ((size_t)(dir * 4 + wave) * 16 * 16) * 64 + lane;
  f32x4 wa[16], wb[16];
#pragma unroll
  for (int tile = 0; tile < 16; tile++) wa[tile] = wp[(size_t)(tile * 16) * 64];
  __syncthreads();
  int cur = 0;
  for (int s = 0; s < Tmax; s++) {
    const bool active = s < Ti;
    const int t = active ? (dir ? (Ti - 1 - s) : s) : 0;
    const size_t rowi = (size_t)(b < B ? b : 0) * T + t;
    const float* xrow = xp + rowi * 2048 + dir * 1024 + wave * 64 + q * 4;
    // x-projection of this step: loaded now, added after the MFMAs (latency hidden under them)
    f32x4 xg[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        xg[g][u] = active ? *(const f32x4*)(xrow + g * 256 + u * 16) : z;
      }
    f32x4 acc[4][4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int u = 0; u < 4; u++) acc[g][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* hrow = hs + cur * LB * HS + bl * HS + 4 * q;
    // W_hh fragments run one k-group (16 x 1 KiB per wave) ahead of the MFMAs in two named
    // register sets; the last refill of set A wraps to k-group 0 = the first group of the NEXT
    // step, so it lands under the gate math and the barrier.
#pragma unroll 1
    for (int kg2 = 0; kg2 < 8; kg2++) {
#pragma unroll
      for (int tile = 0; tile < 16; tile++) wb[tile] = wp[(size_t)(tile * 16 + 2 * kg2 + 1) * 64];
      {
        const f32x4 hv = *(const f32x4*)(hrow + (2 * kg2) * 16);
#pragma unroll
        for (int j = 0; j < 4; j++)   // j outer: back-to-back MFMAs hit different accumulators
#pragma unroll
          for (int tile = 0; tile < 16; tile++)
            acc[tile >> 2][tile & 3] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tile][j], hv[j], acc[tile >> 2][tile & 3], 0, 0, 0);
      }
#pragma unroll
      for (int tile = 0; tile < 16; tile++) wa[tile] = wp[(size_t)(tile * 16 + ((2 * kg2 + 2) & 15)) * 64];
      {
        const f32x4 hv = *(const f32x4*)(hrow + (2 * kg2 + 1) * 16);
#pragma unroll
        for (int j = 0; j < 4; j++)   // j outer: back-to-back MFMAs hit different accumulators
#pragma unroll
          for (int tile = 0; tile < 16; tile++)
            acc[tile >> 2][tile & 3] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wb[tile][j], hv[j], acc[tile >> 2][tile & 3], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int u = 0; u < 4; u++) acc[g][u] += xg[g][u];
    float* hn = hs + (cur ^ 1) * LB * HS + bl * HS + wave * 64 + q * 4;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      f32x4 hv4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (active) {
          const float ig = sigmoidf_(acc[0][u][r]);
          const float fg = sigmoidf_(acc[1][u][r]);
          const float gg = tanhf_(acc[2][u][r]);
          const float og = sigmoidf_(acc[3][u][r]);
          c[u][r] = fg * c[u][r] + ig * gg;
          h[u][r] = og * tanhf_(c[u][r]);
        }
        hv4[r] = h[u][r];
      }
      *(f32x4*)(hn + u * 16) = hv4;
      if (active) *(f32x4*)(rec + rowi * 512 + dir * 256 + wave * 64 + u * 16 + q * 4) = hv4;
    }
    __syncthreads();
    cur ^= 1;
  }
  // padding rows (t >= T_i) of this workgroup's lines and direction: zero, so that what follows (the f16x3 conversion takes
  // the maximum over blocks of 32 rows) never sees stale workspace contents
  for (int ln = 0; ln < LB; ln++) {
    const int bb = b0 + ln;
    if (bb >= B) break;
    const int Tl = tlen ? tlen[bb] : T;
    for (int r = Tl * 64 + tid; r < T * 64; r += 256)
      *(f32x4*)(rec + ((size_t)bb * T + (r >> 6)) * 512 + dir * 256 + (r & 63) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// ---------------------------------------------------------------------------
// The same recurrence on the fp16 MFMA in the f16x3 form (CRNN_F16X3; used by the pipelined device path and by host
// batches too large for the hidden-split kernel): W_hh as (hi, lo) fp16 pairs with one power-of-two scale, h (|h| < 1)
// with the fixed scale 2^14, three v_mfma_f32_16x16x32_f16 per fragment pair, fp32 accumulate -- fp32-accurate gates
// (tools/split_mfma_test.hip) with a matrix phase of 2.6 us instead of 13.7 us per step.  The step is then bound by
// streaming the 1 MiB of W_hh from L2 (same bytes as fp32).  Same work split as lstm_rec_kernel: one 256-thread
// workgroup per (direction, 16 lines), wave w owns hidden units [64 w, 64 w + 64) of all four gates (16 row tiles),
// c in registers, h in LDS (double buffered) as split pairs: line stride 1040 bytes, inside a line [k / 8 mod 4][k / 32]
// blocks of 8 hi + 8 lo halves, so the 16 lanes of a ds_read_b128 phase hit 16 different 16-byte bank groups.
// W_hh streams in the order pack_whh_h3 writes: per wave and k-step of 32, the 16 tiles' hi and lo fragments (1 KiB per
// wave-instruction), through two register sets of 8 tiles each, one half k-step ahead of the MFMAs.
// ---------------------------------------------------------------------------
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
constexpr int HLS = 1040;
// RES (round 3; CRNN_REC_RES=0 keeps the streaming form): a quarter of W_hh stays on the CU for the whole launch -- the
// wave's 32 fragments of k-step 7 in registers (128 VGPRs; 1 wave per SIMD, so the file is there) and its 32 fragments of
// k-step 6 in LDS (32 KiB per wave; h is then single-buffered, one more barrier per step, to make room: 16.6 + 128 KiB) --
// so a step streams 768 KiB instead of 1 MiB from L2, and the first fragments of the next step are fetched two k-steps
// earlier.  The MFMA sequence of every accumulator is unchanged (k-steps ascending; lo hi, hi lo, hi hi inside one), so the
// results are bit-identical to the streaming form and to lstm_rec_split_h3_kernel.
template <bool RES>
__global__ __launch_bounds__(256) void lstm_rec_h3_kernel(const float* __restrict__ xp, const uint4* __restrict__ whh3,
                                                          float* __restrict__ rec, const int32_t* __restrict__ tlen,
                                                          int B, int T, float unscale, const unsigned* __restrict__ gate) {
  CRNN_WHOLE_REGISTER_FILE();
  // gate != nullptr: run only if *gate != 0 (the repeat of a hidden-split launch that gave up, queued behind it by callers
  // that cannot look at the time-out word themselves: the asynchronous device path)
  if (gate && __builtin_nontemporal_load(gate) == 0u) return;
  if (gate && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd((unsigned*)gate + 1, 1u);   // counted (crnn_split_timeouts)
  constexpr int NHB = RES ? 1 : 2;                       // h images
  constexpr int WL = RES ? 4 * 34 * 1024 : 0;            // bytes of LDS-resident fragments
  __shared__ __attribute__((aligned(16))) unsigned char hs[NHB * LB * HLS + WL];
  const int dir = blockIdx.y, b0 = blockIdx.x * LB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, bl = lane & 15;
  const int b = b0 + bl;
  const int Ti = (b < B) ? (tlen ? tlen[b] : T) : 0;
  int Tmax = 0;
  for (int j = 0; j < LB; j++) {
    const int bb = b0 + j;
    const int tj = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    Tmax = tj > Tmax ? tj : Tmax;
  }
  for (int i = tid; i < NHB * LB * HLS / 4; i += 256) ((unsigned*)hs)[i] = 0u;
  float c[4][4], h[4][4];
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int r = 0; r < 4; r++) { c[u][r] = 0.f; h[u][r] = 0.f; }
  // fragment (ks, tile, part) of this wave at wp[((ks * 16 + tile) * 2 + part) * 64]
  const uint4* wp = whh3 + ((size_t)(dir * 4 + wave) * 8 * 16 * 2) * 64 + lane;
  uint4 wa[8][2], wb[8][2];
  uint4 wr[RES ? 15 : 1][2];                             // k-step 7, tiles 0..14, resident (tile 15: LDS)
  uint4* wl = (uint4*)(hs + NHB * LB * HLS) + wave * (34 * 64) + lane;   // k-step 6 (+ tile 15 of k-step 7) in LDS: fragment f at wl[f * 64]
  if (RES) {
#pragma unroll
    for (int t = 0; t < 15; t++) {
      wr[t][0] = wp[(size_t)((7 * 16 + t) * 2) * 64];
      wr[t][1] = wp[(size_t)((7 * 16 + t) * 2 + 1) * 64];
    }
#pragma unroll
    for (int f = 0; f < 32; f++) wl[f * 64] = wp[(size_t)(6 * 16 * 2 + f) * 64];
    wl[32 * 64] = wp[(size_t)((7 * 16 + 15) * 2) * 64];
    wl[33 * 64] = wp[(size_t)((7 * 16 + 15) * 2 + 1) * 64];
  }
#pragma unroll
  for (int t = 0; t < 8; t++) { wa[t][0] = wp[(size_t)(t * 2) * 64]; wa[t][1] = wp[(size_t)(t * 2 + 1) * 64]; }
  __syncthreads();
  int cur = 0;
  for (int s = 0; s < Tmax; s++) {
    const bool active = s < Ti;
    const int t = active ? (dir ? (Ti - 1 - s) : s) : 0;
    const size_t rowi = (size_t)(b < B ? b : 0) * T + t;
    const float* xrow = xp + rowi * 2048 + dir * 1024 + wave * 64 + q * 4;
    f32x4 xg[4][4];
    auto load_xg = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          xg[g][u] = active ? *(const f32x4*)(xrow + g * 256 + u * 16) : z;
        }
    };
    load_xg();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* hrow = hs + (RES ? 0 : cur) * LB * HLS + bl * HLS + q * 256;
    auto mfma8 = [&](const uint4 (&w)[8][2], int t0, f16x8r hh, f16x8r hl) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[i][1]), hh, acc[t0 + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; i++)
        acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[i][0]), hl, acc[t0 + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; i++)
        acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[i][0]), hh, acc[t0 + i], 0, 0, 0);
    };
    constexpr int NKS = RES ? 6 : 8;                     // streamed k-steps
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks++) {
      const f16x8r hh = *(const f16x8r*)(hrow + ks * 32), hl = *(const f16x8r*)(hrow + ks * 32 + 16);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        wb[i][0] = wp[(size_t)((ks * 16 + 8 + i) * 2) * 64];
        wb[i][1] = wp[(size_t)((ks * 16 + 8 + i) * 2 + 1) * 64];
      }
      mfma8(wa, 0, hh, hl);
      // the last refill wraps to k-step 0 of the NEXT step: it lands under the resident k-steps / the gate math
      const int kn = ks + 1 == NKS ? 0 : ks + 1;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        wa[i][0] = wp[(size_t)((kn * 16 + i) * 2) * 64];
        wa[i][1] = wp[(size_t)((kn * 16 + i) * 2 + 1) * 64];
      }
      mfma8(wb, 8, hh, hl);
    }
    if (RES) {
      // k-step 6 from LDS (eight tiles at a time through the set the loop has just freed), k-step 7 from registers; every
      // accumulator still sees k-step 6 before k-step 7.  The reads of a half are issued as one block (the compiler would
      // otherwise wait for each before its MFMA), and the second half's are in flight under the first half's k-step 7.
      const f16x8r hh6 = *(const f16x8r*)(hrow + 6 * 32), hl6 = *(const f16x8r*)(hrow + 6 * 32 + 16);
      f16x8r hh7, hl7;
      uint4 w15[2];          // tile 15 of k-step 7, from LDS
      auto mfma7 = [&](int t0) __attribute__((always_inline)) {
        auto W = [&](int t, int part) __attribute__((always_inline)) { return t < 15 ? wr[t < 15 ? t : 0][part] : w15[part]; };
#pragma unroll
        for (int i = 0; i < 8; i++)
          acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, W(t0 + i, 1)), hh7, acc[t0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; i++)
          acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, W(t0 + i, 0)), hl7, acc[t0 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; i++)
          acc[t0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, W(t0 + i, 0)), hh7, acc[t0 + i], 0, 0, 0);
      };
#pragma unroll
      for (int i = 0; i < 8; i++) { wb[i][0] = wl[(i * 2) * 64]; wb[i][1] = wl[(i * 2 + 1) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
      mfma8(wb, 0, hh6, hl6);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; i++) { wb[i][0] = wl[((8 + i) * 2) * 64]; wb[i][1] = wl[((8 + i) * 2 + 1) * 64]; }
      hh7 = *(const f16x8r*)(hrow + 7 * 32);
      hl7 = *(const f16x8r*)(hrow + 7 * 32 + 16);
      w15[0] = wl[32 * 64];
      w15[1] = wl[33 * 64];
      __builtin_amdgcn_sched_barrier(0);
      mfma7(0);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(wb, 8, hh6, hl6);
      mfma7(8);
      __syncthreads();     // single h image: every wave has read it before anybody writes the next one
    }
    unsigned char* hn = hs + (RES ? 0 : (cur ^ 1)) * LB * HLS + bl * HLS;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      f32x4 hv4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (active) {
          const float ig = sigmoidf_(acc[0 * 4 + u][r] * unscale + xg[0][u][r]);
          const float fg = sigmoidf_(acc[1 * 4 + u][r] * unscale + xg[1][u][r]);
          const float gg = tanhf_(acc[2 * 4 + u][r] * unscale + xg[2][u][r]);
          const float og = sigmoidf_(acc[3 * 4 + u][r] * unscale + xg[3][u][r]);
          c[u][r] = fg * c[u][r] + ig * gg;
          h[u][r] = og * tanhf_(c[u][r]);
        }
        hv4[r] = h[u][r];
      }
      // units k = 64 wave + 16 u + 4 q .. + 4 as split pairs (scale 2^14) into the next h image
      uint2 hi, lo;
      split4(hv4, 16384.f, hi, lo);
      const int ks = wave * 2 + (u >> 1), qq = (u & 1) * 2 + (q >> 1), e = (q & 1) * 4;
      unsigned char* d = hn + qq * 256 + ks * 32 + e * 2;
      *(uint2*)d = hi;
      *(uint2*)(d + 16) = lo;
      if (active) *(f32x4*)(rec + rowi * 512 + dir * 256 + wave * 64 + u * 16 + q * 4) = hv4;
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int ln = 0; ln < LB; ln++) {   // padding rows (t >= T_i): zero (see lstm_rec_kernel)
    const int bb = b0 + ln;
    if (bb >= B) break;
    const int Tl = tlen ? tlen[bb] : T;
    for (int r = Tl * 64 + tid; r < T * 64; r += 256)
      *(f32x4*)(rec + ((size_t)bb * T + (r >> 6)) * 512 + dir * 256 + (r & 63) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

void launch_lstm_rec_h3(const float* xp, const void* whh3, float* rec, const int32_t* tlen, int B, int T, float unscale,
                        hipStream_t s, const unsigned* gate) {
  const char* re = getenv("CRNN_REC_RES");       // read per launch (a test switches it inside one process)
  const bool res = !(re && atoi(re) == 0);
  if (res)
    hipLaunchKernelGGL(lstm_rec_h3_kernel<true>, dim3((B + LB - 1) / LB, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen,
                       B, T, unscale, gate);
  else
    hipLaunchKernelGGL(lstm_rec_h3_kernel<false>, dim3((B + LB - 1) / LB, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen,
                       B, T, unscale, gate);
}

// W_hh [2][1024][256] -> the fragment order lstm_rec_h3_kernel streams, as (hi, lo) fp16 with ONE power-of-two scale;
// returns the factor the kernel multiplies its accumulators by: 1 / (scale_W * 2^14)
float pack_whh_h3(const float* whh_fwd, const float* whh_rev, uint16_t* packed /* 2 * 1024 * 256 * 2 halves */) {
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)1024 * 256; i++) mx = fmaxf(mx, fmaxf(fabsf(whh_fwd[i]), fabsf(whh_rev[i])));
  int e = 0;
  if (mx > 0.f && std::isfinite(mx)) (void)frexpf(mx, &e);
  const float sc = ldexpf(1.f, 14 - (e - 1));
  for (int dir = 0; dir < 2; dir++) {
    const float* w = dir ? whh_rev : whh_fwd;
    for (int wave = 0; wave < 4; wave++)
      for (int ks = 0; ks < 8; ks++)
        for (int tile = 0; tile < 16; tile++)
          for (int lane = 0; lane < 64; lane++) {
            const int g = tile >> 2, u = tile & 3;
            const int row = g * 256 + wave * 64 + u * 16 + (lane & 15);
            const int k0 = ks * 32 + 8 * (lane >> 4);
            uint16_t* dh = packed + ((((size_t)(dir * 4 + wave) * 8 + ks) * 16 + tile) * 2 * 64 + lane) * 8;
            uint16_t* dl = dh + 64 * 8;
            for (int j = 0; j < 8; j++) {
              const float x = w[(size_t)row * 256 + k0 + j] * sc;
              const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
              memcpy(dh + j, &hi, 2);
              memcpy(dl + j, &lo, 2);
            }
          }
  }
  return 1.f / (sc * 16384.f);
}

// ---------------------------------------------------------------------------
// Latency form of the recurrence for small batches (one crop per call, a page's worth of crops): the
// kernel above runs one workgroup per (direction, 16 lines) and is bound by its T sequential steps of
// ~21 us.  Here FOUR workgroups share a (direction, 16 lines) group: workgroup `part` owns hidden
// units [64 part, 64 part + 64) of all four gates, wave w the 16 units [.. + 16 w, ..+16); the wave's
// W_hh slice (4 gate tiles x 256 k = 64 KB) stays in registers for the whole launch, so a step is 256
// MFMAs per wave (3.4 us) instead of 1024 plus 1 MiB of weight traffic.  The price is an exchange of h
// every step, done through the output tensor itself (each step's rows are fresh addresses), and the data
// is its own arrival flag:
//   * the launcher fills the output tensor with the bit pattern 0xFFFFFFFF (no h value has it);
//   * every lane stores its 4 h values write-through (sc1) to rec[line, t, dir, unit..] (the kernel's output
//     anyway) and goes straight on to poll the three 16-byte chunks it needs from the other parts with sc1
//     loads (L1 bypassed) until all four words of each have changed -- no drain of the stores, no barrier, no
//     counter, and no assumption that the 16 bytes of a chunk arrive together; nothing here depends on which
//     XCD a workgroup landed on.  The chunks go into the LDS copy of h.
// A line that is past its own length publishes nothing; everybody keeps its old h from LDS.  When the loop is
// over the rows t >= T_i of every line (never written: still the fill pattern, a NaN) are set to zero.
// The same packed weights, the same MFMA sequence per accumulator and the same gate math as the kernel
// above: results are bit-identical (tests/test_hip_scale.py).  Spins are bounded (SPLIT_MAX_SPINS polls, some
// tens of milliseconds): a workgroup that waits too long sets *tmo and every workgroup leaves; the host then
// repeats the batch with the kernel above and counts the event (crnn_split_timeouts).  A weight that makes an
// h value the fill pattern itself (a NaN with all bits set) ends the same way.
// ---------------------------------------------------------------------------
constexpr unsigned SPLIT_MAX_SPINS = 1u << 16;
// tools/rec_bench.hip builds this file with -DREC_PROF: workgroup (0, 0)'s thread 0 sums the 100 MHz wall clock over the
// phases of its steps into tmo[16 ..] (not in the product build)
#ifdef REC_PROF
#define REC_PROF_DECL long long rp_t = 0, rp_sum[6] = {0, 0, 0, 0, 0, 0}; const bool rp_on = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0
#define REC_STAMP0() do { if (rp_on) rp_t = wall_clock64(); } while (0)
#define REC_STAMP(i) do { if (rp_on) { const long long n_ = wall_clock64(); rp_sum[i] += n_ - rp_t; rp_t = n_; } } while (0)
#define REC_COUNT(i, n) do { if (rp_on) rp_sum[i] += (n); } while (0)
#define REC_PROF_OUT() do { if (rp_on) for (int i_ = 0; i_ < 6; i_++) tmo[16 + i_] = (unsigned)rp_sum[i_]; } while (0)
#else
#define REC_PROF_DECL
#define REC_STAMP0()
#define REC_STAMP(i)
#define REC_COUNT(i, n)
#define REC_PROF_OUT()
#endif
typedef __attribute__((address_space(1))) unsigned gu32_t;


// ---------------------------------------------------------------------------
// The hidden-split recurrence (the work split, exchange protocol and time-out handling described above; its first, f32-MFMA
// form is gone) on the fp16 MFMA in the f16x3 form, i.e. with lstm_rec_h3_kernel's arithmetic.  Workgroup `part` owns hidden units [64 part, 64 part + 64) of
// all four gates, wave w the 16 units [.. + 16 w, ..); the wave's W_hh slice -- 4 gate tiles x 8 k-steps x (hi, lo)
// fragments in pack_whh_h3's order, 64 KB, the same 256 registers as the fp32 slice -- stays resident, h lives in LDS
// as split pairs (scale 2^14, lstm_rec_h3_kernel's layout), and a step is 96 v_mfma_f32_16x16x32_f16 per wave (0.7 us)
// instead of 256 v_mfma_f32_16x16x4_f32 (3.5 us).  The chunks that arrive from the other parts are fp32 (the output
// rows) and are split on the way into LDS.  Same MFMA sequence per accumulator and same gate math as
// lstm_rec_h3_kernel: bit-identical to it (tests/test_hip_scale.py).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_rec_split_h3_kernel(const float* __restrict__ xp,
                                                                const uint4* __restrict__ whh3, float* rec,
                                                                const int32_t* __restrict__ tlen, int B, int T,
                                                                float unscale, unsigned* tmo) {
  CRNN_WHOLE_REGISTER_FILE();
  __shared__ __attribute__((aligned(16))) unsigned char hs[2 * LB * HLS];
  __shared__ int s_fail;
  const int dir = blockIdx.y, grp = blockIdx.x >> 2, part = blockIdx.x & 3, b0 = grp * LB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, bl = lane & 15;
  const int b = b0 + bl;
  const int Ti = (b < B) ? (tlen ? tlen[b] : T) : 0;
  int Tmax = 0;
  for (int j = 0; j < LB; j++) {
    const int bb = b0 + j;
    const int tj = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    Tmax = tj > Tmax ? tj : Tmax;
  }
  for (int i = tid; i < 2 * LB * HLS / 4; i += 256) ((unsigned*)hs)[i] = 0u;
  if (tid == 0) s_fail = 0;
  // exchange loads: 3 other parts x 16 lines x 16 chunks of 4 units = 768 float4, 3 per thread
  int xl_line[3], xl_unit[3], xl_T[3], xl_lds[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = i * 256 + tid;            // 0..767
    const int op = e >> 8;                  // which of the 3 other parts
    const int p2 = op + (op >= part ? 1 : 0);
    xl_line[i] = (e >> 4) & 15;
    xl_unit[i] = p2 * 64 + (e & 15) * 4;
    const int bb = b0 + xl_line[i];
    xl_T[i] = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    const int k = xl_unit[i];               // units k .. k + 3: block [k / 8 mod 4][k / 32], halves (k & 7) .. + 3
    xl_lds[i] = xl_line[i] * HLS + ((k >> 3) & 3) * 256 + (k >> 5) * 32 + (k & 7) * 2;
  }
  const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, 0x7FFFFFFF, 0x00020000);
  float c[4], h[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { c[r] = 0.f; h[r] = 0.f; }
  // the wave's weights: gate g, its 16 units, all 8 k-steps, (hi, lo) -- 64 uint4 per lane, resident
  uint4 w[4][8][2];
  {
    const uint4* wp = whh3 + ((size_t)(dir * 4 + part) * 8 * 16 * 2) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        w[g][ks][0] = wp[(size_t)((ks * 16 + g * 4 + wave) * 2) * 64];
        w[g][ks][1] = wp[(size_t)((ks * 16 + g * 4 + wave) * 2 + 1) * 64];
      }
  }
  __syncthreads();
  int cur = 0;
  // where this lane's four units go in a line's split image
  const int myu = part * 64 + wave * 16 + q * 4;
  const int my_lds = bl * HLS + ((myu >> 3) & 3) * 256 + (myu >> 5) * 32 + (myu & 7) * 2;
  REC_PROF_DECL;
  for (int s = 0; s < Tmax; s++) {
    REC_STAMP0();
    const bool active = s < Ti;
    const int t = active ? (dir ? (Ti - 1 - s) : s) : 0;
    const size_t rowi = (size_t)(b < B ? b : 0) * T + t;
    const float* xrow = xp + rowi * 2048 + dir * 1024 + part * 64 + wave * 16 + q * 4;
    f32x4 xg[4], acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xg[g] = active ? *(const f32x4*)(xrow + g * 256) : z;
      acc[g] = z;
    }
    const unsigned char* hrow = hs + cur * LB * HLS + bl * HLS + q * 256;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const f16x8r hh = *(const f16x8r*)(hrow + ks * 32), hl = *(const f16x8r*)(hrow + ks * 32 + 16);
#pragma unroll
      for (int g = 0; g < 4; g++)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][1]), hh, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; g++)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][0]), hl, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; g++)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][0]), hh, acc[g], 0, 0, 0);
    }
    REC_STAMP(0);
    f32x4 hv4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (active) {
        const float ig = sigmoidf_(acc[0][r] * unscale + xg[0][r]);
        const float fg = sigmoidf_(acc[1][r] * unscale + xg[1][r]);
        const float gg = tanhf_(acc[2][r] * unscale + xg[2][r]);
        const float og = sigmoidf_(acc[3][r] * unscale + xg[3][r]);
        c[r] = fg * c[r] + ig * gg;
        h[r] = og * tanhf_(c[r]);
      }
      hv4[r] = h[r];
    }
    {
      uint2 hi, lo;
      split4(hv4, 16384.f, hi, lo);
      unsigned char* d = hs + (cur ^ 1) * LB * HLS + my_lds;
      *(uint2*)d = hi;
      *(uint2*)(d + 16) = lo;
    }
    if (active)   // write-through store: the output row AND the hand-off to the other three workgroups
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, hv4), rec_rsrc,
                                             (unsigned)((rowi * 512 + dir * 256 + myu) * 4), 0, 16);
    REC_STAMP(2);
    // the other parts' h of this step: the protocol described above (the fill pattern is the arrival flag)
    {
      i32x4 v[3];
      bool act[3], pend[3];
      unsigned ro[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int Tl = xl_T[i], ln = xl_line[i];
        act[i] = s < Tl;
        const int tl = dir ? (Tl - 1 - s) : s;
        ro[i] = (unsigned)((((size_t)(b0 + ln) * T + tl) * 512 + dir * 256 + xl_unit[i]) * 4);
        pend[i] = act[i];
      }
      unsigned spins = 0;
      bool fail = false;
      while (pend[0] || pend[1] || pend[2]) {
#pragma unroll
        for (int i = 0; i < 3; i++)
          if (pend[i]) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, ro[i], 0, 16);
#pragma unroll
        for (int i = 0; i < 3; i++)
          if (pend[i]) pend[i] = v[i][0] == -1 || v[i][1] == -1 || v[i][2] == -1 || v[i][3] == -1;
        if (++spins > SPLIT_MAX_SPINS) { fail = true; break; }
        if ((spins & 255u) == 0 &&
            __hip_atomic_load((gu32_t*)tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fail = true; break; }
      }
      if (fail) {
        __hip_atomic_store((gu32_t*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_fail = 1;
      }
      REC_COUNT(4, spins);
      REC_STAMP(1);
#pragma unroll
      for (int i = 0; i < 3; i++) {
        unsigned char* dst = hs + (cur ^ 1) * LB * HLS + xl_lds[i];
        const unsigned char* old = hs + cur * LB * HLS + xl_lds[i];
        uint2 hi, lo;
        if (act[i]) split4(__builtin_bit_cast(f32x4, v[i]), 16384.f, hi, lo);
        else { hi = *(const uint2*)old; lo = *(const uint2*)(old + 16); }
        *(uint2*)dst = hi;
        *(uint2*)(dst + 16) = lo;
      }
    }
    REC_STAMP(3);
    __syncthreads();
    if (s_fail) return;     // uniform: everybody leaves; the host repeats the batch with the one-workgroup kernel
    __syncthreads();
    REC_STAMP(5);
    cur ^= 1;
  }
  REC_PROF_OUT();
  // padding rows (t >= T_i) of this workgroup's lines, direction and 64 units: zero instead of the fill pattern
  for (int ln = 0; ln < LB; ln++) {
    const int bb = b0 + ln;
    if (bb >= B) break;
    const int Tl = tlen ? tlen[bb] : T;
    for (int r = Tl * 16 + tid; r < T * 16; r += 256)
      *(f32x4*)(rec + ((size_t)bb * T + (r >> 4)) * 512 + dir * 256 + part * 64 + (r & 15) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// ---------------------------------------------------------------------------
// Two-tile form of the hidden-split recurrence (round 3, last session): what a recurrence costs in the PIPELINE is what it
// keeps the other batch's kernels from using -- workgroups x time, every workgroup a whole CU (its waves fill the register
// files) -- and a step of the kernel above is latency (hand-off, loads, barriers), not work.  Here a group of four workgroups
// owns 32 lines as two row tiles of 16 and alternates between them: while the h of tile A's step travels to the other
// three parts, the workgroup computes tile B's step, and the other way round.  A half-step: request the tile's input
// projections; the tile's 96 MFMAs per wave, the polling loads for the OTHER tile's chunks (stored by the other parts in
// the half-step before) issued PK k-steps in; check them (they have had the rest of the MFMAs to arrive; the poll loop of
// the kernel above only runs if one has not), split them into the other tile's next h image; gates, own units into this
// tile's next image and write-through to the output row; ONE barrier (the give-up flag alternates with the half-steps).
// Half as many workgroups as the kernel above, and a step of 32 lines takes about as long as its step of 16.
// Same weights, same MFMA sequence per accumulator, same gate math: bit-identical to both other forms
// (tests/test_hip_scale.py).  Same protocol (fill pattern = arrival flag, bounded polls, *tmo).
// ---------------------------------------------------------------------------
template <int PK>
__global__ __launch_bounds__(256) void lstm_rec_split2_h3_kernel(const float* __restrict__ xp,
                                                                 const uint4* __restrict__ whh3, float* rec,
                                                                 const int32_t* __restrict__ tlen, int B, int T,
                                                                 float unscale, unsigned* tmo) {
  CRNN_WHOLE_REGISTER_FILE();
  constexpr int LBN = 2 * LB;
  __shared__ __attribute__((aligned(16))) unsigned char hs[2 * LBN * HLS];   // [image][tile][line]
  __shared__ int s_fail[2];
  const int dir = blockIdx.y, grp = blockIdx.x >> 2, part = blockIdx.x & 3, b0 = grp * LBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, bl = lane & 15;
  int Ti[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int b = b0 + j * LB + bl;
    Ti[j] = (b < B) ? (tlen ? tlen[b] : T) : 0;
  }
  int Tmax = 0;
  for (int j = 0; j < LBN; j++) {
    const int bb = b0 + j;
    const int tj = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    Tmax = tj > Tmax ? tj : Tmax;
  }
  for (int i = tid; i < 2 * LBN * HLS / 4; i += 256) ((unsigned*)hs)[i] = 0u;
  if (tid < 2) s_fail[tid] = 0;
  // exchange loads of ONE tile: 3 other parts x 16 lines x 16 chunks of 4 units = 768 float4, 3 per thread
  int xl_line[3], xl_unit[3], xl_lds[3], xl_T[2][3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = i * 256 + tid;            // 0..767
    const int op = e >> 8;                  // which of the 3 other parts
    const int p2 = op + (op >= part ? 1 : 0);
    xl_line[i] = (e >> 4) & 15;
    xl_unit[i] = p2 * 64 + (e & 15) * 4;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int bb = b0 + j * LB + xl_line[i];
      xl_T[j][i] = (bb < B) ? (tlen ? tlen[bb] : T) : 0;
    }
    const int k = xl_unit[i];               // units k .. k + 3: block [k / 8 mod 4][k / 32], halves (k & 7) .. + 3
    xl_lds[i] = xl_line[i] * HLS + ((k >> 3) & 3) * 256 + (k >> 5) * 32 + (k & 7) * 2;
  }
  const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, 0x7FFFFFFF, 0x00020000);
  float c[2][4], h[2][4];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) { c[j][r] = 0.f; h[j][r] = 0.f; }
  uint4 w[4][8][2];   // the wave's weights, resident (lstm_rec_split_h3_kernel)
  {
    const uint4* wp = whh3 + ((size_t)(dir * 4 + part) * 8 * 16 * 2) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        w[g][ks][0] = wp[(size_t)((ks * 16 + g * 4 + wave) * 2) * 64];
        w[g][ks][1] = wp[(size_t)((ks * 16 + g * 4 + wave) * 2 + 1) * 64];
      }
  }
  __syncthreads();
  int cur[2] = {0, 0};   // the complete h image of each tile
  REC_PROF_DECL;
  const int myu = part * 64 + wave * 16 + q * 4;
  const int my_lds = bl * HLS + ((myu >> 3) & 3) * 256 + (myu >> 5) * 32 + (myu & 7) * 2;
  // The input projections of a half-step are requested during the half-step before, right after that one's wait for its
  // polling loads (a wait for loads is a wait for ALL loads issued before it, so anything requested earlier would be
  // waited for there): they have the rest of that half-step and this one's MFMAs to arrive.  Unconditional loads (a line
  // past its end reads its row 0 and ignores it).
  f32x4 xgs[2][4];
  auto request_xg = [&](int tile, int step) __attribute__((always_inline)) {
    const int b = b0 + tile * LB + bl;
    const int t = step < Ti[tile] ? (dir ? (Ti[tile] - 1 - step) : step) : 0;
    const float* xrow = xp + ((size_t)(b < B ? b : 0) * T + t) * 2048 + dir * 1024 + part * 64 + wave * 16 + q * 4;
#pragma unroll
    for (int g = 0; g < 4; g++) xgs[tile][g] = *(const f32x4*)(xrow + g * 256);
    asm volatile("" ::: "memory");
  };
  request_xg(0, 0);
  for (int s = 0; s < Tmax; s++) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int o = 1 - j;
      REC_STAMP0();
      // ---- this tile's step
      const int b = b0 + j * LB + bl;
      const bool active = s < Ti[j];
      const int t = active ? (dir ? (Ti[j] - 1 - s) : s) : 0;
      const size_t rowi = (size_t)(b < B ? b : 0) * T + t;
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      // ---- the other tile's hand-off of the half-step before: its step so (tile 1's previous step when this is tile 0);
      // needed only if that tile has another step to compute
      const int so = j == 0 ? s - 1 : s;
      const bool xch = so >= 0 && so + 1 < Tmax;       // (uniform)
      i32x4 v[3];
      bool act[3], pend[3];
      unsigned ro[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int Tl = xl_T[o][i], ln = xl_line[i];
        act[i] = xch && so < Tl;
        const int tl = dir ? (Tl - 1 - so) : so;
        const int bb = b0 + o * LB + ln;
        // (a chunk nobody needs -- a line past its end, a padding line of the group -- is loaded all the same, from a valid
        // row: a load under a condition would be waited for at the end of its branch, one round trip after the other)
        ro[i] = (unsigned)((((size_t)(bb < B ? bb : 0) * T + (act[i] ? tl : 0)) * 512 + dir * 256 + xl_unit[i]) * 4);
        pend[i] = act[i];
      }
      const unsigned char* hrow = hs + (cur[j] * LBN + j * LB) * HLS + bl * HLS + q * 256;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks == PK) {
#pragma unroll
          for (int i = 0; i < 3; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, ro[i], 0, 16);
          asm volatile("" ::: "memory");   // (a pure read to the compiler, which would sink it to its first use)
        }
        const f16x8r hh = *(const f16x8r*)(hrow + ks * 32), hl = *(const f16x8r*)(hrow + ks * 32 + 16);
#pragma unroll
        for (int g = 0; g < 4; g++)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][1]), hh, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][0]), hl, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, w[g][ks][0]), hh, acc[g], 0, 0, 0);
      }
      if (PK >= 8) {
#pragma unroll
        for (int i = 0; i < 3; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, ro[i], 0, 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      REC_STAMP(0);
      // ---- the other tile's chunks: arrived?  (the fill pattern is the arrival flag; lstm_rec_split_h3_kernel's loop otherwise)
      if (xch) {
        bool any = false;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          if (pend[i]) pend[i] = v[i][0] == -1 || v[i][1] == -1 || v[i][2] == -1 || v[i][3] == -1;
          any = any || pend[i];
        }
        unsigned spins = 0;
        bool fail = false;
        while (any) {
#pragma unroll
          for (int i = 0; i < 3; i++)
            if (pend[i]) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, ro[i], 0, 16);
          any = false;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            if (pend[i]) pend[i] = v[i][0] == -1 || v[i][1] == -1 || v[i][2] == -1 || v[i][3] == -1;
            any = any || pend[i];
          }
          if (++spins > SPLIT_MAX_SPINS) { fail = true; break; }
          if ((spins & 255u) == 0 &&
              __hip_atomic_load((gu32_t*)tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fail = true; break; }
        }
        if (fail) {
          __hip_atomic_store((gu32_t*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_fail[j] = 1;
        }
        REC_COUNT(4, spins);
        REC_STAMP(1);
      }
      // ---- the next half-step's input projections (tile o: step s when this is tile 0, else s + 1; past the end: any row)
      request_xg(o, j == 0 ? s : (s + 1 < Tmax ? s + 1 : s));
      if (xch) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          unsigned char* dst = hs + ((cur[o] ^ 1) * LBN + o * LB) * HLS + xl_lds[i];
          const unsigned char* old = hs + (cur[o] * LBN + o * LB) * HLS + xl_lds[i];
          uint2 hi, lo;
          if (act[i]) split4(__builtin_bit_cast(f32x4, v[i]), 16384.f, hi, lo);
          else { hi = *(const uint2*)old; lo = *(const uint2*)(old + 16); }
          *(uint2*)dst = hi;
          *(uint2*)(dst + 16) = lo;
        }
      }
      REC_STAMP(2);
      // ---- this tile's gates
      f32x4 hv4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (active) {
          const float ig = sigmoidf_(acc[0][r] * unscale + xgs[j][0][r]);
          const float fg = sigmoidf_(acc[1][r] * unscale + xgs[j][1][r]);
          const float gg = tanhf_(acc[2][r] * unscale + xgs[j][2][r]);
          const float og = sigmoidf_(acc[3][r] * unscale + xgs[j][3][r]);
          c[j][r] = fg * c[j][r] + ig * gg;
          h[j][r] = og * tanhf_(c[j][r]);
        }
        hv4[r] = h[j][r];
      }
      {
        uint2 hi, lo;
        split4(hv4, 16384.f, hi, lo);
        unsigned char* d = hs + ((cur[j] ^ 1) * LBN + j * LB) * HLS + my_lds;
        *(uint2*)d = hi;
        *(uint2*)(d + 16) = lo;
      }
      if (active)   // write-through store: the output row AND the hand-off to the other three workgroups
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, hv4), rec_rsrc,
                                               (unsigned)((rowi * 512 + dir * 256 + myu) * 4), 0, 16);
      REC_STAMP(3);
      __syncthreads();
      REC_STAMP(5);
      if (s_fail[j]) return;     // uniform (s_fail[j] is next written two barriers from here)
      if (xch) cur[o] ^= 1;      // the other tile's next image is complete: its own units (the half-step before) + the chunks
    }
    // a tile whose hand-off nobody fetched because it has no further step keeps cur; nothing reads it again
  }
  REC_PROF_OUT();
  // padding rows (t >= T_i) of this workgroup's lines, direction and 64 units: zero instead of the fill pattern
  for (int ln = 0; ln < LBN; ln++) {
    const int bb = b0 + ln;
    if (bb >= B) break;
    const int Tl = tlen ? tlen[bb] : T;
    for (int r = Tl * 16 + tid; r < T * 16; r += 256)
      *(f32x4*)(rec + ((size_t)bb * T + (r >> 4)) * 512 + dir * 256 + part * 64 + (r & 15) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// nt: 1 = lstm_rec_split_h3_kernel (16 lines per group of four workgroups), 2 = lstm_rec_split2_h3_kernel (32 lines)
void launch_lstm_rec_split_h3(const float* xp, const void* whh3, float* rec, const int32_t* tlen, int B, int T,
                              float unscale, unsigned* tmo, hipStream_t s, int nt) {
  // tmo: one device word, zeroed by the caller, set to 1 by the kernel when a hand-off timed out.  The output rows double
  // as arrival flags: fill them with a bit pattern no h value has (every row a step can poll must carry it, and with
  // ragged lengths those rows are scattered, so the whole tensor is filled: 37 MB at 256 lines, ~10 us)
  (void)hipMemsetAsync(rec, 0xFF, (size_t)B * T * 512 * sizeof(float), s);
  if (nt == 2) {
    const int groups = (B + 2 * LB - 1) / (2 * LB);
    static const int pk = getenv("CRNN_SPLIT2_PK") ? atoi(getenv("CRNN_SPLIT2_PK")) : 4;
    if (pk <= 0)
      hipLaunchKernelGGL(lstm_rec_split2_h3_kernel<0>, dim3(groups * 4, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen, B, T, unscale, tmo);
    else if (pk < 8)
      hipLaunchKernelGGL(lstm_rec_split2_h3_kernel<4>, dim3(groups * 4, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen, B, T, unscale, tmo);
    else
      hipLaunchKernelGGL(lstm_rec_split2_h3_kernel<8>, dim3(groups * 4, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen, B, T, unscale, tmo);
    return;
  }
  const int groups = (B + LB - 1) / LB;
  hipLaunchKernelGGL(lstm_rec_split_h3_kernel, dim3(groups * 4, 2), dim3(256), 0, s, xp, (const uint4*)whh3, rec, tlen, B,
                     T, unscale, tmo);
}

void launch_lstm_rec(const float* xp, const float* whh_packed, float* rec, const int32_t* tlen, int B,
                     int T, hipStream_t s) {
  hipLaunchKernelGGL(lstm_rec_kernel, dim3((B + LB - 1) / LB, 2), dim3(256), 0, s, xp, whh_packed, rec,
                     tlen, B, T);
}

void pack_whh(const float* whh_fwd, const float* whh_rev, float* packed) {
  for (int dir = 0; dir < 2; dir++) {
    const float* w = dir ? whh_rev : whh_fwd;
    for (int wave = 0; wave < 4; wave++)
      for (int tile = 0; tile < 16; tile++)
        for (int kg = 0; kg < 16; kg++)
          for (int lane = 0; lane < 64; lane++) {
            const int g = tile >> 2, u = tile & 3, q = lane >> 4, bl = lane & 15;
            const int row = g * 256 + wave * 64 + u * 16 + bl;
            const int k = kg * 16 + 4 * q;
            float* dst = packed + ((((size_t)(dir * 4 + wave) * 16 + tile) * 16 + kg) * 64 + lane) * 4;
            for (int j = 0; j < 4; j++) dst[j] = w[(size_t)row * 256 + k + j];
          }
  }
}

__global__ __launch_bounds__(256) void argmax_partials_kernel(const float* __restrict__ pmax,
                                                              const int32_t* __restrict__ pidx,
                                                              int32_t* __restrict__ idx, int M, int nt) {
  // one wave per row: lanes stride over the row's partials (coalesced), then a butterfly; ties -> lowest column
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* pv = pmax + (long)row * nt;
  const int32_t* pi = pidx + (long)row * nt;
  float bv = -INFINITY;
  int bi = 0x7FFFFFFF;
  for (int t = lane; t < nt; t += 64) {
    const float v = pv[t];
    const int i = pi[t];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) idx[row] = bi == 0x7FFFFFFF ? 0 : bi;
}

void launch_argmax_partials(const float* pmax, const int32_t* pidx, int32_t* idx, int M, int ntiles,
                            hipStream_t s) {
  hipLaunchKernelGGL(argmax_partials_kernel, dim3((M + 3) / 4), dim3(256), 0, s, pmax, pidx, idx, M, ntiles);
}

// ---------------------------------------------------------------------------
// Certified bf16 pre-filter for the classifier's arg-max (fp32 path, logits not requested).
// The arg-max over 5530 classes needs every logit only to know which one is largest, and the bf16 MFMA is 16x
// faster than the f32 one.  So: (1) the activations are rounded to bf16 (RNE) and their row norms ||a|| taken;
// (2) the bf16 GEMM writes approximate logits l~ (fp32 accumulate, exact bf16 x bf16 products; stored as bf16, whose
// rounding is added to the bound); (3) per row, with
// the rigorous bound |l~_c - l_c| <= e_c = (2u + u^2) ||a|| ||w_c||  (u = 2^-8, Cauchy-Schwarz over the per-term
// rounding errors, plus slack for the fp32 accumulation), every class whose upper bound l~_c + e_c reaches the best
// lower bound max_c (l~_c - e_c) is a candidate -- the true arg-max is certainly among them -- and (4) the
// candidates (usually one, then nothing is left to do) are scored exactly: fp32 dot products of the fp32
// activations and weights, ties to the lowest index.  The decision is therefore made in fp32; bf16 only prunes.
// ---------------------------------------------------------------------------
// fp32 rows [M][K] -> the f16x3 format with one scale per block of 32 rows (rs[block] = 1 / scale): the rows a wave of
// the GEMM holds per accumulator, so its epilogue needs two scalars per unit.  (One per 128 rows until round 3: 142
// workgroups for the 18,176 rows of a 256-line step left 114 CUs idle, 2.6-3.6 TB/s; 568 workgroups now.)  One
// 256-thread workgroup per block; a thread owns 8 consecutive channels of 32 * K / 2048 rows (all held in registers
// between the maximum and the conversion).  The grid covers whole 128-row GEMM units: blocks past M write their scale.
template <int K>
__global__ __launch_bounds__(256) void rows_to_f16x3_kernel(const float* __restrict__ A, float* __restrict__ out,
                                                            float* __restrict__ rs, int M) {
  constexpr int TPR = K / 8, RPI = 256 / TPR, NIT = F16X3_ROWS / RPI;
  __shared__ float wmax[4];
  const int r0 = blockIdx.x * F16X3_ROWS + threadIdx.x / TPR, g8 = threadIdx.x % TPR;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 v[NIT][2];
  float mx = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int row = r0 + it * RPI;
    const float* a = A + (long)(row < M ? row : 0) * K + g8 * 8;
    v[it][0] = row < M ? *(const f32x4*)a : z;
    v[it][1] = row < M ? *(const f32x4*)(a + 4) : z;
#pragma unroll
    for (int e = 0; e < 4; e++) mx = fmaxf(mx, fmaxf(__builtin_fabsf(v[it][0][e]), __builtin_fabsf(v[it][1][e])));
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; w++) mx = fmaxf(mx, wmax[w]);
  float sc, uns;
  split_scale_from_max(mx, sc, uns);
  if (threadIdx.x == 0) rs[blockIdx.x] = uns;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int row = r0 + it * RPI;
    if (row >= M) continue;
    uint2 h0, l0, h1, l1;
    split4(v[it][0], sc, h0, l0);
    split4(v[it][1], sc, h1, l1);
    uint4* o = (uint4*)(out + (long)row * K + g8 * 8);
    o[0] = uint4{h0.x, h0.y, h1.x, h1.y};
    o[1] = uint4{l0.x, l0.y, l1.x, l1.y};
  }
}
bool launch_rows_to_f16x3(const float* A, float* out, float* rs, int M, int K, hipStream_t s) {
  const dim3 g((M + 127) / 128 * (128 / F16X3_ROWS));
  if (K == 512) hipLaunchKernelGGL((rows_to_f16x3_kernel<512>), g, dim3(256), 0, s, A, out, rs, M);
  else if (K == 256) hipLaunchKernelGGL((rows_to_f16x3_kernel<256>), g, dim3(256), 0, s, A, out, rs, M);
  else return false;
  return true;
}

__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ A, uint16_t* __restrict__ out,
                                                           float* __restrict__ norm, int M, int K, int f16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* a = A + (long)row * K;
  float ss = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 v = *(const f32x4*)(a + k);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    union { __bf16 h[4]; _Float16 f[4]; uint2 u; } pk;
    if (f16) {
#pragma unroll
      for (int e = 0; e < 4; e++) pk.f[e] = (_Float16)v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) pk.h[e] = (__bf16)v[e];
    }
    *(uint2*)(out + (long)row * K + k) = pk.u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) norm[row] = sqrtf(ss) * 1.000001f;
}

void launch_rows_to_bf16(const float* A, uint16_t* out, float* norm, int M, int K, hipStream_t s, int f16) {
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((M + 3) / 4), dim3(256), 0, s, A, out, norm, M, K, f16);
}
uint16_t f32_to_f16_host(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}


// ---------------------------------------------------------------------------
// greedy CTC collapse, one wave per line: lane i looks at timestep base + i and its predecessor, the kept labels are
// compacted with a ballot (position = kept labels before this lane).  (One thread per line walking its T steps took
// 0.028 ms for 256 lines: 256 threads on the whole device.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ tlen,
                                                  int32_t* __restrict__ out, int32_t* __restrict__ len, int B, int T) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const int Ti = tlen ? tlen[b] : T;
  const int32_t* p = idx + (long)b * T;
  int32_t* o = out + (long)b * T;
  int n = 0;
  for (int base = 0; base < Ti; base += 64) {
    const int t = base + lane;
    const int k = t < Ti ? p[t] : 0;
    const int kp = (t > 0 && t < Ti) ? p[t - 1] : -1;
    const bool keep = t < Ti && k != 0 && k != kp;
    const unsigned long long m = __ballot(keep);
    if (keep) o[n + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = k - 1;
    n += __builtin_popcountll(m);
  }
  if (lane == 0) len[b] = n;
}

void launch_ctc(const int32_t* idx, const int32_t* tlen, int32_t* out, int32_t* len, int B, int T,
                hipStream_t s) {
  hipLaunchKernelGGL(ctc_kernel, dim3((B + 3) / 4), dim3(256), 0, s, idx, tlen, out, len, B, T);
}

// ---------------------------------------------------------------------------
// preprocessing
// ---------------------------------------------------------------------------
__constant__ float c_norm_lut[256];

__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ gray,
                                                        float* __restrict__ out,
                                                        const int32_t* __restrict__ w, int B, int W,
                                                        int Wp) {
  const long total = (long)B * 32 * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long row = i / Wp;  // b*32 + y
    const int b = (int)(row >> 5);
    const int wb = w ? w[b] : W;
    out[i] = (x < wb && x < W) ? c_norm_lut[gray[row * W + x]] : 0.f;
  }
}

void launch_normalize(const uint8_t* gray, float* out, const int32_t* w, int B, int W, int Wp,
                      hipStream_t s) {
  hipLaunchKernelGGL(normalize_kernel, dim3(ew_blocks((long)B * 32 * Wp)), dim3(256), 0, s, gray, out, w,
                     B, W, Wp);
}

constexpr int PREC = 22;

__device__ __forceinline__ int src_gray(const uint8_t* row, int x, int ch) {
  if (ch == 1) return row[x];
  const int b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
  return (19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread makes 4 consecutive bytes of the job's intermediate image (rows are contiguous, so the
// flattened pixel index runs over rows; no idle lanes at row ends, one 4-byte store per thread).
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src,
                                                       const ResizeJob* __restrict__ jobs,
                                                       const int32_t* __restrict__ coef,
                                                       uint8_t* __restrict__ tmp) {
  const ResizeJob j = jobs[blockIdx.z];
  const int npix = j.th * j.dw;
  const int p0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (p0 >= npix) return;
  const int32_t* bh = coef + j.coef_off;
  const int32_t* kh = bh + 2 * j.dw;
  unsigned packed = 0;
  int yy = p0 / j.dw, xx = p0 - yy * j.dw;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (p0 + e < npix) {
      const int xmin = bh[2 * xx], xmax = bh[2 * xx + 1];
      const int32_t* k = kh + (long)xx * j.ksh;
      const uint8_t* row = src + j.src_off + (long)(yy + j.ybox_first) * j.stride;
      int ss = 1 << (PREC - 1);
      for (int x = 0; x < xmax; x++) ss += src_gray(row, x + xmin, j.ch) * k[x];
      packed |= (unsigned)clip8(ss) << (8 * e);
    }
    xx++;
    if (xx == j.dw) { xx = 0; yy++; }
  }
  uint8_t* o = tmp + j.tmp_off + p0;        // tmp_off is 16-byte aligned, p0 a multiple of 4
  if (p0 + 3 < npix) *(unsigned*)o = packed;
  else for (int e = 0; p0 + e < npix; e++) o[e] = (uint8_t)(packed >> (8 * e));
}

// Row-staged form: a block owns RH_ROWS source rows of one job; it converts each source pixel to gray ONCE
// into LDS (coalesced reads of the source row; the per-output-pixel form above converts every source
// pixel 2-3 times, once per overlapping window) and then resamples from LDS.  Sources wider than
// RH_MAXW fall back to the kernel above.  Same integer arithmetic, bit for bit.
constexpr int RH_ROWS = 4, RH_MAXW = 4096;

__global__ __launch_bounds__(256) void resize_h_rows_kernel(const uint8_t* __restrict__ src,
                                                            const ResizeJob* __restrict__ jobs,
                                                            const int32_t* __restrict__ coef,
                                                            uint8_t* __restrict__ tmp) {
  __shared__ uint8_t gray[RH_ROWS][RH_MAXW];
  const ResizeJob j = jobs[blockIdx.z];
  const int r0 = blockIdx.x * RH_ROWS;
  if (r0 >= j.th) return;
  const int nr = min(RH_ROWS, j.th - r0);
  for (int r = 0; r < nr; r++) {
    const uint8_t* row = src + j.src_off + (long)(r0 + r + j.ybox_first) * j.stride;
    for (int x = threadIdx.x; x < j.sw; x += 256) gray[r][x] = (uint8_t)src_gray(row, x, j.ch);
  }
  __syncthreads();
  const int32_t* bh = coef + j.coef_off;
  const int32_t* kh = bh + 2 * j.dw;
  const int npix = nr * j.dw;
  uint8_t* o = tmp + j.tmp_off + (long)r0 * j.dw;
  for (int p = threadIdx.x; p < npix; p += 256) {
    const int r = p / j.dw, xx = p - r * j.dw;
    const int xmin = bh[2 * xx], xmax = bh[2 * xx + 1];
    const int32_t* k = kh + (long)xx * j.ksh;
    int ss = 1 << (PREC - 1);
    for (int x = 0; x < xmax; x++) ss += (int)gray[r][x + xmin] * k[x];
    o[p] = (uint8_t)clip8(ss);
  }
}

__global__ __launch_bounds__(256) void resize_v_kernel(const ResizeJob* __restrict__ jobs,
                                                       const int32_t* __restrict__ coef,
                                                       const uint8_t* __restrict__ tmp,
                                                       float* __restrict__ out, int Wp) {
  const ResizeJob j = jobs[blockIdx.z];
  const int yy = blockIdx.y;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= Wp) return;
  float v = 0.f;
  if (xx < j.dw) {
    const int32_t* bv = coef + j.coef_off + 2 * j.dw + (long)j.dw * j.ksh;
    const int32_t* kv = bv + 64;
    const int ymin = bv[2 * yy], ymax = bv[2 * yy + 1];
    const int32_t* k = kv + yy * j.ksv;
    const uint8_t* col = tmp + j.tmp_off + xx;
    int ss = 1 << (PREC - 1);
    for (int y = 0; y < ymax; y++) ss += (int)col[(long)(y + ymin) * j.dw] * k[y];
    v = c_norm_lut[clip8(ss)];
  }
  out[((long)blockIdx.z * 32 + yy) * Wp + xx] = v;
}

void launch_resize_h(const uint8_t* src, const ResizeJob* jobs, const int32_t* coef, uint8_t* tmp, int n,
                     int max_th, int max_dw, int max_sw, hipStream_t s) {
  if (max_sw <= RH_MAXW) {
    hipLaunchKernelGGL(resize_h_rows_kernel, dim3((max_th + RH_ROWS - 1) / RH_ROWS, 1, n), dim3(256), 0, s, src, jobs,
                       coef, tmp);
    return;
  }
  const long maxpix = (long)max_th * max_dw;
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((maxpix + 1023) / 1024), 1, n), dim3(256), 0, s, src, jobs,
                     coef, tmp);
}
void launch_resize_v(const uint8_t* /*src*/, const ResizeJob* jobs, const int32_t* coef,
                     const uint8_t* tmp, float* out, int n, int Wp, hipStream_t s) {
  hipLaunchKernelGGL(resize_v_kernel, dim3((Wp + 255) / 256, 32, n), dim3(256), 0, s, jobs, coef, tmp,
                     out, Wp);
}

float to_f16x3_host(const float* src, size_t n, uint16_t* dst) {
  float mx = 0.f;
  for (size_t i = 0; i < n; i++) mx = fmaxf(mx, fabsf(src[i]));
  int e = 0;
  if (mx > 0.f && std::isfinite(mx)) (void)frexpf(mx, &e);   // mx = f 2^e, f in [0.5, 1): floor(log2 mx) = e - 1
  const float s = ldexpf(1.f, 14 - (e - 1)), uns = ldexpf(1.f, (e - 1) - 14);
  for (size_t i = 0; i < n; i++) {
    const float x = src[i] * s;
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    const size_t blk = i / 8, k = i % 8;
    memcpy(dst + blk * 16 + k, &hi, 2);
    memcpy(dst + blk * 16 + 8 + k, &lo, 2);
  }
  return uns;
}

void upload_norm_lut(const float* lut256, hipStream_t s) {
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_norm_lut), lut256, 256 * sizeof(float), 0,
                               hipMemcpyHostToDevice, s);
}


// One empty launch from this translation unit: the runtime loads a code object (and sets a stream's queue up) at the
// first launch that needs it; crnn_api.hip does these at device set-up so that no later call is the first one.
__global__ void warm_main_kernel() {}
void launch_warm_main(hipStream_t s) { hipLaunchKernelGGL(warm_main_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace crnn
