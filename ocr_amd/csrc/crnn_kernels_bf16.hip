// gfx950 kernels of the bf16 path: bf16 activations / weights, fp32 accumulation
// (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16), fp32 bias, gates and logits.
// Same structure and the same NHWC / batch-major layouts as the fp32 kernels in
// crnn_kernels.hip; this path reports character-level agreement with fp32, it is not
// held to the 1e-3 logit bound.
#include "crnn_kernels.h"

namespace crnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution / GEMM, bf16 operands.  128x128 tile, BK = 64 bf16
// (the same 128 bytes per row per K-tile as the fp32 kernel), rows padded to
// 144 bytes in LDS, one LDS buffer + register prefetch, raw buffer loads.
// MFMA 32x32x16: lane l holds A[row l&31][k = 8*(l>>5) + 0..7] as one 16-byte
// LDS read; 4 k-steps per K-tile, 2x2 tiles per wave.
// ---------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BKH = 64, LDB = 144;  // LDB: LDS row stride in bytes

typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
// SW (bf16 output, N % 128 == 0, ldc % 8 == 0, no arg-max): the W fragment is the first MFMA operand, so a lane ends with 4
// consecutive columns of one row, and the block's output goes through the (by then free) LDS tile buffer as whole
// 128-byte lines of C instead of 2-byte pieces of 32 different rows.
template <int OUT_BF16, bool SW = false>   // 0: fp32 output, 1: bf16 output, 2: fp16 operands and output (same bytes, v_mfma_f32_32x32x16_f16)
__global__ __launch_bounds__(256, 3) void conv_gemm_bf16_kernel(const ConvGemm p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bm = blockIdx.x / ntn, bn = blockIdx.x - bm * ntn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int lrow = tid >> 3;
  const int lk = (tid & 7) * 8;  // bf16 elements
  const int hw = p.Hout * p.Wout;
  const __bf16* A = (const __bf16*)p.A;
  const __bf16* Wt = (const __bf16*)p.Wt;

  const long shift = ((long)p.pad * p.Win + p.pad) * p.Cin;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(A - shift), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, 0x7FFFFFFF, 0x00020000);
  unsigned aoff[4], amask[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = m0 + lrow + 32 * i;
    m = m < p.M ? m : p.M - 1;
    const int b = m / hw, r = m - b * hw;
    const int y = r / p.Wout, x = r - y * p.Wout;
    aoff[i] = (unsigned)(((((long)b * p.Hin + y) * p.Win + x) * p.Cin + lk) * 2);
    unsigned mk = 0;
    for (int ky = 0; ky < p.ks; ky++)
      for (int kx = 0; kx < p.ks; kx++) {
        const int yy = y - p.pad + ky, xx = x - p.pad + kx;
        if ((unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win) mk |= 1u << (ky * p.ks + kx);
      }
    amask[i] = mk;
    int n = n0 + lrow + 32 * i;
    n = n < p.N ? n : p.N - 1;
    boff[i] = (unsigned)(((long)n * p.K + lk) * 2);
  }
  const int nk = p.K / BKH;
  i32x4 ra[4], rb[4];
  int l_tap = 0, l_ky = 0, l_kx = 0, l_ci = 0, l_k0 = 0;
  auto gload = [&]() {
    const int toff = ((l_ky * p.Win + l_kx) * p.Cin + l_ci) * 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned vo = ((amask[i] >> l_tap) & 1u) ? aoff[i] : 0x80000000u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, vo, toff, 0);
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, boff[i], l_k0 * 2, 0);
    }
    // next K-tile: taps cycle fastest, then the channel chunk (weights are packed in this order)
    l_k0 += BKH;
    l_tap++;
    l_kx++;
    const bool wrap_x = l_kx == p.ks;
    l_kx = wrap_x ? 0 : l_kx;
    l_ky += wrap_x ? 1 : 0;
    const bool wrap_t = l_ky == p.ks;
    l_ky = wrap_t ? 0 : l_ky;
    l_tap = wrap_t ? 0 : l_tap;
    l_ci += wrap_t ? BKH : 0;
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *(i32x4*)(smem + (lrow + 32 * i) * LDB + lk * 2) = ra[i];
      *(i32x4*)(smem + BM * LDB + (lrow + 32 * i) * LDB + lk * 2) = rb[i];
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
    const unsigned char* a = smem + (wm * 64 + (lane & 31)) * LDB + 16 * (lane >> 5);
    const unsigned char* b = smem + BM * LDB + (wn * 64 + (lane & 31)) * LDB + 16 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const bf16x8 a0 = *(const bf16x8*)(a + ks * 32);
      const bf16x8 a1 = *(const bf16x8*)(a + 32 * LDB + ks * 32);
      const bf16x8 b0 = *(const bf16x8*)(b + ks * 32);
      const bf16x8 b1 = *(const bf16x8*)(b + 32 * LDB + ks * 32);
      if (OUT_BF16 == 2) {
        const f16x8g ha0 = __builtin_bit_cast(f16x8g, a0), ha1 = __builtin_bit_cast(f16x8g, a1);
        const f16x8g hb0 = __builtin_bit_cast(f16x8g, b0), hb1 = __builtin_bit_cast(f16x8g, b1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc[1][1], 0, 0, 0);
      } else if (SW) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
      } else {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  };
  gload();
  sstore();
  __syncthreads();
  for (int kt = 0; kt < nk - 1; kt++) {
    gload();
    compute();
    __syncthreads();
    sstore();
    __syncthreads();
  }
  compute();

  if (!OUT_BF16 && p.pmax) {   // fused classifier arg-max (wave-uniform branch)
    __syncthreads();
    argmax_epilogue(acc, p, (float*)smem, (int*)smem + 256, m0, n0, bn, ntn, wm, wn, lane, tid);
    if (!p.C) return;
  }

  if (SW) {
    // acc[i][j][4 g + e] = C[m0 + wm 64 + i 32 + (lane & 31)][n0 + wn 64 + j 32 + 8 g + 4 (lane >> 5) + e]
    __syncthreads();   // every wave is done with the operand tiles: the buffer becomes four 64-row output tiles
    unsigned char* wt = smem + wave * (64 * 144);
    const int act = p.act, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = m0 + wm * 64 + i * 32 + (lane & 31);
      bool keep = true;
      if (p.wvalid) {
        const int mm = m < p.M ? m : p.M - 1;
        const int b = mm / hw, x = (mm - b * hw) % p.Wout;
        keep = x < p.wvalid[b];
      }
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int nl = j * 32 + 8 * g + 4 * h;
          const f32x4 bs = *(const f32x4*)(p.bias + n0 + wn * 64 + nl);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float v = acc[i][j][4 * g + e] + bs[e];
            const float neg = act == 2 ? 0.2f * v : (act == 1 ? 0.f : v);
            v = v > 0.f ? v : neg;
            o[e] = (__bf16)(keep ? v : 0.f);
          }
          *(bf16x4*)(wt + (i * 32 + (lane & 31)) * 144 + nl * 2) = o;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 8 + (lane >> 3);
      const i32x4 v = *(const i32x4*)(wt + row * 144 + (lane & 7) * 16);
      const int m = m0 + wm * 64 + row;
      if (m < p.M) *(i32x4*)((__bf16*)p.C + (long)m * p.ldc + n0 + wn * 64 + (lane & 7) * 8) = v;
    }
    return;
  }
  store_epilogue<OUT_BF16>(acc, p, p.C, m0, n0, wm, wn, lane);
}

void launch_conv_gemm_bf16(const ConvGemm& p, int out_mode, hipStream_t s) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  if (out_mode == 2) hipLaunchKernelGGL(conv_gemm_bf16_kernel<2>, dim3(ntm * ntn), dim3(256), 0, s, p);
  else if (out_mode && !p.pmax && p.N % 128 == 0 && p.ldc % 8 == 0 && p.nbatch <= 1)
    hipLaunchKernelGGL((conv_gemm_bf16_kernel<1, true>), dim3(ntm * ntn), dim3(256), 0, s, p);   // line-wide stores through LDS
  else if (out_mode) hipLaunchKernelGGL(conv_gemm_bf16_kernel<1>, dim3(ntm * ntn), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv_gemm_bf16_kernel<0>, dim3(ntm * ntn), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------
// Plain GEMM C = A W^T + bias with 2-byte operands (fp16 or bf16) on the direct-to-LDS pipeline: the classifier's
// pre-filter (18176 x 5530 x 512 per 256-line step) and the linear layers of the bf16 path.  Such a GEMM is bound by
// the rate at which a CU can pull operands through L1 (about 45 GB/s per CU, 11 TB/s over the chip, measured on the
// Winograd GEMMs), so the unit is as large as the registers allow: 256 x 256 per block, 8 waves as 2 (M) x 4 (N),
// wave tile 128 x 64 = 8 accumulators of v_mfma_f32_32x32x16 (128 registers).  K in steps of 64 (128 bytes per row,
// the XOR-swizzled row-pair layout of wino4_gemm_lw_kernel), two 64 KB stages; every wave issues 8 of the 64 loads of a
// step.  The barrier of a step sits after its last fragment read and before its last 8 MFMAs: behind it the stage is
// refilled with step + 2, ahead of it every wave has waited for its loads of step + 1.  Persistent, units with the
// column block fastest (the 22 column blocks of the classifier share their A rows in one XCD's L2).
// The W fragment is the first MFMA operand, so a lane ends with 4 consecutive columns of one row: 8-byte (2-byte
// output) or 16-byte (fp32 output) stores.  The bias of a unit comes in by the same direct-to-LDS path (a global
// load in the epilogue would make the compiler drain every load in flight).
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds16_ptr_t;
#ifndef G16_ABL
#define G16_ABL 0   // tools/gemm16_bench.hip: 1 no MFMAs, 2 no loads, 4 no stores
#endif
template <int F16, int OUT>   // F16: fp16 operands (else bf16); OUT 0: fp32, 1: bf16, 2: fp16
__global__ __launch_bounds__(512, 1) void gemm16_dma_kernel(const ConvGemm p) {
  constexpr int TB = 256, SROW = TB * 32, STG = 2 * SROW;   // floats: one operand's rows of a stage, a stage
  __shared__ __attribute__((aligned(1024))) float smem[2 * STG + 2 * TB + 8 * 16 * 36];   // 128 KB + the bias of two units + 8 store tiles (18 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = (p.N + TB - 1) / TB, ntm = (p.M + TB - 1) / TB;
  const int NU = ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.K >> 6;
  const unsigned lds0 = (unsigned)(size_t)(lds16_ptr_t)smem;
  auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  auto dma1 = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff) __attribute__((always_inline)) {
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc) : "memory");
  };
  const unsigned long long pa = (unsigned long long)p.A, pb = (unsigned long long)p.Wt, pbias = (unsigned long long)p.bias;
  const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
  const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
  const i32x4 s_rsrc = {(int)(unsigned)pbias, (int)((pbias >> 32) & 0xFFFF), (int)(p.N * 4), 0x00020000};   // past N: zeros
  // loads: instruction i of this wave fills LDS bytes [(wave * 4 + i) * 1024, +1024) of an operand's stage rows: 16-byte
  // position P = (wave * 4 + i) * 64 + lane -> row pair S = P >> 4, chunk slot c16 = (P & 15) ^ (S & 15)
  int l_u = L, l_ks = 0, l_st = 0, l_par = 0;
  unsigned aoff[4], boff[4];
  auto unit_offsets = [&]() __attribute__((always_inline)) {
    const int mt = l_u / ntn, nt = l_u - mt * ntn;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int P = (wave * 4 + i) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      const int r = 2 * S + (c16 >> 3);
      int m = mt * TB + r, n = nt * TB + r;
      m = m < p.M ? m : p.M - 1;
      n = n < p.N ? n : p.N - 1;
      aoff[i] = (unsigned)(((long)m * p.K + (c16 & 7) * 8) * 2);
      boff[i] = (unsigned)(((long)n * p.K + (c16 & 7) * 8) * 2);
    }
  };
  unit_offsets();
  auto gload = [&]() __attribute__((always_inline)) {
    const unsigned sa = lds0 + (unsigned)l_st * (STG * 4) + (unsigned)wave * 4096u, sb = sa + SROW * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      dma(a_rsrc, sa + i * 1024, aoff[i], l_ks * 128);
      dma(b_rsrc, sb + i * 1024, boff[i], l_ks * 128);
    }
    if (l_ks == 0 && wave < 4) {   // the unit's 256 bias values, 64 per wave
      const int nt = l_u % ntn;
      dma1(s_rsrc, lds0 + 2 * STG * 4 + (unsigned)(l_par * TB + wave * 64) * 4u, (unsigned)((nt * TB + wave * 64 + lane) * 4));
    }
    l_st ^= 1;
    if (++l_ks == nk) {
      l_ks = 0;
      l_par ^= 1;
      if (l_u + G < NU) l_u += G;   // past the last unit: repeats it (never read)
      unit_offsets();
    }
  };
  // fragments: chunk c of row R at float offset (R >> 1) * 64 + ((((R & 1) << 3) | c) ^ ((R >> 1) & 15)) * 4; k16-step j
  // of a K = 64 step reads chunk 2 j + (lane >> 5)
  int fa[4], fb[4];
  {
    const int Ra = wm * 128 + (lane & 31), Rb = wn * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = 2 * j + h;
      fa[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | c) ^ ((Ra >> 1) & 15)) << 2);
      fb[j] = SROW + (Rb >> 1) * 64 + (((((Rb & 1) << 3) | c) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 af[2][4], bf[2][2];
  auto frag = [&](int stage, int j, int set) __attribute__((always_inline)) {
    const float* sb = smem + stage * STG;
#pragma unroll
    for (int mt = 0; mt < 4; mt++) af[set][mt] = *(const s16x8*)(sb + fa[j] + mt * 1024);   // + 32 rows: same swizzle
#pragma unroll
    for (int nt = 0; nt < 2; nt++) bf[set][nt] = *(const s16x8*)(sb + fb[j] + nt * 1024);
  };
  auto mfma8 = [&](int set) __attribute__((always_inline)) {
    if (G16_ABL & 1) { asm volatile("" ::"v"(af[set][0]), "v"(af[set][1]), "v"(af[set][2]), "v"(af[set][3]), "v"(bf[set][0]), "v"(bf[set][1])); return; }
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        if (F16)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, bf[set][nt]), __builtin_bit_cast(f16x8g, af[set][mt]), acc[mt][nt], 0, 0, 0);
        else
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[set][nt]), __builtin_bit_cast(bf16x8, af[set][mt]), acc[mt][nt], 0, 0, 0);
      }
  };
  int c_u = L, c_par = 0, kc = 0;
  // Epilogue.  2-byte outputs go through a per-wave LDS tile (16 rows x 64 columns, rows padded to 144 bytes) so that
  // eight lanes write one full 128-byte line of C; straight from the accumulators a lane writes 8 bytes of 32 different
  // rows (measured: the stores then cost 0.065 ms of the classifier filter's 0.19 ms; tools/gemm16_bench.hip).
  unsigned char* wt = (unsigned char*)(smem + 2 * STG + 2 * TB) + wave * (16 * 144);
  auto flush = [&]() __attribute__((always_inline)) {
    const int mt0 = c_u / ntn, nt0 = c_u - mt0 * ntn;
    const float* bl = smem + 2 * STG + c_par * TB + wn * 64 + 4 * (lane >> 5);
    if (OUT != 0 && (p.ldc & 7) == 0) {   // (rows of C 16-byte aligned; columns N .. ldc - 1 may be written: padding)
      const int r = lane & 15, hrow = (lane >> 4) & 1;   // lanes 0-15 / 16-31 hold rows 0-15 / 16-31 of a 32-row tile
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (hrow == half) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
              for (int g = 0; g < 4; g++) {
                const f32x4 bs = *(const f32x4*)(bl + nt * 32 + 8 * g);
                f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                v += bs;
                unsigned char* d = wt + r * 144 + (nt * 32 + 8 * g + 4 * (lane >> 5)) * 2;
                if (OUT == 1) {
                  typedef __bf16 bf16x4b __attribute__((ext_vector_type(4)));
                  *(bf16x4b*)d = bf16x4b{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                } else {
                  typedef _Float16 f16x4b __attribute__((ext_vector_type(4)));
                  *(f16x4b*)d = f16x4b{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                }
              }
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const int rr = i * 8 + (lane >> 3);
            const i32x4 v = *(const i32x4*)(wt + rr * 144 + (lane & 7) * 16);
            const int m = mt0 * TB + wm * 128 + mt * 32 + half * 16 + rr;
            const int n = nt0 * TB + wn * 64 + (lane & 7) * 8;
            if (m < p.M && n < p.N && n + 8 <= p.ldc && !((G16_ABL & 4) && v[0] != 12345)) *(i32x4*)((uint16_t*)p.C + (long)m * p.ldc + n) = v;
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
          for (int r2 = 0; r2 < 16; r2++) acc[mt][nt][r2] = 0.f;
      }
      kc = 0;
      c_u += G;
      c_par ^= 1;
      return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 bs = *(const f32x4*)(bl + nt * 32 + 8 * g);
        const int n = nt0 * TB + wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
          const int m = mt0 * TB + wm * 128 + mt * 32 + (lane & 31);
          f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
          v += bs;
#pragma unroll
          for (int e = 0; e < 4; e++) acc[mt][nt][4 * g + e] = 0.f;
          if (m >= p.M || n >= p.N || ((G16_ABL & 4) && v[0] != 12345.f)) continue;
          const long off = (long)m * p.ldc + n;
          if (n + 3 < p.N) {
            if (OUT == 0) {
              typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));
              *(f32x4a*)(p.C + off) = v;
            } else if (OUT == 1) {
              typedef __bf16 bf16x4a __attribute__((ext_vector_type(4), aligned(2)));
              bf16x4a o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              *(bf16x4a*)((__bf16*)p.C + off) = o;
            } else {
              typedef _Float16 f16x4a __attribute__((ext_vector_type(4), aligned(2)));
              f16x4a o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
              *(f16x4a*)((_Float16*)p.C + off) = o;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (n + e < p.N) {
                if (OUT == 0) p.C[off + e] = v[e];
                else if (OUT == 1) ((__bf16*)p.C)[off + e] = (__bf16)v[e];
                else ((_Float16*)p.C)[off + e] = (_Float16)v[e];
              }
          }
        }
      }
    kc = 0;
    c_u += G;
    c_par ^= 1;
  };
  const int nit = my_units * nk;
  gload();
  gload();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the first step (and its bias) landed; the second may fly
  __syncthreads();
  int st = 0;
  frag(0, 0, 0);
#pragma unroll 1
  for (int it = 0; it < nit; it++) {
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stage st is in registers; my loads of the next step landed
    __syncthreads();
    if (!(G16_ABL & 2)) gload();                                   // step it + 2 into stage st
    st ^= 1;
    frag(st, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool gemm16_dma_ok(const ConvGemm& p) {
  return p.ks == 1 && p.K % 64 == 0 && p.K >= 128 && !p.pmax && !p.wvalid && p.act == 0 && p.nbatch <= 1 && p.C &&
         (double)p.M * p.K * 2 < 2.0e9 && (double)p.N * p.K * 2 < 2.0e9 && (long)p.M * p.N >= 256L * 256 * 64;
}
// f16: 1 = fp16 operands, 0 = bf16; out: 0 fp32, 1 bf16, 2 fp16
void launch_gemm16_dma(const ConvGemm& p, int f16, int out, hipStream_t s) {
  const int nu = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int g = nu < persistent_blocks() ? 8 * ((nu + 7) / 8) : persistent_blocks();
  if (f16 && out == 2) hipLaunchKernelGGL((gemm16_dma_kernel<1, 2>), dim3(g), dim3(512), 0, s, p);
  else if (!f16 && out == 1) hipLaunchKernelGGL((gemm16_dma_kernel<0, 1>), dim3(g), dim3(512), 0, s, p);
  else if (!f16 && out == 0) hipLaunchKernelGGL((gemm16_dma_kernel<0, 0>), dim3(g), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((gemm16_dma_kernel<1, 0>), dim3(g), dim3(512), 0, s, p);
}

// ---------------------------------------------------------------------------
// conv0 + ReLU + pool0 (fp32 arithmetic on the fp32 input), bf16 NHWC output
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_pool_bf16_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ w9x64,
                                                              const float* __restrict__ bias,
                                                              __bf16* __restrict__ out,
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
          for (int kx = 0; kx < 3; kx++) a += wr[ky * 3 + kx] * patch[py + ky][px + kx];
        if (py == 0 && px == 0) best = a;
        else {
#pragma unroll
          for (int c = 0; c < 4; c++) best[c] = a[c] > best[c] ? a[c] : best[c];
        }
      }
    bf16x4 o;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float v = best[c];
      if (act == 1) v = v > 0.f ? v : 0.f;
      else if (act == 2) v = v > 0.f ? v : 0.2f * v;
      if (w1valid && x1 >= w1valid[b]) v = 0.f;
      o[c] = (__bf16)v;
    }
    *(bf16x4*)(out + pix * 64 + cg * 4) = o;
  }
}

void launch_conv0_pool_bf16(const float* in, const float* w9x64, const float* bias, void* out,
                            const int32_t* w1valid, int B, int W, int act, hipStream_t s) {
  long npix = (long)B * 16 * (W >> 1);
  long blocks = (npix + 15) / 16;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(conv0_pool_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, w9x64, bias,
                     (__bf16*)out, w1valid, B, W, act);
}

// ---------------------------------------------------------------------------
// NHWC max pools on bf16 (8 channels per 16-byte access; max is exact)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 max8(bf16x8 a, bf16x8 b) {
  bf16x8 r;
#pragma unroll
  for (int c = 0; c < 8; c++) r[c] = (float)a[c] > (float)b[c] ? a[c] : b[c];
  return r;
}

__global__ __launch_bounds__(256) void pool_2x2s2_bf16_kernel(const __bf16* __restrict__ in,
                                                              __bf16* __restrict__ out, int B, int H,
                                                              int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, C8 = C >> 3;
  const long total = (long)B * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long pix = i / C8;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const __bf16* p = in + (((long)b * H + 2 * y) * W + 2 * x) * C + c8 * 8;
    bf16x8 v = max8(max8(*(const bf16x8*)p, *(const bf16x8*)(p + C)),
                    max8(*(const bf16x8*)(p + (long)W * C), *(const bf16x8*)(p + (long)W * C + C)));
    *(bf16x8*)(out + i * 8) = v;
  }
}

__global__ __launch_bounds__(256) void pool_2x2s21p01_bf16_kernel(const __bf16* __restrict__ in,
                                                                  __bf16* __restrict__ out, int B, int H,
                                                                  int W, int C) {
  const int Ho = H >> 1, Wo = W + 1, C8 = C >> 3;
  const long total = (long)B * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long pix = i / C8;
    const int x = (int)(pix % Wo);
    pix /= Wo;
    const int y = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const __bf16* r0 = in + (((long)b * H + 2 * y) * W) * C + c8 * 8;
    const __bf16* r1 = r0 + (long)W * C;
    bf16x8 v;
    if (x < W) {
      v = max8(*(const bf16x8*)(r0 + (long)x * C), *(const bf16x8*)(r1 + (long)x * C));
      if (x > 0) v = max8(v, max8(*(const bf16x8*)(r0 + (long)(x - 1) * C), *(const bf16x8*)(r1 + (long)(x - 1) * C)));
    } else {
      v = max8(*(const bf16x8*)(r0 + (long)(x - 1) * C), *(const bf16x8*)(r1 + (long)(x - 1) * C));
    }
    *(bf16x8*)(out + i * 8) = v;
  }
}

static unsigned ew_blocks16(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

void launch_pool_2x2s2_bf16(const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
  long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(pool_2x2s2_bf16_kernel, dim3(ew_blocks16(total)), dim3(256), 0, s, (const __bf16*)in,
                     (__bf16*)out, B, H, W, C);
}
void launch_pool_2x2s21p01_bf16(const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
  long total = (long)B * (H / 2) * (W + 1) * (C / 8);
  hipLaunchKernelGGL(pool_2x2s21p01_bf16_kernel, dim3(ew_blocks16(total)), dim3(256), 0, s,
                     (const __bf16*)in, (__bf16*)out, B, H, W, C);
}

// ---------------------------------------------------------------------------
// LSTM recurrence, bf16 operands.  One 256-thread workgroup per (direction, 16
// lines) as in the fp32 kernel; gates^T = W_hh . h^T on v_mfma_f32_16x16x32_bf16
// (lane l: A[row l&15][k = 8*(l>>4)+0..7], B[k = 8*(l>>4)+0..7][col l&15]).
// W_hh (bf16, 512 KiB per direction) streams from L2 in packed fragment order;
// h is kept in LDS as bf16 (the MFMA operand), c and the gates in fp32.
// ---------------------------------------------------------------------------
constexpr int LB = 16, HSB = 264;  // HSB: bf16 elements per h row in LDS (528 bytes)

__device__ __forceinline__ float sigmoid_b(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * x));
}
__device__ __forceinline__ float tanh_b(float x) {
  const float t = __builtin_amdgcn_exp2f(-2.885390082f * __builtin_fabsf(x));
  const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return __builtin_copysignf(r, x);
}

__global__ __launch_bounds__(256) void lstm_rec_bf16_kernel(const float* __restrict__ xp,
                                                            const __bf16* __restrict__ whhp,
                                                            __bf16* __restrict__ rec,
                                                            const int32_t* __restrict__ tlen, int B, int T) {
  CRNN_WHOLE_REGISTER_FILE();
  __shared__ __attribute__((aligned(16))) __bf16 hs[2 * LB * HSB];
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
  for (int i = tid; i < 2 * LB * HSB; i += 256) hs[i] = (__bf16)0.f;
  float c[4][4], h[4][4];
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int r = 0; r < 4; r++) { c[u][r] = 0.f; h[u][r] = 0.f; }
  // packed: [dir][wave][kstep 8][tile 16][lane 64] x 8 bf16; streamed through two register sets of 8 tiles, half a
  // k-step ahead of the MFMAs (as lstm_rec_h3_kernel: with the loads left to the compiler a step took 11 us, as long as
  // the f16x3 form with twice the bytes)
  const bf16x8* wp = (const bf16x8*)whhp + ((size_t)(dir * 4 + wave) * 8 * 16) * 64 + lane;
  bf16x8 wa[8], wb[8];
#pragma unroll
  for (int i = 0; i < 8; i++) wa[i] = wp[(size_t)i * 64];
  __syncthreads();
  int cur = 0;
  for (int s = 0; s < Tmax; s++) {
    const bool active = s < Ti;
    const int t = active ? (dir ? (Ti - 1 - s) : s) : 0;
    const size_t rowi = (size_t)(b < B ? b : 0) * T + t;
    const float* xrow = xp + rowi * 2048 + dir * 1024 + wave * 64 + q * 4;
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
    const __bf16* hrow = hs + cur * LB * HSB + bl * HSB + 8 * q;
#pragma unroll 1
    for (int ks = 0; ks < 8; ks++) {
      const bf16x8 hv = *(const bf16x8*)(hrow + ks * 32);
#pragma unroll
      for (int i = 0; i < 8; i++) wb[i] = wp[(size_t)(ks * 16 + 8 + i) * 64];
#pragma unroll
      for (int i = 0; i < 8; i++)
        acc[i >> 2][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[i], hv, acc[i >> 2][i & 3], 0, 0, 0);
      const int kn = (ks + 1) & 7;   // the last refill wraps to k-step 0 of the NEXT step: it lands under the gate math
#pragma unroll
      for (int i = 0; i < 8; i++) wa[i] = wp[(size_t)(kn * 16 + i) * 64];
#pragma unroll
      for (int i = 0; i < 8; i++)
        acc[2 + (i >> 2)][i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i], hv, acc[2 + (i >> 2)][i & 3], 0, 0, 0);
    }
    __bf16* hn = hs + (cur ^ 1) * LB * HSB + bl * HSB + wave * 64 + q * 4;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      bf16x4 hv4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (active) {
          const float ig = sigmoid_b(acc[0][u][r] + xg[0][u][r]);
          const float fg = sigmoid_b(acc[1][u][r] + xg[1][u][r]);
          const float gg = tanh_b(acc[2][u][r] + xg[2][u][r]);
          const float og = sigmoid_b(acc[3][u][r] + xg[3][u][r]);
          c[u][r] = fg * c[u][r] + ig * gg;
          h[u][r] = og * tanh_b(c[u][r]);
        }
        hv4[r] = (__bf16)h[u][r];
      }
      *(bf16x4*)(hn + u * 16) = hv4;
      if (active) *(bf16x4*)(rec + rowi * 512 + dir * 256 + wave * 64 + u * 16 + q * 4) = hv4;
    }
    __syncthreads();
    cur ^= 1;
  }
}

void launch_lstm_rec_bf16(const float* xp, const void* whh_packed, void* rec, const int32_t* tlen, int B,
                          int T, hipStream_t s) {
  hipLaunchKernelGGL(lstm_rec_bf16_kernel, dim3((B + LB - 1) / LB, 2), dim3(256), 0, s, xp,
                     (const __bf16*)whh_packed, (__bf16*)rec, tlen, B, T);
}

// host: fp32 -> bf16 round-to-nearest-even (finite weights)
uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

void pack_whh_bf16(const float* whh_fwd, const float* whh_rev, uint16_t* packed) {
  for (int dir = 0; dir < 2; dir++) {
    const float* w = dir ? whh_rev : whh_fwd;
    for (int wave = 0; wave < 4; wave++)
      for (int ks = 0; ks < 8; ks++)
        for (int tile = 0; tile < 16; tile++)
          for (int lane = 0; lane < 64; lane++) {
            const int g = tile >> 2, u = tile & 3, q = lane >> 4, bl = lane & 15;
            const int row = g * 256 + wave * 64 + u * 16 + bl;
            const int k = ks * 32 + 8 * q;
            uint16_t* dst = packed + ((((size_t)(dir * 4 + wave) * 8 + ks) * 16 + tile) * 64 + lane) * 8;
            for (int j = 0; j < 8; j++) dst[j] = f32_to_bf16_host(w[(size_t)row * 256 + k + j]);
          }
  }
}


// One empty launch from this translation unit: the runtime loads a code object (and sets a stream's queue up) at the
// first launch that needs it; crnn_api.hip does these at device set-up so that no later call is the first one.
__global__ void warm_bf16_kernel() {}
void launch_warm_bf16(hipStream_t s) { hipLaunchKernelGGL(warm_bf16_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace crnn
