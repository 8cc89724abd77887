// Standalone development bench for the "f16x3" GEMM: fp32-accurate products from fp16 MFMAs.
// Every fp32 operand x is stored as two fp16 numbers hi = fp16(x s), lo = fp16(x s - hi) (s a power of two per
// tensor that keeps lo a normal fp16), 8 hi then 8 lo per 8 consecutive k (32 bytes, the same bytes per element as
// fp32), and  A B^T = (hi_a hi_b + hi_a lo_b + lo_a hi_b) / (s_a s_b)  on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: 3 MFMAs at 16x the f32 MFMA rate.  tools/split_mfma_test.hip measures the accuracy (as good as the
// f32 MFMA or better); this file measures the speed of the Winograd batched GEMM M[c] = V[c] U[c]^T on it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/h3_gemm_bench.hip -o tools/h3_gemm_bench
// Run:   tools/h3_gemm_bench [ncomp M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct H3Gemm {
  const void* V;   // [ncomp][ntile_pad][K] split pairs (4 bytes per element)
  const void* U;   // [ncomp][N][K] split pairs
  float* M;        // [ncomp][ntile_pad][N]
  int ntile_pad, N, K, ncomp;
  float unscale;   // 1 / (s_V s_U)
  float* clk;      // PP_CLOCK: block 0's core cycles and 100 MHz ticks
};

constexpr int BK = 32;

// Units of (NWM * 32 WMT) x (NWN * 32 WNT) with 8 waves as NWM (M) x NWN (N), wave tile 32 WMT x 32 WNT; three stages of
// K = 32, direct-to-LDS loads issued by every wave; LDS layout and swizzle of wino4_gemm_kernel (a row's K = 32 step is
// 128 bytes = 8 chunks of 16 bytes: hi(k0-7) lo(k0-7) hi(k8-15) lo(k8-15) ...).
template <int WMT, int WNT, int NWM, int NWN>
__global__ __launch_bounds__(512, 1) void h3_gemm_kernel(const H3Gemm p) {
  constexpr int NT = 512, TBM = NWM * 32 * WMT, TBN = NWN * 32 * WNT, SA = TBM * 32, SB = TBN * 32, NLA = TBM / 64, NLB = TBN / 64;
  static_assert(NWM * NWN == 8, "8 waves");
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // [B0 B1 B2 | A0 A1 A2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  unsigned off[NLA > NLB ? NLA : NLB];
#pragma unroll
  for (int i = 0; i < (NLA > NLB ? NLA : NLB); i++) {
    const int P = i * NT + tid, S = P >> 4, c16 = (P & 15) ^ (S & 15);
    off[i] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
  }
  const int nk = p.K / BK;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 1024u;
  auto dma = [&](i32x4 rsrc, unsigned ldsaddr, unsigned voff, int soff) __attribute__((always_inline)) {
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(ldsaddr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.K * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.K * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  int l_u = L, l_k0 = 0;
  unsigned long long pa, pb, pm_unused;
  bases(l_u, pa, pb, pm_unused);
  auto gload = [&](int stage) __attribute__((always_inline)) {
    const i32x4 a_rsrc = {(int)(unsigned)pa, (int)((pa >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
    const i32x4 b_rsrc = {(int)(unsigned)pb, (int)((pb >> 32) & 0xFFFF), 0x7FFFFFFF, 0x00020000};
    const unsigned sa = lds0 + (3 * SB + stage * SA) * 4, sb = lds0 + stage * SB * 4;
#pragma unroll
    for (int i = 0; i < NLA; i++) dma(a_rsrc, sa + i * NT * 16, off[i], l_k0 * 4);
#pragma unroll
    for (int i = 0; i < NLB; i++) dma(b_rsrc, sb + i * NT * 16, off[i], l_k0 * 4);
    l_k0 += BK;
    if (l_k0 == p.K) {
      l_k0 = 0;
      if (l_u + G < NU) l_u += G;
      unsigned long long pm;
      bases(l_u, pa, pb, pm);
    }
  };
  // fragment addresses (floats inside a stage) of the wave's first 32-row sub-tile; sub-tile t is 1024 t floats further
  int ahi[2], alo[2], bhi[2], blo[2];
  {
    const int Ra = wm * (32 * WMT) + (lane & 31), Rb = wn * (32 * WNT) + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = 4 * j + 2 * h;
      ahi[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | c) ^ ((Ra >> 1) & 15)) << 2);
      alo[j] = (Ra >> 1) * 64 + (((((Ra & 1) << 3) | (c + 1)) ^ ((Ra >> 1) & 15)) << 2);
      bhi[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | c) ^ ((Rb >> 1) & 15)) << 2);
      blo[j] = (Rb >> 1) * 64 + (((((Rb & 1) << 3) | (c + 1)) ^ ((Rb >> 1) & 15)) << 2);
    }
  }
  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int mt = 0; mt < WMT; mt++)
#pragma unroll
    for (int nt = 0; nt < WNT; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
  int kc = 0, c_u = L;
#ifdef STORE_DWORD
  const unsigned moff = (unsigned)((wm * (32 * WMT) + 4 * (lane >> 5)) * p.N + wn * (32 * WNT) + (lane & 31));
#else
  const unsigned moff = (unsigned)((wm * (32 * WMT) + (lane & 31)) * p.N + wn * (32 * WNT) + 4 * (lane >> 5));
#endif
  const float unscale = p.unscale;
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
#ifdef STORE_DWORD
#pragma unroll
    for (int mt = 0; mt < WMT; mt++)
#pragma unroll
      for (int nt = 0; nt < WNT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          dst[(mt * 32 + (r & 3) + 8 * (r >> 2)) * p.N + nt * 32] = acc[mt][nt][r] * unscale;
          acc[mt][nt][r] = 0.f;
        }
#else
#pragma unroll
    for (int mt = 0; mt < WMT; mt++)
#pragma unroll
      for (int nt = 0; nt < WNT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 v = {acc[mt][nt][4 * g] * unscale, acc[mt][nt][4 * g + 1] * unscale, acc[mt][nt][4 * g + 2] * unscale,
                           acc[mt][nt][4 * g + 3] * unscale};
          *(gf32x4*)(dst + mt * 32 * p.N + nt * 32 + 8 * g) = v;
#pragma unroll
          for (int e = 0; e < 4; e++) acc[mt][nt][4 * g + e] = 0.f;
        }
#endif
    kc = 0;
    c_u += G;
  };
  f16x8 ah[2][WMT], al[2][WMT], bh[2][WNT], bl[2][WNT];
  auto frag = [&](int stage, int j, int set) __attribute__((always_inline)) {
    const float* sa = smem + 3 * SB + stage * SA;
    const float* sb = smem + stage * SB;
#pragma unroll
    for (int t = 0; t < WMT; t++) {
      ah[set][t] = *(const f16x8*)(sa + ahi[j] + 1024 * t);
      al[set][t] = *(const f16x8*)(sa + alo[j] + 1024 * t);
    }
#pragma unroll
    for (int t = 0; t < WNT; t++) {
      bh[set][t] = *(const f16x8*)(sb + bhi[j] + 1024 * t);
      bl[set][t] = *(const f16x8*)(sb + blo[j] + 1024 * t);
    }
  };
#ifdef STORE_DWORD
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0)
#else
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
  auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < WNT; nt++)
#pragma unroll
      for (int mt = 0; mt < WMT; mt++)
        acc[mt][nt] = MF(bl[set][nt], ah[set][mt], acc[mt][nt]);
#pragma unroll
    for (int nt = 0; nt < WNT; nt++)
#pragma unroll
      for (int mt = 0; mt < WMT; mt++)
        acc[mt][nt] = MF(bh[set][nt], al[set][mt], acc[mt][nt]);
#pragma unroll
    for (int nt = 0; nt < WNT; nt++)
#pragma unroll
      for (int mt = 0; mt < WMT; mt++)
        acc[mt][nt] = MF(bh[set][nt], ah[set][mt], acc[mt][nt]);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, nx = (st + 1) % 3;
#ifndef ABL_NOFRAG
    frag(st, 1, 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
#ifndef ABL_NODMA
    __builtin_amdgcn_s_waitcnt(0x0070 | (NLA + NLB));   // vmcnt(one step's loads), lgkmcnt(0)
#endif
#ifndef ABL_NOBAR
    __syncthreads();
#endif
#ifndef ABL_NODMA
    gload(st);
#endif
#ifndef ABL_NOFRAG
    frag(nx, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
#ifndef ABL_NOSTORE
    if (++kc == nk) flush();
#endif
  };
  const int nit = my_units * nk;
  gload(0);
  gload(1);
  gload(2);
  constexpr int LPS = NLA + NLB;
  __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * LPS));
  __syncthreads();
  frag(0, 0, 0);
  frag(0, 1, 1);
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  __builtin_amdgcn_s_waitcnt(0x0F70);
}

// 128 x 128 units with NLW dedicated loader waves (waves 8 .. 8 + NLW - 1): they issue all direct-to-LDS loads of the
// block (32 per K step) and are the only waves that wait on vmcnt; the eight MFMA waves read fragments, issue MFMAs
// and store finished products.  One barrier per K step, executed by all waves.
template <int NLW>
__global__ __launch_bounds__(512 + 64 * NLW, 1) void h3_gemm_lw_kernel(const H3Gemm p) {
  constexpr int NWN = 4, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32, QW = 16 / NLW;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.K / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.K * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.K * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  if (wave >= 8) {
    const int lw = wave - 8;
    unsigned off[QW];
#pragma unroll
    for (int q = 0; q < QW; q++) {
      const int P = (lw * QW + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      off[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
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
      if (l_k0 == p.K) {
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
  const unsigned moff = (unsigned)((wm * 64 + (lane & 31)) * p.N + wn * 32 + 4 * (lane >> 5));
  const float unscale = p.unscale;
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 v = {acc[mt][4 * g] * unscale, acc[mt][4 * g + 1] * unscale, acc[mt][4 * g + 2] * unscale,
                         acc[mt][4 * g + 3] * unscale};
        *(gf32x4*)(dst + mt * 32 * p.N + 8 * g) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) acc[mt][4 * g + e] = 0.f;
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
// Ping-pong form: units of 256 (M) x 128 (N), eight MFMA waves as 4 (M) x 2 (N) with a 64 x 64 wave tile (8 fragment reads
// per 12 MFMAs) in TWO GROUPS of four (one wave of each group per SIMD) that run half a K-step apart: while group 0 issues
// the 12 MFMAs of a k16 half-step, group 1 reads its fragments for the next one, and the other way round, a barrier
// between the phases -- so a SIMD's matrix pipe always has one wave's MFMAs and the fragment-read latency of the other wave
// is under them.  Four loader waves issue the direct-to-LDS loads (48 pieces of 1 KiB per K-step, three stages of 48 KB).
// Barrier B_t closes slot t - 1.  Group 0: reads of (K-step s, half j) in slot 4 s + 2 j, MFMAs in the slot after; group 1
// one slot later.  Loader: the batch of K-step s + 2 (stage (s + 2) % 3, last read in K-step s - 1, whose reads both groups
// have waited for by the start of slot 4 s) in slots 4 s + 1 .. 4 s + 3, then vmcnt(12): K-step s + 1's batch has landed
// before B_{4 s + 4}, and group 0 first reads it in slot 4 s + 4.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef PP_NOBAR
#define PP_BARRIER() do {} while (0)
#else
#define PP_BARRIER() __builtin_amdgcn_s_barrier()
#endif
__global__ __launch_bounds__(768, 1) void h3_gemm_pp_kernel(const H3Gemm p) {
  constexpr int TBM = 256, TBN = 128, SA = TBM * 32, SB = TBN * 32, STG = SA + SB;
  __shared__ __attribute__((aligned(1024))) float smem[3 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.K / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.K * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.K * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  if (wave >= 8) {
    const int lw = wave - 8;
    unsigned offa[8], offb[4];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int P = (lw * 8 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offa[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int P = (lw * 4 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offb[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
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
#ifdef PP_NODMA
      if (p.K == 12345) {
#endif
      if (which == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
      } else if (which == 1) {
#pragma unroll
        for (int q = 4; q < 8; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) dma(b_rsrc, ldsB + so + q * 1024, offb[q], l_k0 * 4);
      }
#ifdef PP_NODMA
      }
#endif
      if (which == 2) {
        l_k0 += BK;
        lst = lst == 2 ? 0 : lst + 1;
        if (l_k0 == p.K) {
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
    PP_BARRIER();                          // B_0
#pragma unroll 1
    for (int s = 0; s < nit; s++) {
      const bool more = s + 2 < nit;
      PP_BARRIER();
      if (more) part(0);
      PP_BARRIER();
      if (more) part(1);
      PP_BARRIER();
      if (more) {
        part(2);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
    }
    PP_BARRIER();
    return;
  }
  const int grp = wave >> 2, wm = wave & 3, wn = grp;
#ifdef PP_CLOCK
  const unsigned long long pp_t0 = __builtin_readcyclecounter(), pp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
  // A operand first: a lane ends with column n = lane & 31 of rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5): one store
  // instruction writes two full 128-byte lines
  const unsigned moff = (unsigned)((wm * 64 + 4 * (lane >> 5)) * p.N + wn * 64 + (lane & 31));
  const float unscale = p.unscale;
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
#ifdef ABL_NOSTORE
          if (acc[mt][nt][r] == 12345.f)
#endif
          dst[(mt * 32 + (r & 3) + 8 * (r >> 2)) * p.N + nt * 32] = acc[mt][nt][r] * unscale;
          acc[mt][nt][r] = 0.f;
        }
    kc = 0;
    c_u += G;
  };
  f16x8 ah[2], al[2], bh[2], bl[2];
#ifdef PP_FRAGONCE
  bool first_frag = true;
#endif
  auto frag = [&](int stage, int j) __attribute__((always_inline)) {
    const float* sb = smem + stage * STG;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      ah[t] = *(const f16x8*)(sb + ahi[j] + 1024 * t);
      al[t] = *(const f16x8*)(sb + alo[j] + 1024 * t);
      bh[t] = *(const f16x8*)(sb + bhi[j] + 1024 * t);
      bl[t] = *(const f16x8*)(sb + blo[j] + 1024 * t);
    }
  };
  auto mfma12 = [&]() __attribute__((always_inline)) {
#ifdef PP_M16
    // timing experiment: the same FLOPs as 48 v_mfma_f32_16x16x32_f16 (16 accumulators of 4 registers); results meaningless
    f32x4* a4 = (f32x4*)&acc[0][0];
#pragma unroll
    for (int rep = 0; rep < 3; rep++)
#pragma unroll
      for (int t = 0; t < 16; t++)
        a4[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rep == 1 ? al[t & 1] : ah[t & 1], rep == 0 ? bl[(t >> 1) & 1] : bh[(t >> 1) & 1], a4[t], 0, 0, 0);
    return;
#endif
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
  };
  auto half = [&](int stage, int j) __attribute__((always_inline)) {
#ifdef PP_NOFRAG
    if (p.K == 12345)
#endif
#ifdef PP_FRAGONCE
    if (first_frag)
#endif
    frag(stage, j);
#ifdef PP_FRAGONCE
    first_frag = false;
#endif
    __builtin_amdgcn_sched_barrier(0);
    PP_BARRIER();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfma12();
#ifdef PP_DOUBLE
    mfma12();
#endif
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    half(st, 0);
    PP_BARRIER();
    half(st, 1);
    if (++kc == nk) flush();
    PP_BARRIER();
  };
  PP_BARRIER();                            // B_0
  if (grp == 1) PP_BARRIER();              // group 1 runs one slot behind
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  if (grp == 0) PP_BARRIER();
#ifdef PP_CLOCK
  if (blockIdx.x == 0 && tid == 0) {
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    p.clk[0] = (float)(t1 - pp_t0);
    p.clk[1] = (float)(r1 - pp_r0);
  }
#endif
}

__global__ __launch_bounds__(768, 1) void h3_gemm_pp2_kernel(const H3Gemm p) {
  constexpr int TBM = 256, TBN = 128, SA = TBM * 32, SB = TBN * 32, STG = SA + SB;
  __shared__ __attribute__((aligned(1024))) float smem[3 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  const int nk = p.K / BK;
  const int nit = my_units * nk;
  auto bases = [&](int u, unsigned long long& a, unsigned long long& b, unsigned long long& m) __attribute__((always_inline)) {
    const int nt = u % ntn, q = u / ntn, mt = q % ntm, comp = q / ntm;
    a = (unsigned long long)p.V + ((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.K * 4ull;
    b = (unsigned long long)p.U + ((unsigned long long)comp * p.N + (unsigned long long)nt * TBN) * p.K * 4ull;
    m = (unsigned long long)p.M +
        (((unsigned long long)comp * p.ntile_pad + (unsigned long long)mt * TBM) * p.N + (unsigned long long)nt * TBN) * 4ull;
  };
  if (wave >= 8) {
    const int lw = wave - 8;
    unsigned offa[8], offb[4];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int P = (lw * 8 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offa[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int P = (lw * 4 + q) * 64 + lane, S = P >> 4, c16 = (P & 15) ^ (S & 15);
      offb[q] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
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
        for (int q = 0; q < 6; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
      } else {
#pragma unroll
        for (int q = 6; q < 8; q++) dma(a_rsrc, ldsA + so + q * 1024, offa[q], l_k0 * 4);
#pragma unroll
        for (int q = 0; q < 4; q++) dma(b_rsrc, ldsB + so + q * 1024, offb[q], l_k0 * 4);
        l_k0 += BK;
        lst = lst == 2 ? 0 : lst + 1;
        if (l_k0 == p.K) {
          l_k0 = 0;
          if (l_u + G < NU) l_u += G;
          bases(l_u, pa, pb, pm);
        }
      }
    };
    // prologue: batches 0 and 1 whole; batch 0 has landed before B_0.  Every MFMA wave waits for its fragment reads BEFORE
    // the barrier that ends its read slot, so buffer b % 3 is free once B_{2 b + 2} has passed (group 1 reads K-step b in slot
    // 2 b + 1): batch b + 3 may go out from slot 2 b + 2 = 2 (b + 3) - 4 on.  Batch s + 2: first half in slot 2 s, second half
    // in slot 2 s + 1, and then vmcnt(12): batch s + 1 has landed before B_{2 s + 2}; group 0 reads it in slot 2 s + 2.
    part(0); part(1);
    if (nit > 1) { part(0); part(1); }
    if (nit > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();                          // B_0
#pragma unroll 1
    for (int s = 0; s < nit; s++) {
      const bool more = s + 2 < nit;
      if (more) part(0);
      PP_BARRIER();
      if (more) {
        part(1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
    }
    PP_BARRIER();
    return;
  }
  const int grp = wave >> 2, wm = wave & 3, wn = grp;
#ifdef PP_CLOCK
  const unsigned long long pp_t0 = __builtin_readcyclecounter(), pp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
  // A operand first: a lane ends with column n = lane & 31 of rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5): one store
  // instruction writes two full 128-byte lines
  const unsigned moff = (unsigned)((wm * 64 + 4 * (lane >> 5)) * p.N + wn * 64 + (lane & 31));
  const float unscale = p.unscale;
  auto flush = [&]() __attribute__((always_inline)) {
    unsigned long long ua, ub, pm;
    bases(c_u, ua, ub, pm);
    __attribute__((address_space(1))) float* dst = (__attribute__((address_space(1))) float*)pm + moff;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
#ifdef ABL_NOSTORE
          if (acc[mt][nt][r] == 12345.f)
#endif
          dst[(mt * 32 + (r & 3) + 8 * (r >> 2)) * p.N + nt * 32] = acc[mt][nt][r] * unscale;
          acc[mt][nt][r] = 0.f;
        }
    kc = 0;
    c_u += G;
  };
  f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
  auto frag = [&](int stage, int j) __attribute__((always_inline)) {
    const float* sb = smem + stage * STG;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      ah[j][t] = *(const f16x8*)(sb + ahi[j] + 1024 * t);
      al[j][t] = *(const f16x8*)(sb + alo[j] + 1024 * t);
      bh[j][t] = *(const f16x8*)(sb + bhi[j] + 1024 * t);
      bl[j][t] = *(const f16x8*)(sb + blo[j] + 1024 * t);
    }
  };
  auto mfma12 = [&](int j) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j][mt], bl[j][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j][mt], bh[j][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j][mt], bh[j][nt], acc[mt][nt], 0, 0, 0);
  };
  auto step = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
#ifdef PP_NOFRAG
    if (p.K == 12345)
#endif
    {
      frag(st, 0);
      frag(st, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only: the stage is in registers before the barrier (see the loader)
    PP_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfma12(0);
    mfma12(1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (++kc == nk) flush();
    PP_BARRIER();
  };
  PP_BARRIER();                            // B_0
  if (grp == 1) PP_BARRIER();              // group 1 runs one slot behind
  int it = 0;
#pragma unroll 1
  for (; it + 3 <= nit; it += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  if (it < nit) step(std::integral_constant<int, 0>{});
  if (it + 1 < nit) step(std::integral_constant<int, 1>{});
  if (grp == 0) PP_BARRIER();
#ifdef PP_CLOCK
  if (blockIdx.x == 0 && tid == 0) {
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    p.clk[0] = (float)(t1 - pp_t0);
    p.clk[1] = (float)(r1 - pp_r0);
  }
#endif
}

static float run_pp(const H3Gemm& p, int iters, int which = 1) {
  if (p.ntile_pad % 256 || p.N % 128) return -1.f;
  auto k = which == 2 ? h3_gemm_pp2_kernel : h3_gemm_pp_kernel;
  const int units = p.ncomp * (p.ntile_pad / 256) * (p.N / 128);
  const int g = units < 256 ? 8 * ((units + 7) / 8) : 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(g), dim3(768), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k, dim3(g), dim3(768), 0, 0, p);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int NLW>
static float run_lw(const H3Gemm& p, int iters) {
  auto k = h3_gemm_lw_kernel<NLW>;
  const int units = p.ncomp * (p.ntile_pad / 128) * (p.N / 128);
  const int g = units < 256 ? 8 * ((units + 7) / 8) : 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(g), dim3(512 + 64 * NLW), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k, dim3(g), dim3(512 + 64 * NLW), 0, 0, p);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int WMT, int WNT, int NWM, int NWN>
static float run_shape(const H3Gemm& p, int iters) {
  constexpr int TBM = NWM * 32 * WMT, TBN = NWN * 32 * WNT;
  const size_t lds = 3 * (size_t)(TBM + TBN) * 32 * 4;
  if (p.ntile_pad % TBM || p.N % TBN) return -1.f;
  auto k = h3_gemm_kernel<WMT, WNT, NWM, NWN>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int units = p.ncomp * (p.ntile_pad / TBM) * (p.N / TBN);
  const int g = units < 256 ? 8 * ((units + 7) / 8) : 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(g), dim3(512), lds, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k, dim3(g), dim3(512), lds, 0, p);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

// ------------------------------------------------------------------------------------------------------------
static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

static float pow2_scale(const std::vector<float>& v, float target) {
  float mx = 0;
  for (float x : v) mx = std::fmax(mx, std::fabs(x));
  return mx > 0 ? std::exp2(std::floor(std::log2(target / mx))) : 1.f;
}
// rows of K floats -> rows of K split pairs (8 hi, 8 lo per 8 k)
static void to_split(const std::vector<float>& src, std::vector<_Float16>& dst, float s) {
  dst.resize(src.size() * 2);
  for (size_t i = 0; i < src.size(); i++) {
    const float x = src[i] * s;
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    const size_t blk = i / 8, e = i % 8;
    dst[blk * 16 + e] = hi;
    dst[blk * 16 + 8 + e] = lo;
  }
}

int main(int argc, char** argv) {
  int ncomp = 36, M = 4608, N = 512, K = 512;
  if (argc > 4) { ncomp = atoi(argv[1]); M = atoi(argv[2]); N = atoi(argv[3]); K = atoi(argv[4]); }
  const int iters = argc > 5 ? atoi(argv[5]) : 20;
  srand(3);
  // one component's worth of random data, repeated over the components (the values do not matter for speed)
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  for (auto& x : A) { const double n = nrand(); x = (float)(n > 0 ? n : 0.1 * n); }
  for (auto& x : B) x = (float)(0.03 * nrand());
  const float sa = pow2_scale(A, 8192.f), sb = pow2_scale(B, 8192.f);
  std::vector<_Float16> As, Bs;
  to_split(A, As, sa);
  to_split(B, Bs, sb);
  void *dV, *dU;
  float* dM;
  const size_t vb = (size_t)M * K * 4, ub = (size_t)N * K * 4, mb = (size_t)M * N * 4;
  CK(hipMalloc(&dV, vb * ncomp)); CK(hipMalloc(&dU, ub * ncomp)); CK(hipMalloc((void**)&dM, mb * ncomp));
  for (int c = 0; c < ncomp; c++) {
    CK(hipMemcpy((char*)dV + c * vb, As.data(), vb, hipMemcpyHostToDevice));
    CK(hipMemcpy((char*)dU + c * ub, Bs.data(), ub, hipMemcpyHostToDevice));
  }
  float* dClk;
  CK(hipMalloc((void**)&dClk, 64));
  H3Gemm p{dV, dU, dM, M, N, K, ncomp, 1.f / (sa * sb), dClk};
  const int shape = argc > 6 ? atoi(argv[6]) : 0;
  float ms = shape == 1 ? run_shape<2, 2, 4, 2>(p, iters) : shape == 2 ? run_shape<2, 2, 2, 4>(p, iters)
           : shape == 3 ? run_shape<4, 1, 2, 4>(p, iters) : shape == 20 ? run_pp(p, iters) : shape == 21 ? run_pp(p, iters, 2) : shape == 12 ? run_lw<2>(p, iters) : shape == 14 ? run_lw<4>(p, iters)
           : run_shape<2, 1, 2, 4>(p, iters);
  if (ms < 0) { printf("shape %d does not divide M / N\n", shape); return 0; }
  const double flop = 2.0 * ncomp * M * N * K;
  printf("shape %d ncomp %d M %d N %d K %d: %.4f ms  %.1f TF/s fp32-equivalent (%.1f TF/s of f16 MFMA work), %.2f TB/s of V+U+M\n", shape, ncomp, M, N,
         K, ms, flop / ms / 1e9, 3 * flop / ms / 1e9, (double)ncomp * (vb + ub + mb) / ms / 1e9);
  // check: component 0 and the last one, a sample of entries against float64
  std::vector<float> C((size_t)M * N);
  double worst = 0, rms = 0;
  for (int c : {0, ncomp - 1}) {
    CK(hipMemcpy(C.data(), (char*)dM + c * mb, mb, hipMemcpyDeviceToHost));
    for (int t = 0; t < 4000; t++) {
      const int i = rand() % M, j = rand() % N;
      double s = 0;
      for (int k = 0; k < K; k++) s += (double)A[(size_t)i * K + k] * B[(size_t)j * K + k];
      worst = std::fmax(worst, std::fabs(s - C[(size_t)i * N + j]));
      rms += s * s;
    }
  }
  rms = std::sqrt(rms / 8000);
#ifdef PP_CLOCK
  { float ck[2]; CK(hipMemcpy(ck, dClk, 8, hipMemcpyDeviceToHost));
    printf("block 0: %.0f core cycles, %.0f ticks of 100 MHz -> %.3f GHz\n", ck[0], ck[1], ck[0] / ck[1] * 0.1); }
#endif
  printf("check: max |err| %.3e (rms of C %.3g, relative %.2e)\n", worst, rms, worst / rms);
  return worst / rms < 1e-5 ? 0 : 1;
}
