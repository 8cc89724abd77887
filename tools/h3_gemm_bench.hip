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
};

constexpr int BK = 32;

// 128 x 128 units, 8 waves as 2 (M) x 4 (N), wave tile 64 x 32; three 32 KB stages of K = 32, direct-to-LDS loads
// issued by every wave (2 + 2 per step); LDS layout and swizzle of wino4_gemm_kernel (a row's K = 32 step is 128 bytes =
// 8 chunks of 16 bytes: hi(k0-7) lo(k0-7) hi(k8-15) lo(k8-15) ...).
__global__ __launch_bounds__(512, 1) void h3_gemm_kernel(const H3Gemm p) {
  constexpr int NWN = 4, NT = 512, TBM = 128, TBN = 128, SA = TBM * 32, SB = TBN * 32, NLA = 2;
  __shared__ __attribute__((aligned(1024))) float smem[3 * (SA + SB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int ntn = p.N / TBN, ntm = p.ntile_pad / TBM;
  const int NU = p.ncomp * ntm * ntn, G = gridDim.x;
  const int chunk = G >> 3;
  const int L = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (L >= NU) return;
  const int my_units = (NU - L + G - 1) / G;
  unsigned aoff[NLA], boff[2];
#pragma unroll
  for (int i = 0; i < NLA; i++) {
    const int P = i * NT + tid, S = P >> 4, c16 = (P & 15) ^ (S & 15);
    aoff[i] = (unsigned)(((2 * S + (c16 >> 3)) * p.K + (c16 & 7) * 4) * 4);
    boff[i] = aoff[i];
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
    for (int i = 0; i < NLA; i++) dma(a_rsrc, sa + i * NT * 16, aoff[i], l_k0 * 4);
#pragma unroll
    for (int i = 0; i < 2; i++) dma(b_rsrc, sb + i * NT * 16, boff[i], l_k0 * 4);
    l_k0 += BK;
    if (l_k0 == p.K) {
      l_k0 = 0;
      if (l_u + G < NU) l_u += G;
      unsigned long long pm;
      bases(l_u, pa, pb, pm);
    }
  };
  // fragment addresses (floats inside a stage): chunk c of row R sits at (R >> 1) * 64 + ((((R & 1) << 3) | c) ^ ((R >> 1) & 15)) * 4;
  // k16-step j of a K = 32 step: hi = chunk 4 j + 2 h, lo = the chunk after it (h = lane >> 5)
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
  __builtin_amdgcn_s_waitcnt(0x0F70);
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
  H3Gemm p{dV, dU, dM, M, N, K, ncomp, 1.f / (sa * sb)};
  const int units = ncomp * (M / 128) * (N / 128);
  const int g = units < 256 ? 8 * ((units + 7) / 8) : 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(h3_gemm_kernel, dim3(g), dim3(512), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(h3_gemm_kernel, dim3(g), dim3(512), 0, 0, p);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double flop = 2.0 * ncomp * M * N * K;
  printf("ncomp %d M %d N %d K %d: %.4f ms  %.1f TF/s fp32-equivalent (%.1f TF/s of f16 MFMA work), %.2f TB/s of V+U+M\n", ncomp, M, N,
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
  printf("check: max |err| %.3e (rms of C %.3g, relative %.2e)\n", worst, rms, worst / rms);
  return worst / rms < 1e-5 ? 0 : 1;
}
