// Accuracy experiment (not product code): an fp32 GEMM C = A B^T done three ways on gfx950 --
//   (a) v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate),
//   (b) fp16 two-way split, three products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16
//       (operands scaled by a power of two per tensor so that the low parts stay normal fp16),
//   (c) the same three products accumulated in THREE separate accumulators and added at the end,
// each against a float64 host reference.  Prints max / mean |error| relative to the RMS of C.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/split_mfma_test tools/split_mfma_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void gemm_f32(const float* A, const float* B, float* C, int M, int N, int K) {
  const int lane = threadIdx.x, tm = blockIdx.x * 32, tn = blockIdx.y * 32;
  f32x16 acc = {};
  const float* a = A + (long)(tm + (lane & 31)) * K + (lane >> 5);
  const float* b = B + (long)(tn + (lane & 31)) * K + (lane >> 5);
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
  for (int r = 0; r < 16; r++)
    C[(long)(tm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * N + tn + (lane & 31)] = acc[r];
}

template <int MODE>
__global__ void gemm_split(const _Float16* Ah, const _Float16* Al, const _Float16* Bh, const _Float16* Bl, float* C,
                           int M, int N, int K, float unscale) {
  const int lane = threadIdx.x, tm = blockIdx.x * 32, tn = blockIdx.y * 32;
  f32x16 acc = {}, acc1 = {}, acc2 = {};
  const long ao = (long)(tm + (lane & 31)) * K + 8 * (lane >> 5), bo = (long)(tn + (lane & 31)) * K + 8 * (lane >> 5);
  for (int k = 0; k < K; k += 16) {
    const f16x8 ah = *(const f16x8*)(Ah + ao + k), al = *(const f16x8*)(Al + ao + k);
    const f16x8 bh = *(const f16x8*)(Bh + bo + k), bl = *(const f16x8*)(Bl + bo + k);
    if (MODE == 0) {   // one accumulator, small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    } else if (MODE == 1) {   // three accumulators
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    } else {   // hi*hi only (plain fp16 GEMM) for scale
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  if (MODE == 1) acc = acc + (acc1 + acc2);
  for (int r = 0; r < 16; r++)
    C[(long)(tm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * N + tn + (lane & 31)] = acc[r] * unscale;
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

static float pow2_scale(const std::vector<float>& v, float target) {
  float mx = 0;
  for (float x : v) mx = std::fmax(mx, std::fabs(x));
  return mx > 0 ? std::exp2(std::floor(std::log2(target / mx))) : 1.f;
}

int main(int argc, char** argv) {
  const int M = 256, N = 256;
  const int K = argc > 1 ? atoi(argv[1]) : 512;
  const int dist = argc > 2 ? atoi(argv[2]) : 0;   // 0: relu(normal) x normal*0.02; 1: wide dynamic range (lognormal)
  srand(1);
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  for (auto& x : A) {
    const double n = nrand();
    x = dist == 0 ? (float)(n > 0 ? n : 0.0) : (float)(std::exp(3.0 * nrand()) * (n > 0 ? 1 : -1));
  }
  for (auto& x : B) x = (float)(dist == 0 ? 0.02 * nrand() : 0.02 * std::exp(2.0 * nrand()) * (nrand() > 0 ? 1 : -1));
  std::vector<double> R((size_t)M * N);
  double rms = 0;
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      double s = 0;
      for (int k = 0; k < K; k++) s += (double)A[(size_t)i * K + k] * B[(size_t)j * K + k];
      R[(size_t)i * N + j] = s;
      rms += s * s;
    }
  rms = std::sqrt(rms / ((double)M * N));
  const float tgt = argc > 3 ? (float)atof(argv[3]) : 8192.f;
  const float sa = pow2_scale(A, tgt), sb = pow2_scale(B, 8192.f);
  printf("A scaled to max <= %g\n", tgt);
  std::vector<_Float16> Ah(A.size()), Al(A.size()), Bh(B.size()), Bl(B.size());
  for (size_t i = 0; i < A.size(); i++) { const float x = A[i] * sa; Ah[i] = (_Float16)x; Al[i] = (_Float16)(x - (float)Ah[i]); }
  for (size_t i = 0; i < B.size(); i++) { const float x = B[i] * sb; Bh[i] = (_Float16)x; Bl[i] = (_Float16)(x - (float)Bh[i]); }
  float *dA, *dB, *dC;
  _Float16 *dAh, *dAl, *dBh, *dBl;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMalloc(&dAh, A.size() * 2); hipMalloc(&dAl, A.size() * 2); hipMalloc(&dBh, B.size() * 2); hipMalloc(&dBl, B.size() * 2);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dAh, Ah.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dAl, Al.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dBh, Bh.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dBl, Bl.data(), B.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> C((size_t)M * N);
  auto report = [&](const char* name) {
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, mean = 0;
    for (size_t i = 0; i < C.size(); i++) { const double e = std::fabs((double)C[i] - R[i]); mx = std::fmax(mx, e); mean += e; }
    printf("K=%d dist=%d %-28s max %.3e mean %.3e (relative to rms(C) = %.4g)\n", K, dist, name, mx / rms, mean / C.size() / rms, rms);
  };
  const dim3 g(M / 32, N / 32);
  hipLaunchKernelGGL(gemm_f32, g, dim3(64), 0, 0, dA, dB, dC, M, N, K);
  report("f32 mfma");
  const float un = 1.f / (sa * sb);
  hipLaunchKernelGGL(gemm_split<0>, g, dim3(64), 0, 0, dAh, dAl, dBh, dBl, dC, M, N, K, un);
  report("f16x3 one accumulator");
  hipLaunchKernelGGL(gemm_split<1>, g, dim3(64), 0, 0, dAh, dAl, dBh, dBl, dC, M, N, K, un);
  report("f16x3 three accumulators");
  hipLaunchKernelGGL(gemm_split<2>, g, dim3(64), 0, 0, dAh, dAl, dBh, dBl, dC, M, N, K, un);
  report("f16 hi*hi only");
  // host fp32 sequential sum for comparison
  {
    for (int i = 0; i < M; i++)
      for (int j = 0; j < N; j++) {
        float s = 0;
        for (int k = 0; k < K; k++) s = std::fmaf(A[(size_t)i * K + k], B[(size_t)j * K + k], s);
        C[(size_t)i * N + j] = s;
      }
    double mx = 0, mean = 0;
    for (size_t i = 0; i < C.size(); i++) { const double e = std::fabs((double)C[i] - R[i]); mx = std::fmax(mx, e); mean += e; }
    printf("K=%d dist=%d %-28s max %.3e mean %.3e\n", K, dist, "host fp32 fmaf chain", mx / rms, mean / C.size() / rms);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
