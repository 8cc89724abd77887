// Timing harness (not product code) for gemm16_dma_kernel on the classifier pre-filter shape (18176 x 5530 x 512, fp16).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ocr_amd/csrc [-DG16_ABL=bits] -o tools/gemm16_bench tools/gemm16_bench.hip
#include "../ocr_amd/csrc/crnn_kernels_bf16.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crnn;
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 18176, N = argc > 2 ? atoi(argv[2]) : 5530, K = argc > 3 ? atoi(argv[3]) : 512;
  const int ldc = (N + 7) & ~7;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
  for (auto& x : ha) { const _Float16 h = (_Float16)((rand() & 0xFFFF) / 32768.f - 1.f); x = __builtin_bit_cast(uint16_t, h); }
  for (auto& x : hw) { const _Float16 h = (_Float16)(0.05f * ((rand() & 0xFFFF) / 32768.f - 1.f)); x = __builtin_bit_cast(uint16_t, h); }
  void *A, *W, *C; float* bias;
  hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&C, (size_t)M * ldc * 2); hipMalloc(&bias, N * 4);
  hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  ConvGemm q{};
  q.A = (const float*)A; q.Wt = (const float*)W; q.bias = bias; q.C = (float*)C; q.M = M; q.N = N; q.K = K; q.ldc = ldc;
  q.Hin = q.Win = q.Hout = q.Wout = 1; q.Cin = K; q.ks = 1;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float t_new = 0, t_old = 0;
  const int reps = 10;
  for (int it = -2; it < reps; it++) {
    float ms;
    hipEventRecord(e0, s); launch_gemm16_dma(q, 1, 2, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); if (it >= 0) t_new += ms;
    hipEventRecord(e0, s); launch_conv_gemm_bf16(q, 2, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); if (it >= 0) t_old += ms;
  }
  const double fl = 2.0 * M * N * K;
  printf("M %d N %d K %d: gemm16_dma %.4f ms (%.0f TF/s) | conv_gemm_bf16<2> %.4f ms (%.0f TF/s)\n", M, N, K, t_new / reps,
         fl / (t_new / reps) * 1e-9, t_old / reps, fl / (t_old / reps) * 1e-9);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
