// Timing harness (not product code) for wino4_fused_h3_kernel against wino4_gemm_h3(_lw)_kernel + wino4_out_kernel on
// the five layer shapes of a 256-line step.  Build (optionally with -DFUSED_ABL=<bits>, see the kernel):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ocr_amd/csrc -o tools/fused_bench tools/fused_bench.hip
#include "../ocr_amd/csrc/crnn_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crnn;
static void fill_h3(std::vector<uint16_t>& v, size_t nfloat, float scale) {
  v.resize(nfloat * 2);
  for (size_t i = 0; i < nfloat; i += 8)
    for (int e = 0; e < 8; e++) {
      const float x = scale * ((rand() & 0xFFFF) / 32768.f - 1.f);
      const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
      v[2 * i + e] = __builtin_bit_cast(uint16_t, hi);
      v[2 * i + 8 + e] = __builtin_bit_cast(uint16_t, lo);
    }
}
int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  struct Sh { const char* name; int H, W, Cin, N, pool; } sh[5] = {
      {"conv1", 16, 140, 64, 128, 1}, {"conv2", 8, 70, 128, 256, 0}, {"conv3", 8, 70, 256, 256, 0},
      {"conv4", 4, 71, 256, 512, 0}, {"conv5", 4, 71, 512, 512, 0}};
  const int B = 256;
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int l = 0; l < 5; l++) {
    const Sh& q = sh[l];
    if (argc > 2 && atoi(argv[2]) != l + 1) continue;
    const long tiles = wino4_tiles(B, q.H, q.W), tp = wino4_pad(tiles);
    std::vector<uint16_t> hv, hu;
    fill_h3(hv, (size_t)36 * tp * q.Cin, 16000.f);
    fill_h3(hu, (size_t)36 * q.N * q.Cin, 16000.f);
    float *V, *U, *M, *vs, *bias, *y;
    hipMalloc(&V, hv.size() * 2); hipMalloc(&U, hu.size() * 2); hipMalloc(&M, (size_t)36 * tp * q.N * 4);
    hipMalloc(&vs, tp * 4); hipMalloc(&bias, q.N * 4); hipMalloc(&y, (size_t)B * q.H * (q.W + 1) * q.N * 4);
    hipMemcpy(V, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(U, hu.data(), hu.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> one(tp, 1e-6f);
    hipMemcpy(vs, one.data(), tp * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, q.N * 4);
    Wino4Fused f{};
    f.V = V; f.U = U; f.vs = vs; f.bias = bias; f.y = y; f.wvalid = nullptr; f.us = 1e-6f; f.B = B; f.H = q.H; f.W = q.W;
    f.Cin = q.Cin; f.N = q.N; f.act = 1; f.pool = q.pool; f.ntile_pad = (int)tp; f.ntile = (int)tiles;
    Wino4Gemm w{};
    w.V = V; w.U = U; w.M = M; w.ntile_pad = (int)tp; w.N = q.N; w.Cin = q.Cin; w.ncomp = 36; w.h3 = 1;
    float tf = 0, tg = 0, to = 0;
    for (int it = -3; it < reps; it++) {
      float ms;
      hipEventRecord(e0, s); launch_wino4_fused(f, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); if (it >= 0) tf += ms;
      hipEventRecord(e0, s); launch_wino4_gemm(w, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); if (it >= 0) tg += ms;
      hipEventRecord(e0, s); launch_wino4_out(M, bias, y, nullptr, B, q.H, q.W, q.N, 1, q.pool, s, vs, 1e-6f); hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); if (it >= 0) to += ms;
    }
    const double fl = 2.0 * 36 * tiles * q.N * q.Cin;
    printf("%s K=%d N=%d tiles=%ld: fused %.4f ms (%.0f TF fp32-eq) | gemm %.4f + out %.4f = %.4f ms\n", q.name, q.Cin, q.N,
           tiles, tf / reps, fl / (tf / reps) * 1e-9, tg / reps, to / reps, (tg + to) / reps);
    hipFree(V); hipFree(U); hipFree(M); hipFree(vs); hipFree(bias); hipFree(y);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
