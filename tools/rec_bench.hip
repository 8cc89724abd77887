// Timing harness (not product code) for the hidden-split recurrences: launches lstm_rec_split_h3_kernel (nt = 1) or
// lstm_rec_split2_h3_kernel (nt = 2) alone on the device on synthetic projections and prints the time per launch and, from a
// -DREC_PROF build of crnn_kernels.hip, where workgroup (0, 0) spends its steps.
//   tools/rec_bench [nt] [B] [T] [reps]        build: see tools/build_rec_bench.sh
#include "crnn_kernels.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crnn;
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 1, B = argc > 2 ? atoi(argv[2]) : 256, T = argc > 3 ? atoi(argv[3]) : 71;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const size_t rows = (size_t)B * T;
  std::vector<float> xp(rows * 2048), wf((size_t)1024 * 256), wr((size_t)1024 * 256);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (auto& v : xp) v = 2.f * rnd();
  for (auto& v : wf) v = 0.2f * rnd();
  for (auto& v : wr) v = 0.2f * rnd();
  std::vector<uint16_t> packed((size_t)2 * 1024 * 256 * 2);
  const float uns = pack_whh_h3(wf.data(), wr.data(), packed.data());
  float *d_xp, *d_rec;
  void* d_w;
  unsigned* d_sync;
  hipMalloc(&d_xp, xp.size() * 4); hipMalloc(&d_rec, rows * 512 * 4); hipMalloc(&d_w, packed.size() * 2); hipMalloc(&d_sync, 4096);
  hipMemcpy(d_xp, xp.data(), xp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
  hipMemset(d_sync, 0, 4096);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) launch_lstm_rec_split_h3(d_xp, d_w, d_rec, nullptr, B, T, uns, d_sync, s, nt);
  hipStreamSynchronize(s);
  hipMemsetAsync(d_sync, 0, 4096, s);
  hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) launch_lstm_rec_split_h3(d_xp, d_w, d_rec, nullptr, B, T, uns, d_sync, s, nt);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned h[32];
  hipMemcpy(h, d_sync, 128, hipMemcpyDeviceToHost);
  std::vector<float> rec(rows * 512);
  hipMemcpy(rec.data(), d_rec, rec.size() * 4, hipMemcpyDeviceToHost);
  double cs = 0;
  for (float v : rec) cs += v;
  const int iters = T * (nt == 2 ? 2 : 1);
  printf("nt %d B %d T %d: %.4f ms per launch (memset included), %.2f us per step, time-out word %u, checksum %.6f\n", nt, B, T,
         ms / reps, ms / reps * 1e3 / T, h[0], cs);
  // last launch's profile (ticks of 10 ns): per (half-)step averages
  printf("  per %s (us): mfma+issue %.3f  poll/wait %.3f  chunks->lds %.3f  gates+store %.3f  barrier %.3f  | re-polls per step %.2f\n",
         nt == 2 ? "half-step" : "step", h[16] * 0.01 / iters, h[17] * 0.01 / iters, h[18] * 0.01 / iters, h[19] * 0.01 / iters,
         h[21] * 0.01 / iters, (double)h[20] / iters);
  return 0;
}
