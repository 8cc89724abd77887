// Race hunt (not product code): the hidden-split recurrence beside a memory-bound aggressor on another stream.  Two
// different projection tensors alternate, so that a value left over from the launch before is recognisable.  Reference:
// the same launches on a quiet device.   tools/rec_race [nt] [iters] [aggressor 0/1]
#include "crnn_kernels.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace crnn;
__global__ void hammer(const float4* a, float4* b, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = a[i];
      v.x += 1.f;
      b[i] = v;
    }
}
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 200, agg = argc > 3 ? atoi(argv[3]) : 1;
  const int B = 256, T = 71;
  const size_t rows = (size_t)B * T;
  std::vector<float> xp[2], wf((size_t)1024 * 256), wr((size_t)1024 * 256);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.f - 0.5f; };
  for (int k = 0; k < 2; k++) { xp[k].resize(rows * 2048); for (auto& v : xp[k]) v = 2.f * rnd(); }
  for (auto& v : wf) v = 0.2f * rnd();
  for (auto& v : wr) v = 0.2f * rnd();
  std::vector<uint16_t> packed((size_t)2 * 1024 * 256 * 2);
  const float uns = pack_whh_h3(wf.data(), wr.data(), packed.data());
  float *d_xp[2], *d_rec;
  void* d_w;
  unsigned* d_sync;
  for (int k = 0; k < 2; k++) { hipMalloc(&d_xp[k], rows * 2048 * 4); hipMemcpy(d_xp[k], xp[k].data(), rows * 2048 * 4, hipMemcpyHostToDevice); }
  hipMalloc(&d_rec, rows * 512 * 4); hipMalloc(&d_w, packed.size() * 2); hipMalloc(&d_sync, 4096);
  hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
  hipMemset(d_sync, 0, 4096);
  const size_t hn = (size_t)64 << 20;   // float4s: 1 GiB each
  float4 *ha, *hb;
  hipMalloc(&ha, hn * 16); hipMalloc(&hb, hn * 16);
  hipMemset(ha, 0, hn * 16);
  hipStream_t s, s2;
  hipStreamCreate(&s); hipStreamCreate(&s2);
  std::vector<float> ref[2], got(rows * 512);
  for (int k = 0; k < 2; k++) {
    ref[k].resize(rows * 512);
    launch_lstm_rec_split_h3(d_xp[k], d_w, d_rec, nullptr, B, T, uns, d_sync, s, 1);
    hipStreamSynchronize(s);
    hipMemcpy(ref[k].data(), d_rec, rows * 512 * 4, hipMemcpyDeviceToHost);
  }
  long nbad = 0, bad_launches = 0;
  for (int it = 0; it < iters; it++) {
    const int k = it & 1;
    if (agg) hipLaunchKernelGGL(hammer, dim3(2048), dim3(256), 0, s2, ha, hb, hn, 1);
    launch_lstm_rec_split_h3(d_xp[k], d_w, d_rec, nullptr, B, T, uns, d_sync, s, nt);
    hipStreamSynchronize(s);
    hipMemcpy(got.data(), d_rec, rows * 512 * 4, hipMemcpyDeviceToHost);
    if (memcmp(got.data(), ref[k].data(), rows * 512 * 4) != 0) {
      bad_launches++;
      // the first wrong element in time order of each (line, dir)
      int shown = 0;
      for (int b = 0; b < B && shown < 6; b++)
        for (int dir = 0; dir < 2 && shown < 6; dir++)
          for (int sidx = 0; sidx < T; sidx++) {
            const int t = dir ? T - 1 - sidx : sidx;
            const float* g = got.data() + ((size_t)b * T + t) * 512 + dir * 256;
            const float* w = ref[k].data() + ((size_t)b * T + t) * 512 + dir * 256;
            int nb = 0, first = -1;
            for (int u = 0; u < 256; u++) if (memcmp(g + u, w + u, 4)) { nb++; if (first < 0) first = u; }
            if (!nb) continue;
            nbad += nb;
            const float* o = ref[k ^ 1].data() + ((size_t)b * T + t) * 512 + dir * 256;   // the launch before wrote this here
            printf("it %d line %d (bl %d) dir %d step %d t %d: %d units differ, first unit %d got %.6f want %.6f | other input's %.6f\n",
                   it, b, b & 15, dir, sidx, t, nb, first, g[first], w[first], o[first]);
            shown++;
            break;   // later steps of this (line, dir) follow from the first
          }
    }
    if (agg) hipStreamSynchronize(s2);
  }
  unsigned h[4];
  hipMemcpy(h, d_sync, 16, hipMemcpyDeviceToHost);
  printf("nt %d iters %d aggressor %d: %ld launches with wrong output, time-out word %u\n", nt, iters, agg, bad_launches, h[0]);
  return bad_launches ? 1 : 0;
}
