// Microbenchmark (not product code): latency of a one-word hand-off between two workgroups through global memory, the
// mechanism of lstm_rec_split_h3_kernel's h exchange, as a function of the cache policy of the store / the polling load
// and of whether the two workgroups sit on one XCD.  Every slot is written once (pre-filled with 0xFFFFFFFF), polls are
// bounded.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/xchg_lat tools/xchg_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned* bufA, unsigned* bufB, int rounds, int partner, unsigned* out) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
  if (me < 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)bufA, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bufB, 0, 0x7FFFFFFF, 0x00020000);
  const unsigned lane = threadIdx.x;
  unsigned fails = 0;
  const long long t0 = wall_clock64();
  for (int i = 0; i < rounds; i++) {
    const unsigned off = (unsigned)(i * 64 + lane) * 16u;   // 16 bytes per lane, a fresh 1 KiB per round
    const i32x4 tok = {i, i + 1, i + 2, i + 3};
    if (me == 0) __builtin_amdgcn_raw_buffer_store_b128(tok, ra, off, 0, ST);
    // poll the other side's slot of this round (B polls A's, then answers; A polls B's)
    const __amdgpu_buffer_rsrc_t rp = me == 0 ? rb : ra;
    unsigned spins = 0;
    i32x4 v;
    do {
      asm volatile("" ::: "memory");   // (the builtin load is a pure read to the compiler: keep it inside the loop)
      v = __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, LD);
    } while ((v[0] == -1 || v[1] == -1 || v[2] == -1 || v[3] == -1) && ++spins < 200000u);
    if (spins >= 200000u) { fails++; break; }
    if (me == 1) __builtin_amdgcn_raw_buffer_store_b128(tok, rb, off, 0, ST);
  }
  const long long t1 = wall_clock64();
  if (lane == 0) { out[me * 4 + 0] = (unsigned)(t1 - t0); out[me * 4 + 1] = fails; out[me * 4 + 2] = xcc; }
}
template <int ST, int LD>
void run(const char* name, unsigned* A, unsigned* B, unsigned* out, int rounds, int partner, size_t bytes) {
  hipMemset(A, 0xFF, bytes); hipMemset(B, 0xFF, bytes); hipMemset(out, 0, 64);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((pingpong<ST, LD>), dim3(16), dim3(64), 0, 0, A, B, rounds, partner, out);
  hipDeviceSynchronize();
  unsigned h[8];
  hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  // wall_clock64 ticks at 100 MHz
  printf("%-28s partner %2d  xcc %u / %u  %7.1f ns per one-way hand-off  fails %u %u\n", name, partner, h[2] & 0xF, h[6] & 0xF,
         h[0] * 10.0 / rounds / 2.0, h[1], h[5]);
}
int main() {
  const int rounds = 2000;
  const size_t bytes = (size_t)rounds * 1024;
  unsigned *A, *B, *out;
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&out, 64);
  for (int partner : {1, 8, 1, 8}) {
    run<16, 16>("store sc1, load sc1 (now)", A, B, out, rounds, partner, bytes);
    run<0, 16>("store plain, load sc1", A, B, out, rounds, partner, bytes);
    run<1, 1>("store sc0, load sc0", A, B, out, rounds, partner, bytes);
    run<0, 1>("store plain, load sc0", A, B, out, rounds, partner, bytes);
    run<17, 17>("store sc0 sc1, load sc0 sc1", A, B, out, rounds, partner, bytes);
    run<2, 2>("store nt, load nt", A, B, out, rounds, partner, bytes);
  }
  return 0;
}
