// Peak issue rate of v_mfma_f32_32x32x2_f32 vs number of independent accumulators per wave and
// waves per SIMD (experiment; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the fused Winograd kernel's inner step without global loads: per k-group 3 ds_read_b128 -> 8 MFMAs
// MODE 0: as in the kernel (reads one k-group ahead); 1: + barrier per step; 2: no LDS reads
template <int MODE>
__global__ __launch_bounds__(256, 2) void k2(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[194 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = threadIdx.x; i < 192 * 36; i += 256) smem[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc[2];
  for (int r = 0; r < 16; r++) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  const int ao = (wm * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
  const int bo = 128 * 36 + (wn * 32 + (lane & 31)) * 36 + 4 * (lane >> 5);
  f32x4 a0[2], a1[2], b0[2];
  a0[0] = *(const f32x4*)(smem + ao); a1[0] = *(const f32x4*)(smem + ao + 32 * 36); b0[0] = *(const f32x4*)(smem + bo);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kg = 0; kg < 4; kg++) {
      const int c = kg & 1;
      if (MODE != 2) {
        // plain LDS reads; the row shifts with the iteration so nothing can be hoisted
        const int nx = ((kg + 1) & 3) * 8 + (it & 1) * 36;
        if (MODE == 6) {   // two 8-byte reads per fragment
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          f32x2 t0 = *(const f32x2*)(smem + ao + nx), t1 = *(const f32x2*)(smem + ao + nx + 2);
          f32x2 t2 = *(const f32x2*)(smem + ao + 32 * 36 + nx), t3 = *(const f32x2*)(smem + ao + 32 * 36 + nx + 2);
          f32x2 t4 = *(const f32x2*)(smem + bo + nx), t5 = *(const f32x2*)(smem + bo + nx + 2);
          a0[c ^ 1] = f32x4{t0[0], t0[1], t1[0], t1[1]};
          a1[c ^ 1] = f32x4{t2[0], t2[1], t3[0], t3[1]};
          b0[c ^ 1] = f32x4{t4[0], t4[1], t5[0], t5[1]};
        } else {
          a0[c ^ 1] = *(const f32x4*)(smem + ao + nx);
          if (MODE != 4) a1[c ^ 1] = *(const f32x4*)(smem + ao + 32 * 36 + nx);
          if (MODE != 4 && MODE != 5) b0[c ^ 1] = *(const f32x4*)(smem + bo + nx);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (MODE == 3) {   // accumulators forced into AccVGPRs
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(a0[c][j]), "v"(b0[c][j]));
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(a1[c][j]), "v"(b0[c][j]));
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c][j], b0[c][j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c][j], b0[c][j], acc[1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1) __syncthreads();
  }
  float s = 0;
  for (int r = 0; r < 16; r++) s += acc[0][r] + acc[1][r];
  if (s == 12345.f) out[0] = s;
}
template <int MODE> void run2(float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k2<MODE><<<512, 256>>>(d, 100);
  hipEventRecord(e0);
  k2<MODE><<<512, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 512.0 * 4 * iters * 32 * 4096.0;
  printf("k2 mode %d (2 waves/SIMD): %.3f ms  %.1f TF/s\n", MODE, ms, fl / ms / 1e9);
}
template <int NACC> void run(int wps, float* d) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<256 * wps, 256>>>(d, 100, 1.f, 2.f);
  hipEventRecord(e0);
  k<NACC><<<256 * wps, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 256.0 * wps * 4 * iters * 16 * 4096.0;
  printf("nacc %d waves/SIMD %d: %.3f ms  %.1f TF/s\n", NACC, wps, ms, fl / ms / 1e9);
}
// core clock during an MFMA kernel: s_memtime (shader clock) against s_memrealtime (100 MHz)
__global__ __launch_bounds__(256) void kclk(unsigned long long* st, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float x = 1.f + threadIdx.x, y = 2.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { st[blockIdx.x * 2] = c1 - c0; st[blockIdx.x * 2 + 1] = r1 - r0; }
  if (s == 12345.f) st[0] = 0;
}
__global__ __launch_bounds__(256) void kstream(const float4* a, float4* b, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      float4 v = a[i]; v.x += 1.f; b[i] = v;
    }
}
void clocks() {
  unsigned long long* st; hipMalloc(&st, 1024 * 16);
  float4 *a, *b; const size_t n = 64u << 20; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);   // 2 x 1 GiB
  hipMemset(a, 0, n * 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int mode = 0; mode < 2; mode++) {
    hipDeviceSynchronize();
    if (mode) kstream<<<2048, 256, 0, s2>>>(a, b, n, 8);
    kclk<<<1024, 256, 0, s1>>>(st, 4000);
    hipDeviceSynchronize();
    unsigned long long h[2048]; hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
    double f = 0; for (int i = 0; i < 1024; i++) f += (double)h[2 * i] / (double)h[2 * i + 1];
    printf("MFMA kernel %s: core clock %.0f MHz (s_memtime / s_memrealtime x 100 MHz)\n",
           mode ? "with an HBM streaming kernel beside it" : "alone", f / 1024 * 100);
  }
}
int main() {
  float* d; hipMalloc(&d, 4);
  for (int wps = 1; wps <= 4; wps *= 2) { run<1>(wps, d); run<2>(wps, d); run<4>(wps, d); }
  clocks();
  run2<0>(d); run2<1>(d); run2<2>(d); run2<3>(d); run2<4>(d); run2<5>(d); run2<6>(d);
  return 0;
}
