// Semantics check of buffer_load_dwordx4 ... lds on gfx950: where does lane l's 16 bytes land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* g, float* out) {
  __shared__ __attribute__((aligned(16))) float s[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7FFFFFFF, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane l fetches global chunk (63 - l) of its wave's 1 KiB; destination = wave's 1 KiB in LDS (+ 4096 B)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(s + 1024 + wave * 256), 16, (wave * 64 + (63 - lane)) * 16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = s[i];
}
int main() {
  std::vector<float> h(4096); for (int i = 0; i < 4096; i++) h[i] = i;
  float *g, *o; hipMalloc(&g, 16384); hipMalloc(&o, 8192);
  hipMemcpy(g, h.data(), 16384, hipMemcpyHostToDevice);
  k<<<1, 128>>>(g, o);
  std::vector<float> r(2048); hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; w++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
    const float want = (w * 64 + (63 - l)) * 4 + e, got = r[1024 + w * 256 + l * 4 + e];
    if (want != got) { if (bad < 8) printf("wave %d lane %d e %d want %g got %g\n", w, l, e, want, got); bad++; }
  }
  for (int i = 0; i < 1024; i++) if (r[i] != -1.f) bad++;
  printf("lds dma b128: %s (%d mismatches)\n", bad ? "UNEXPECTED LAYOUT" : "lane l -> base + 16*l OK", bad);
  return 0;
}
