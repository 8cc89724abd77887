// Co-run test (test tooling, not product code).  Two questions, both about kernels of DIFFERENT streams sharing a CU:
//  1. does conv0_win4_h3_kernel (victim, stream 1; B = 1: eight blocks) give the same V as alone while an "aggressor"
//     free-runs on stream 2?        tools/corun_test <aggressor> <iterations> [W] [unfused]
//       aggressors: 0 none, 1 / 5 / 6 conv_gemm_bf16_kernel (fp16 / bf16 / fp32 output: the classifier filter of one crop),
//       7 conv_gemm_kernel (f32 MFMA), 2 / 4 / 12 LDS spin, 3 / 15 VALU spin, 8-11, 13, 14 pure MFMA spin (11: bf16
//       32x32x16 on 1024 blocks, 14: f16, 13: f32 32x32x2)
//  2. which packed-fp32 instruction forms go wrong beside an MFMA kernel?   CORUN_PK=<form> tools/corun_test <aggressor> <launches>
//       forms: 0 v_pk_fma_f32 op_sel:[0,1,0], 1 op_sel_hi:[1,0,1], 2 as 0 with s_nop 4 between, 3 no modifier,
//       5 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], 6 v_pk_mul_f32 op_sel:[0,1], 7 v_pk_fma_f32 op_sel:[1,0,0],
//       8 v_pk_add_f16 op_sel:[0,1], 9 v_pk_max_f16 op_sel:[0,1] op_sel_hi:[1,0] (16-bit packed forms: never wrong);
//       the LOW result of a chain is compared with the same chain on v_fma_f32 / v_add_f32 / v_mul_f32
// Measured on MI355X (round 2): with the kernel as first built (packed FMAs in its conv0 phase) aggressors 1 / 5 / 6 /
// 11 / 14 changed V in 3097 / 3173 / 2170 / 247 / 18 of 6000-8000 runs, every other aggressor in none; forms 0 / 2 / 5 / 6
// gave 192-720 / 1888 / 480 / 400 wrong low results in 2-2.6e9 chains beside aggressor 11, forms 1 / 3 / 7 / 8 / 9 none, and
// no form any without the aggressor.  With the kernel compiled without packed fp32 (crnn_kernels_nopk.hip): none.
// Exit status 1 when anything differs.  Built by ocr_amd/csrc/Makefile (target tools), run by tests/test_hip_corun.py.
#include "crnn_kernels.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace crnn;

__global__ void cmp_kernel(const uint32_t* a, const uint32_t* ref, long n, unsigned* cnt) {
  unsigned bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) bad += a[i] != ref[i];
  if (bad) atomicAdd(cnt, bad);
}

// dummy aggressors: LDS bytes as a template parameter; every thread writes and reads its own words for a while
template <int LDSB>
__global__ __launch_bounds__(256) void lds_spin_kernel(float* out, int iters) {
  __shared__ float buf[LDSB / 4 > 0 ? LDSB / 4 : 1];
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    for (int i = threadIdx.x; i < LDSB / 4; i += 256) buf[i] = (float)(it + i);
    __syncthreads();
    for (int i = threadIdx.x; i < LDSB / 4; i += 256) acc += buf[(i * 7 + it) % (LDSB / 4)];
    __syncthreads();
  }
  if (acc == -1.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void valu_spin_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int it = 0; it < iters * 64; it++) a = a * 1.0001f + 0.5f;
  if (a == -1.f) out[0] = a;
}

typedef float f2v __attribute__((ext_vector_type(2)));
// micro-victims: a chain of 8 dependent packed FMAs in one asm block against the same chain on v_fma_f32
//   FORM 0: op_sel:[0,1,0] (both halves use x.hi), s_nop 0 between     1: op_sel_hi:[1,0,1] (both use x.lo), s_nop 0
//   FORM 2: op_sel:[0,1,0], s_nop 4 between                             3: no op_sel, s_nop 0
//   FORM 4: op_sel:[0,1,0], independent (not chained) results summed later
#define PKSTEP(SEL, NOP) "v_pk_fma_f32 %0, %1, %2, %0 " SEL "\n\ts_nop " NOP "\n\t"
// 16-bit packed forms with op_sel (the compiler uses them all the time): form 8 v_pk_add_f16 op_sel:[0,1], 9 v_pk_max_f16
// op_sel:[0,1] op_sel_hi:[1,0]; reference: the same instruction without op_sel on an operand swizzled by integer ops
template <int FORM>
__global__ __launch_bounds__(256) void pk16_victim_kernel(unsigned* bad, int iters) {
  unsigned nbad = 0;
  for (int it = 0; it < iters; it++) {
    unsigned a = 0x3c003800u + (threadIdx.x << 3) + ((unsigned)it & 0xFFu);          // two small positive halves
    unsigned b = 0x34003a00u + (threadIdx.x << 1) + (((unsigned)it >> 3) & 0x3Fu);
    unsigned r = a, e = a;
    if (FORM == 8) {
      const unsigned bs = (b >> 16) | (b & 0xFFFF0000u);   // {b.hi, b.hi}
      asm volatile("v_pk_add_f16 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0\n\tv_pk_add_f16 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0\n\t"
                   "v_pk_add_f16 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0\n\tv_pk_add_f16 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0" : "+v"(r) : "v"(b));
      asm volatile("v_pk_add_f16 %0, %0, %1\n\ts_nop 1\n\tv_pk_add_f16 %0, %0, %1\n\ts_nop 1\n\t"
                   "v_pk_add_f16 %0, %0, %1\n\ts_nop 1\n\tv_pk_add_f16 %0, %0, %1\n\ts_nop 1" : "+v"(e) : "v"(bs));
    } else {
      const unsigned bs = (b >> 16) | (b << 16);           // {b.hi, b.lo}
      asm volatile("v_pk_max_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0" : "+v"(r) : "v"(b));
      asm volatile("v_pk_max_f16 %0, %0, %1\n\ts_nop 1" : "+v"(e) : "v"(bs));
    }
    nbad += r != e;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int FORM>
__global__ __launch_bounds__(256) void pk_victim_kernel(unsigned* bad, int iters) {
  const float t = (float)(threadIdx.x + blockIdx.x * 256);
  f2v w = {0.001f * t + 0.5f, 0.002f * t - 0.25f};
  unsigned nbad = 0;
  for (int it = 0; it < iters; it++) {
    f2v x = {0.37f * (float)(it & 1023) + 0.01f * t, 0.11f * (float)(it & 511) - 0.02f * t};
    f2v r = {0.125f, -0.5f};
    float e0 = r[0], e1 = r[1];
    const float xs = (FORM == 1) ? x[0] : x[1];
    if (FORM == 0) asm volatile(PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0")
                                PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0") PKSTEP("op_sel:[0,1,0]", "0")
                                : "+v"(r) : "v"(w), "v"(x));
    else if (FORM == 1) asm volatile(PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0")
                                     PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0") PKSTEP("op_sel_hi:[1,0,1]", "0")
                                     : "+v"(r) : "v"(w), "v"(x));
    else if (FORM == 2) asm volatile(PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4")
                                     PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4") PKSTEP("op_sel:[0,1,0]", "4")
                                     : "+v"(r) : "v"(w), "v"(x));
    else if (FORM == 3) asm volatile(PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0") PKSTEP("", "0")
                                     : "+v"(r) : "v"(w), "v"(x));
    if (FORM == 5) {   // r = {r.lo + x.hi, r.hi + x.lo}
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\t"
                   "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0" : "+v"(r) : "v"(x));
      for (int k = 0; k < 4; k++) {
        asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 1" : "+v"(e0) : "v"(x[1]));
        asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 1" : "+v"(e1) : "v"(x[0]));
      }
    } else if (FORM == 6) {   // r = {r.lo * x.hi, r.hi * x.hi}
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\ts_nop 0\n\t"
                   "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\ts_nop 0" : "+v"(r) : "v"(x), "v"(w));
      for (int k = 0; k < 2; k++) {
        asm volatile("v_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_mul_f32 %0, %0, %2\n\ts_nop 1" : "+v"(e0) : "v"(x[1]), "v"(w[1]));
        asm volatile("v_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_mul_f32 %0, %0, %2\n\ts_nop 1" : "+v"(e1) : "v"(x[1]), "v"(w[1]));
      }
    } else if (FORM == 7) {   // op_sel on src0: r = {w.hi * x.lo + r.lo, w.hi * x.hi + r.hi}
      asm volatile(PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0")
                   PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0") PKSTEP("op_sel:[1,0,0]", "0")
                   : "+v"(r) : "v"(w), "v"(x));
      for (int k = 0; k < 8; k++) {
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e0) : "v"(w[1]), "v"(x[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e1) : "v"(w[1]), "v"(x[1]));
      }
    } else if (FORM == 3) {
      for (int k = 0; k < 8; k++) {
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e0) : "v"(w[0]), "v"(x[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e1) : "v"(w[1]), "v"(x[1]));
      }
    } else {
      for (int k = 0; k < 8; k++) {
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e0) : "v"(w[0]), "v"(xs));
        asm volatile("v_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(e1) : "v"(w[1]), "v"(xs));
      }
    }
    (void)e1;   // only the LOW half is checked (the form under test redirects the low result's source)
    nbad += __builtin_bit_cast(unsigned, r[0]) != __builtin_bit_cast(unsigned, e0);
  }
  if (nbad) atomicAdd(bad, nbad);
}
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters) {
  f32x16v acc0 = {}, acc1 = {};
  bf16x8v a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(e * 0.5f); }
  for (int it = 0; it < iters * 16; it++) {
    if (KIND == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    } else if (KIND == 1) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)b[0], (float)a[1], acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, b), __builtin_bit_cast(f16x8v, a), acc1, 0, 0, 0);
    }
  }
  float sum = 0.f;
  for (int e = 0; e < 16; e++) sum += acc0[e] + acc1[e];
  if (sum == -1.f) out[0] = sum;
}

int main(int argc, char** argv) {
  const int aggr = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = argc > 2 ? atoi(argv[2]) : 5000;
  const int W = argc > 3 ? atoi(argv[3]) : 256;
  const int unfused = argc > 4 ? atoi(argv[4]) : 0;   // 1: victim = conv0_pool_kernel + wino4_in_h3_kernel; 2: cls_select_kernel (fp16)
  if (getenv("CORUN_PK")) {   // micro-victim mode: tools/corun_test <aggressor> <iters>   with CORUN_PK=<form>
    const int form = atoi(getenv("CORUN_PK"));
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    unsigned* bad; float* out;
    hipMalloc(&bad, 64); hipMalloc(&out, 256); hipMemset(bad, 0, 64);
    for (int it = 0; it < iters; it++) {
      if (aggr == 11) hipLaunchKernelGGL(mfma_spin_kernel<0>, dim3(1024), dim3(256), 0, s2, out, 40);
      if (aggr == 14) hipLaunchKernelGGL(mfma_spin_kernel<2>, dim3(1024), dim3(256), 0, s2, out, 40);
      if (aggr == 13) hipLaunchKernelGGL(mfma_spin_kernel<1>, dim3(1024), dim3(256), 0, s2, out, 40);
      switch (form) {
        case 0: hipLaunchKernelGGL(pk_victim_kernel<0>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 1: hipLaunchKernelGGL(pk_victim_kernel<1>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 2: hipLaunchKernelGGL(pk_victim_kernel<2>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 8: hipLaunchKernelGGL(pk16_victim_kernel<8>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 9: hipLaunchKernelGGL(pk16_victim_kernel<9>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 5: hipLaunchKernelGGL(pk_victim_kernel<5>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 6: hipLaunchKernelGGL(pk_victim_kernel<6>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        case 7: hipLaunchKernelGGL(pk_victim_kernel<7>, dim3(64), dim3(256), 0, s1, bad, 400); break;
        default: hipLaunchKernelGGL(pk_victim_kernel<3>, dim3(64), dim3(256), 0, s1, bad, 400); break;
      }
      if (it % 32 == 31) { hipStreamSynchronize(s1); hipStreamSynchronize(s2); }
    }
    hipStreamSynchronize(s1); hipStreamSynchronize(s2);
    unsigned hb = 0;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("pk form %d, aggressor %d: %u wrong low results of %ld chains\n", form, aggr, hb, (long)iters * 64 * 256 * 400);
    return hb != 0;
  }
  const int B = 1, W1 = W / 2;
  const long nt = wino4_tiles(B, 16, W1), tp = wino4_pad(nt);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  srand(1);
  std::vector<float> hin((size_t)B * 32 * W), hw(9 * 64), hb(64);
  for (auto& v : hin) v = (rand() & 0xFF) / 127.5f - 1.f;
  for (auto& v : hw) v = ((rand() & 0xFFFF) / 32768.f - 1.f) * 0.3f;
  for (auto& v : hb) v = ((rand() & 0xFFFF) / 32768.f - 1.f) * 0.1f;
  float *in, *w9, *bias, *V, *Vref, *vs, *out;
  unsigned* cnt;
  const size_t vbytes = (size_t)36 * tp * 64 * 4;
  hipMalloc(&in, hin.size() * 4); hipMalloc(&w9, hw.size() * 4); hipMalloc(&bias, 256);
  hipMalloc(&V, vbytes); hipMalloc(&Vref, vbytes); hipMalloc(&vs, tp * 4); hipMalloc(&out, 256);
  hipMalloc(&cnt, (size_t)iters * 4);
  float* p0;
  hipMalloc(&p0, (size_t)B * 16 * W1 * 64 * 4);
  // victim 2: the classifier's candidate selection + exact scoring (cls_select_kernel, fp16 logits: the other kernel that
  // lives in crnn_kernels_nopk.hip) on 2048 rows of random activations; its output (the arg-max indices) goes to the
  // first words of Vout, the rest of Vout stays zero
  const int CM = 2048, CN = 5530, CK = 512, cldl = (CN + 7) & ~7;
  float *cA = nullptr, *cW = nullptr, *cbias = nullptr, *canorm = nullptr, *cwnorm = nullptr;
  uint16_t *cA16 = nullptr, *cW16 = nullptr, *cL = nullptr;
  float cwnmax = 0.f;
  if (unfused == 2) {
    std::vector<float> a((size_t)CM * CK), w((size_t)CN * CK), bb(CN), wn((CN + 7) / 8 * 8, 0.f);
    for (auto& v : a) v = ((rand() & 0xFFFF) / 32768.f - 1.f) * 0.9f;
    for (auto& v : w) v = ((rand() & 0xFFFF) / 32768.f - 1.f) * 0.6f;
    for (auto& v : bb) v = ((rand() & 0xFFFF) / 32768.f - 1.f) * 0.1f;
    std::vector<uint16_t> w16(w.size());
    for (size_t i = 0; i < w.size(); i++) w16[i] = f32_to_f16_host(w[i]);
    for (int c = 0; c < CN; c++) {
      double ss = 0;
      for (int k = 0; k < CK; k++) ss += (double)w[(size_t)c * CK + k] * w[(size_t)c * CK + k];
      wn[c] = (float)(std::sqrt(ss) * 1.000001);
      cwnmax = std::max(cwnmax, wn[c]);
    }
    hipMalloc(&cA, a.size() * 4); hipMalloc(&cW, w.size() * 4); hipMalloc(&cbias, (size_t)cldl * 4); hipMalloc(&canorm, CM * 4);
    hipMalloc(&cwnorm, wn.size() * 4); hipMalloc(&cA16, a.size() * 2); hipMalloc(&cW16, w.size() * 2);
    hipMalloc(&cL, (size_t)CM * cldl * 2);
    hipMemcpy(cA, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(cW, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemset(cbias, 0, (size_t)cldl * 4);
    hipMemcpy(cbias, bb.data(), bb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(cwnorm, wn.data(), wn.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(cW16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice);
    launch_rows_to_bf16(cA, cA16, canorm, CM, CK, s1, 1);
    ConvGemm cq{};
    cq.A = (const float*)cA16; cq.Wt = (const float*)cW16; cq.bias = cbias; cq.C = (float*)cL; cq.M = CM; cq.N = CN; cq.K = CK; cq.ldc = cldl;
    cq.Hin = cq.Win = cq.Hout = cq.Wout = 1; cq.Cin = CK; cq.ks = 1; cq.pad = 0; cq.act = 0;
    launch_conv_gemm_bf16(cq, 2, s1);        // approximate logits, fp16, once
    hipStreamSynchronize(s1);
  }
  uint16_t* w3 = nullptr;
  float us3 = 1.f;
  auto victim = [&](float* Vout) {
    if (unfused == 2) {
      launch_cls_select(cL, cA, cW, cbias, canorm, cwnorm, (int32_t*)Vout, CM, CN, CK, cldl, s1, 1, cwnmax);
    } else if (unfused) {
      launch_conv0_pool(in, w9, bias, p0, nullptr, B, W, 1, s1);
      launch_wino4_in(p0, Vout, B, 16, W1, 64, s1, vs);
    } else {
      launch_conv0_win4(in, w9, bias, Vout, vs, nullptr, B, W, 1, s1, w3, us3);   // the product form: conv0 on the fp16 MFMA
    }
  };
  {
    std::vector<float> w16f((size_t)64 * 16, 0.f);
    for (int o = 0; o < 64; o++)
      for (int t = 0; t < 9; t++) w16f[(size_t)o * 16 + t] = hw[(size_t)t * 64 + o];
    std::vector<uint16_t> w3h(w16f.size() * 2);
    us3 = to_f16x3_host(w16f.data(), w16f.size(), w3h.data()) * (1.f / 16384.f);
    hipMalloc(&w3, w3h.size() * 2);
    hipMemcpy(w3, w3h.data(), w3h.size() * 2, hipMemcpyHostToDevice);
  }
  hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w9, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice);
  hipMemset(V, 0, vbytes); hipMemset(Vref, 0, vbytes); hipMemset(cnt, 0, (size_t)iters * 4);
  // reference: alone, twice (must agree with itself)
  victim(Vref);
  hipStreamSynchronize(s1);
  victim(V);
  hipLaunchKernelGGL(cmp_kernel, dim3(64), dim3(256), 0, s1, (const uint32_t*)V, (const uint32_t*)Vref, (long)(vbytes / 4), cnt);
  hipStreamSynchronize(s1);
  unsigned c0 = 0;
  hipMemcpy(&c0, cnt, 4, hipMemcpyDeviceToHost);
  printf("alone: %u differing words\n", c0);
  hipMemset(cnt, 0, 4);
  // aggressor operands: the classifier's filter GEMM of one crop
  const int M = 120, N = 5530, K = 512, ldl = (N + 7) & ~7;
  uint16_t *A16, *W16, *L;
  float* cb;
  hipMalloc(&A16, (size_t)M * K * 4); hipMalloc(&W16, (size_t)N * K * 4); hipMalloc(&L, (size_t)M * ldl * 4);
  hipMemset(A16, 0, (size_t)M * K * 4); hipMemset(W16, 0, (size_t)N * K * 4); hipMalloc(&cb, (size_t)ldl * 4);
  {
    std::vector<uint16_t> t((size_t)N * K);
    for (auto& v : t) { const _Float16 x = (_Float16)(((rand() & 0xFFFF) / 32768.f - 1.f) * 0.05f); v = __builtin_bit_cast(uint16_t, x); }
    hipMemcpy(W16, t.data(), t.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(A16, t.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemset(cb, 0, (size_t)ldl * 4);
  }
  ConvGemm q{};
  q.A = (const float*)A16; q.Wt = (const float*)W16; q.bias = cb; q.C = (float*)L; q.M = M; q.N = N; q.K = K; q.ldc = ldl;
  q.Hin = q.Win = q.Hout = q.Wout = 1; q.Cin = K; q.ks = 1; q.pad = 0; q.act = 0;
  auto aggressor = [&]() {
    switch (aggr) {
      case 1: launch_conv_gemm_bf16(q, 2, s2); break;
      case 2: hipLaunchKernelGGL(lds_spin_kernel<36864>, dim3(88), dim3(256), 0, s2, out, 8); break;
      case 3: hipLaunchKernelGGL(valu_spin_kernel, dim3(88), dim3(256), 0, s2, out, 40); break;
      case 4: hipLaunchKernelGGL(lds_spin_kernel<2048>, dim3(512), dim3(256), 0, s2, out, 30); break;
      case 5: launch_conv_gemm_bf16(q, 1, s2); break;
      case 6: launch_conv_gemm_bf16(q, 0, s2); break;
      case 7: launch_conv_gemm(q, s2); break;
      case 8: hipLaunchKernelGGL(mfma_spin_kernel<0>, dim3(88), dim3(256), 0, s2, out, 40); break;
      case 9: hipLaunchKernelGGL(mfma_spin_kernel<1>, dim3(88), dim3(256), 0, s2, out, 40); break;
      case 10: hipLaunchKernelGGL(mfma_spin_kernel<2>, dim3(88), dim3(256), 0, s2, out, 40); break;
      case 11: hipLaunchKernelGGL(mfma_spin_kernel<0>, dim3(1024), dim3(256), 0, s2, out, 40); break;
      case 12: hipLaunchKernelGGL(lds_spin_kernel<36864>, dim3(1024), dim3(256), 0, s2, out, 8); break;
      case 13: hipLaunchKernelGGL(mfma_spin_kernel<1>, dim3(1024), dim3(256), 0, s2, out, 40); break;
      case 14: hipLaunchKernelGGL(mfma_spin_kernel<2>, dim3(1024), dim3(256), 0, s2, out, 40); break;
      case 15: hipLaunchKernelGGL(valu_spin_kernel, dim3(1024), dim3(256), 0, s2, out, 40); break;
      default: break;
    }
  };
  for (int it = 0; it < iters; it++) {
    aggressor();
    if (it % 3 == 1) aggressor();
    victim(V);
    hipLaunchKernelGGL(cmp_kernel, dim3(64), dim3(256), 0, s1, (const uint32_t*)V, (const uint32_t*)Vref, (long)(vbytes / 4), cnt + it);
    if (it % 5 == 2) aggressor();
    if (it % 32 == 31) { hipStreamSynchronize(s1); hipStreamSynchronize(s2); }
    if (getenv("CORUN_CAPTURE")) {
      hipStreamSynchronize(s1);
      unsigned c = 0;
      hipMemcpy(&c, cnt + it, 4, hipMemcpyDeviceToHost);
      if (c) {
        std::vector<uint32_t> hv(vbytes / 4), hr(vbytes / 4);
        std::vector<float> hs(tp);
        hipMemcpy(hv.data(), V, vbytes, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), Vref, vbytes, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), vs, tp * 4, hipMemcpyDeviceToHost);
        FILE* f = fopen(getenv("CORUN_CAPTURE"), "wb");
        long hdr[4] = {tp, nt, W, it};
        fwrite(hdr, 8, 4, f); fwrite(hv.data(), 4, hv.size(), f); fwrite(hr.data(), 4, hr.size(), f); fwrite(hs.data(), 4, hs.size(), f);
        fwrite(hin.data(), 4, hin.size(), f);
        fclose(f);
        printf("captured iteration %d: %u words differ\n", it, c);
        hipStreamSynchronize(s2);
        return 0;
      }
    }
  }
  hipStreamSynchronize(s1); hipStreamSynchronize(s2);
  std::vector<unsigned> hc(iters);
  hipMemcpy(hc.data(), cnt, (size_t)iters * 4, hipMemcpyDeviceToHost);
  long badit = 0, words = 0;
  for (unsigned c : hc) { badit += c != 0; words += c; }
  printf("aggressor %d, W %d (%ld tiles): %ld of %d iterations differ (%ld words)\n", aggr, W, nt, badit, iters, words);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("hip error: %s\n", hipGetErrorString(e));
  return badit != 0;
}
