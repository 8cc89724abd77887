// Host side of the C ABI declared in include/crnn_hip.h: weight repacking,
// workspace management, Pillow-exact resize coefficients, kernel sequencing.
// There is no CPU compute fallback here: without a HIP device every compute
// entry point fails with CRNN_ERR_HIP.
#include "../../include/crnn_hip.h"
#include "crnn_kernels.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace crnn;

namespace {

const int NM[7] = {64, 128, 256, 256, 512, 512, 512};
const int KS[7] = {3, 3, 3, 3, 3, 3, 2};
const bool HAS_BN[7] = {false, false, true, false, true, false, true};

struct Geo {  // activation geometry for n lines of padded width W
  int n, W, W1, W2, W3, W4, T;
  explicit Geo(int n_ = 0, int W_ = 0) : n(n_), W(W_) {
    W1 = W / 2; W2 = W1 / 2; W3 = W2 + 1; W4 = W3 + 1; T = W3;
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// Workspace of crnn_forward_device_noalloc (the entry point that must not allocate): a fixed array of the code
// object.  In host code the array's NAME is only a stand-in for the symbol, not a device address: the address is
// asked for once per device at set-up (ensure_device, with that device current; every GPU has its own copy at its
// own address) and only that pointer is ever handed to kernels, memsets or copies.
static __device__ __align__(256) unsigned char scratch[CRNN_NOALLOC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
// Hidden-split recurrence launches are chained per device and process: each waits for the previous one's event, so that two
// of them -- of different lanes or of different handles -- are never both partly resident and spinning for workgroups that
// cannot be placed (that would only end by their time-outs).  The event belongs to the process that created it (fork).
struct SplitChain {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  pid_t pid = 0;
};
SplitChain g_split[MAX_DEVICES];
void* g_scratch_ptr[MAX_DEVICES] = {};      // [device ordinal] -> device address of `scratch` on that GPU (this process)
pid_t g_scratch_pid[MAX_DEVICES] = {};
std::mutex g_scratch_mu;

constexpr int MAX_LANES = 8;
constexpr int MAXK = 48;

struct Lane {
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
  hipEvent_t kev[MAXK + 1] = {};
  const char* kname[MAXK] = {};
  double kflop[MAXK] = {};
  double kbytes[MAXK] = {};
  int nk = 0;        // launches recorded by the last run
  DevBuf arena;      // activations
  DevBuf pre;        // preprocessing scratch (sources, temp, coefficients, jobs)
  DevBuf io;         // per-call small buffers (outputs of the host path)
  DevBuf stage;      // PINNED HOST memory: packed sources on their way to the device
  std::mutex mu;     // held by whichever call is using this lane's stream and buffers
  bool split_used = false;   // the last run_network on this lane used the hidden-split recurrence
  DevBuf sync;       // time-out word of the hidden-split recurrence
  DevBuf page;       // resident page of this lane (crnn_set_page uses lane 0, crnn_predict_page any free lane)
  int page_h = 0, page_w = 0, page_ch = 0;
  int pblocks = 256;           // blocks of the persistent GEMM launches of the next network run (set_persistent_blocks)
  bool want_p0 = false;        // crnn_debug_stage(.., 0): keep the pooled conv0 output (no conv0 / input transform fusion)
  int want_stage = -1;         // crnn_debug_stage(.., 3 / 4 / 5): keep pool2's / conv4's / pool3's output (no chained transform kernel there)
  float* st[11] = {};          // where the stages of the last run live (crnn_debug_stage)
  size_t st_floats[11] = {};
};

}  // namespace

struct crnn_handle {
  crnn_config cfg{};
  std::mutex mu;
  bool loaded = false;

  // host copies of the prepared weights (uploaded lazily, re-uploaded after fork)
  std::vector<float> h_conv_w[7], h_conv_b[7];
  std::vector<float> h_wih[2], h_bih[2], h_whhp[2], h_emb_w[2], h_emb_b[2], h_lin_w, h_lin_b;
  float lut[256];
  // Winograd F(4x4,3x3) weights U[36][Cout][Cin] for the layers in wino4_mask (fp32 path; used for batches of at least
  // wino4_min_n lines: measured faster from one line up, so the default is 1; a layer outside the mask runs as direct
  // implicit GEMM)
  std::vector<float> h_wino4[7];
  unsigned wino4_mask = 0;
  // f16x3 (CRNN_F16X3, default on; fp32 precision only): the Winograd GEMMs of the F(4x4) / F(4,2) layers run on the fp16
  // MFMA with operands split into (hi, lo) fp16 pairs -- fp32-accurate products at 16/3 of the f32 MFMA rate
  // (crnn_kernels.hip).  The weights are then kept only in that form: h_wino4h[i] + their unscale factor.
  bool h3 = true;
  std::vector<uint16_t> h_wino4h[7];
  float wino4_us[7] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  uint16_t* d_wino4h[7] = {};
  // ... and the LSTM input projections and the first embedding (lin_gemm_h3_kernel): W in the split form + unscale factor
  std::vector<uint16_t> h_wih_h3[2], h_emb0_h3;
  float wih_us[2] = {1.f, 1.f}, emb0_us = 1.f;
  uint16_t* d_wih_h3[2] = {};
  uint16_t* d_emb0_h3 = nullptr;
  // ... and conv0 inside conv0_win4_h3_kernel (CRNN_CONV0_MFMA=0: its scalar fp32 form): [64][16] taps, nine used
  std::vector<uint16_t> h_conv0_h3;
  float conv0_us3 = 1.f;   // 1 / (2^14 s_weights): the kernel scales the pixels by 2^14
  uint16_t* d_conv0_h3 = nullptr;
  // ... and the recurrence (lstm_rec_h3_kernel; CRNN_REC_H3=0 keeps the f32 MFMA kernel)
  bool rec_h3 = true;
  std::vector<uint16_t> h_whh3[2];
  float whh3_unscale[2] = {1.f, 1.f};
  uint16_t* d_whh3[2] = {};
  int wino4_min_n = 1;
  // classifier arg-max through the certified bf16 pre-filter when the logits are not requested (fp32 path;
  // CRNN_CLS_FILTER=0: plain fp32 GEMM with the fused arg-max)
  bool cls_filter = true;
  bool cls_f16 = false;   // the filter's operands and approximate logits in fp16 instead of bf16 (LSTM head; set at load)
  std::vector<float> h_cls_wnorm;   // ||w_c|| of the classifier rows (rounded up)
  float cls_wnmax = 0.f;            // the largest of them
  bool wino6 = true;      // conv6 as 1-D Winograd F(4,2) (fp32 path; CRNN_WINO6=0: implicit GEMM)
  bool lin_gemm = true;   // LSTM input projections and classifier on lin_gemm_kernel (CRNN_LIN_GEMM=0: conv_gemm_kernel)
  // recurrence: batches of up to split_max lines (4 workgroups per (direction, 16 lines): 256 workgroups, one per CU,
  // at 512 lines -- all of them must be resident at once) use the hidden-split latency kernel in the synchronous host
  // paths (CRNN_LSTM_SPLIT=<max lines>, 0 = never); measured faster than the one-workgroup kernel up to 1024 lines
  int split_max = 512;
  // (at most one hidden-split kernel runs on a DEVICE at a time, whichever handle of this process launched it:
  // g_split below)
  std::atomic<int64_t> split_timeouts{0};   // batches repeated because the hidden-split recurrence gave up (crnn_split_timeouts)
  // Load-time channel equalisation (equalize_channels below; CRNN_EQUALIZE=0: off): the power-of-two factor that
  // output channel c of conv i (chan_scale[i]) / of the first embedding (chan_scale[7]) carries relative to the
  // reference's activation.  crnn_debug_stage divides it out again; nothing else sees it.
  bool equalize = true;
  std::vector<float> chan_scale[8];
  // bf16 copies (precision == CRNN_BF16): conv1..6, W_ih, embeddings / linear, packed W_hh
  std::vector<uint16_t> hb_conv_w[7], hb_wih[2], hb_whhp[2], hb_emb_w[2], hb_lin_w;

  // device state
  pid_t dev_pid = 0;
  bool dev_ready = false;
  bool weights_on_dev = false;
  float* d_conv_w[7] = {};
  float* d_conv_b[7] = {};
  float* d_wih[2] = {};
  float* d_bih[2] = {};
  float* d_whhp[2] = {};
  float* d_emb_w[2] = {};
  float* d_emb_b[2] = {};
  float* d_lin_w = nullptr;
  float* d_lin_b = nullptr;
  float* d_wino4[7] = {};
  float* d_cls_wnorm = nullptr;
  float* d_zero_bias = nullptr;   // 512 zeros: the Winograd GEMMs add the bias in the output transform
  uint16_t* db_conv_w[7] = {};
  uint16_t* db_wih[2] = {};
  uint16_t* db_whhp[2] = {};
  uint16_t* db_emb_w[2] = {};
  uint16_t* db_lin_w = nullptr;
  std::atomic<bool> kprof{false};   // (read by the timing queries without the handle mutex)
  void* scratch_ptr = nullptr;   // device address of the no-alloc workspace on cfg.device (set by ensure_device)

  // Lanes: independent (stream, workspace) pairs.  Successive crnn_forward_device calls go to
  // successive lanes, so the latency-bound LSTM recurrence of one batch (32 workgroups) runs
  // beside the conv stack of the next batch and kernel tails of one lane are filled by another.
  Lane lanes[MAX_LANES];
  int nlanes = 1;
  int next_lane = 0;
  int next_host_lane = 0;
  // lane of the most recent network run (timing / profiling queries; those are single-caller measurement aids:
  // they lock that lane while they read its events, so a concurrent run cannot re-record them mid-query)
  std::atomic<Lane*> last{&lanes[0]};
};

namespace {

// The lane a call works on and its error text are per calling thread: host-path calls of different
// threads run concurrently on different lanes of one handle (gunicorn.conf.py:24 runs two threads per
// worker over the one global CRNN object).  Locking order: handle mutex, then lane mutex.
thread_local Lane* tl_cur = nullptr;
thread_local std::string tl_err;

int fail(const crnn_handle* h, int code, const std::string& msg) {
  (void)h;
  tl_err = msg;
  return code;
}

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(h, CRNN_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
  } while (0)

int ensure_buf(crnn_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
    HIPCHK(h, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 4096;
  HIPCHK(h, hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}

// pinned host staging (page and crop uploads): pageable hipMemcpy2D ran at 0.5 GB/s for an 11.6 MB page
int ensure_host(crnn_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
    HIPCHK(h, hipHostFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 4096;
  HIPCHK(h, hipHostMalloc(&b.p, want, hipHostMallocDefault));
  b.cap = want;
  return 0;
}

int upload(crnn_handle* h, const std::vector<float>& src, float** dst) {
  if (src.empty()) { *dst = nullptr; return 0; }
  HIPCHK(h, hipMalloc((void**)dst, src.size() * sizeof(float)));
  HIPCHK(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int upload16(crnn_handle* h, const std::vector<uint16_t>& src, uint16_t** dst) {
  if (src.empty()) { *dst = nullptr; return 0; }
  HIPCHK(h, hipMalloc((void**)dst, src.size() * sizeof(uint16_t)));
  HIPCHK(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  return 0;
}

// Lazy per-process device initialisation (the model object is built before
// gunicorn forks its workers; a HIP context does not survive fork()).
int ensure_device(crnn_handle* h, bool need_weights = true) {
  if (need_weights && !h->loaded)
    return fail(h, CRNN_ERR_STATE, "weights not loaded (call crnn_load_weights first)");
  if (h->dev_pid != getpid()) {
    // first use in this process (or first use after a fork: the parent's pointers
    // belong to its context -- drop them, do not free)
    h->dev_ready = false;
    h->weights_on_dev = false;
    h->scratch_ptr = nullptr;
    for (Lane& L : h->lanes) {
      L.ev_valid = false;
      L.nk = 0;
      L.arena = DevBuf(); L.pre = DevBuf(); L.io = DevBuf(); L.stage = DevBuf();
      L.page = DevBuf(); L.page_h = L.page_w = L.page_ch = 0;
      L.sync = DevBuf();
    }
  }
  if (!h->dev_ready) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      return fail(h, CRNN_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (h->cfg.device < 0 || h->cfg.device >= ndev) return fail(h, CRNN_ERR_ARG, "device ordinal out of range");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->cfg.device < MAX_DEVICES) {   // once per device and process: where this GPU keeps its copy of `scratch`
      std::lock_guard<std::mutex> lk(g_scratch_mu);
      if (!g_scratch_ptr[h->cfg.device] || g_scratch_pid[h->cfg.device] != getpid()) {
        void* ptr = nullptr;
        HIPCHK(h, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[h->cfg.device] = ptr;
        g_scratch_pid[h->cfg.device] = getpid();
      }
      h->scratch_ptr = g_scratch_ptr[h->cfg.device];
    }
    for (int l = 0; l < h->nlanes; l++) {
      Lane& L = h->lanes[l];
      HIPCHK(h, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
      for (int i = 0; i < 5; i++) HIPCHK(h, hipEventCreate(&L.ev[i]));
      for (int i = 0; i <= MAXK; i++) HIPCHK(h, hipEventCreate(&L.kev[i]));
    }
    upload_norm_lut(h->lut, tl_cur->stream);
    HIPCHK(h, hipMalloc((void**)&h->d_zero_bias, 512 * sizeof(float)));
    HIPCHK(h, hipMemset(h->d_zero_bias, 0, 512 * sizeof(float)));
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
    // whatever the runtime allocates lazily -- a stream's queue at its first launch, a code object at the first launch of
    // one of its kernels -- is taken now, at set-up: crnn_forward_device_noalloc promises to take no device memory later
    for (int l = 0; l < h->nlanes; l++) {
      hipStream_t ls = h->lanes[l].stream;
      launch_warm_main(ls);
      launch_warm_bf16(ls);
      launch_warm_nopk(ls);
      HIPCHK(h, hipMemsetAsync(h->d_zero_bias, 0, 16, ls));
      HIPCHK(h, hipEventRecord(h->lanes[l].ev[0], ls));
      HIPCHK(h, hipStreamSynchronize(ls));
    }
    h->dev_pid = getpid();
    h->dev_ready = true;
  } else {
    HIPCHK(h, hipSetDevice(h->cfg.device));
  }
  if (need_weights && !h->weights_on_dev) {
    int rc;
    for (int i = 0; i < 7; i++) {
      if ((rc = upload(h, h->h_conv_w[i], &h->d_conv_w[i]))) return rc;
      if ((rc = upload(h, h->h_conv_b[i], &h->d_conv_b[i]))) return rc;
    }
    for (int l = 0; l < 2; l++) {
      if ((rc = upload(h, h->h_wih[l], &h->d_wih[l]))) return rc;
      if ((rc = upload(h, h->h_bih[l], &h->d_bih[l]))) return rc;
      if ((rc = upload(h, h->h_whhp[l], &h->d_whhp[l]))) return rc;
      if ((rc = upload(h, h->h_emb_w[l], &h->d_emb_w[l]))) return rc;
      if ((rc = upload(h, h->h_emb_b[l], &h->d_emb_b[l]))) return rc;
    }
    if ((rc = upload(h, h->h_lin_w, &h->d_lin_w))) return rc;
    if ((rc = upload(h, h->h_lin_b, &h->d_lin_b))) return rc;
    for (int i = 0; i < 7; i++)
      if ((rc = upload16(h, h->hb_conv_w[i], &h->db_conv_w[i]))) return rc;
    for (int i = 0; i < 7; i++)
      if ((rc = upload(h, h->h_wino4[i], &h->d_wino4[i]))) return rc;
    for (int i = 0; i < 7; i++)
      if ((rc = upload16(h, h->h_wino4h[i], &h->d_wino4h[i]))) return rc;
    for (int l = 0; l < 2; l++)
      if ((rc = upload16(h, h->h_wih_h3[l], &h->d_wih_h3[l]))) return rc;
    for (int l = 0; l < 2; l++)
      if ((rc = upload16(h, h->h_whh3[l], &h->d_whh3[l]))) return rc;
    if ((rc = upload16(h, h->h_emb0_h3, &h->d_emb0_h3))) return rc;
    if ((rc = upload16(h, h->h_conv0_h3, &h->d_conv0_h3))) return rc;
    if ((rc = upload(h, h->h_cls_wnorm, &h->d_cls_wnorm))) return rc;
    for (int l = 0; l < 2; l++) {
      if ((rc = upload16(h, h->hb_wih[l], &h->db_wih[l]))) return rc;
      if ((rc = upload16(h, h->hb_whhp[l], &h->db_whhp[l]))) return rc;
      if ((rc = upload16(h, h->hb_emb_w[l], &h->db_emb_w[l]))) return rc;
    }
    if ((rc = upload16(h, h->hb_lin_w, &h->db_lin_w))) return rc;
    h->weights_on_dev = true;
  }
  return 0;
}

// ---- Pillow Resample.c coefficient computation (BILINEAR, support 1.0) ------
// Same double-precision expressions as precompute_coeffs / normalize_coeffs_8bpc
// so that the 22-bit fixed-point taps are bit-identical to Pillow's.
// (no FMA contraction: Pillow's build evaluates these as separate double operations)
int resample_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
#pragma clang fp contract(off)
  float in0 = 0.0f, in1 = (float)in_size;
  double scale, filterscale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; xx++) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; x++) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      const double w = v < 1.0 ? 1.0 - v : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; x++) {
      if (ww != 0.0) k[x] /= ww;
      const double c = k[x];
      kk[(size_t)xx * ksize + x] = c < 0 ? (int32_t)(-0.5 + c * (1 << 22)) : (int32_t)(0.5 + c * (1 << 22));
    }
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

int resized_width(int sw, int sh, int img_h) {
#pragma clang fp contract(off)
  const double scale = (double)sh * 1.0 / (double)img_h;
  return (int)((double)sw / scale);
}

const float* find_tensor(const std::map<std::string, const crnn_tensor*>& m, const std::string& name,
                         std::initializer_list<int64_t> shape, std::string& err) {
  auto it = m.find(name);
  if (it == m.end()) { err = "missing tensor " + name; return nullptr; }
  const crnn_tensor* t = it->second;
  if (!t->data || t->ndim != (int)shape.size()) { err = "bad rank for " + name; return nullptr; }
  int i = 0;
  for (int64_t s : shape)
    if (t->shape[i++] != s) { err = "bad shape for " + name; return nullptr; }
  return t->data;
}

// activation workspace: carve the arena for geometry g
struct Plan {
  float *x, *p0, *c1, *p1, *c2, *c3, *p2, *c4, *c5, *p3, *seq, *xp, *rec0, *mid, *rec1, *logits;
  int32_t *wl;   // [8][n] per-layer width arrays
  int32_t *raw;  // [n*T]
  uint16_t* a16;  // [n*T][512] bf16 copy of the classifier's input (certified pre-filter)
  float* anorm;   // [n*T] its row norms
  float* pmax;   // [n*T][n-tiles of the classifier]
  float *wv, *wm; // Winograd V[16][tiles][Cin] and Mo[16][tiles][Cout] (largest enabled layer)
  float *vs, *vs2; // f16x3: per-tile unscale factors of the current layer's V (largest tile count); vs2: of the next
                   // layer's V while a chained transform kernel (wino4_out_in_h3_kernel) still reads this layer's
  float *a3, *rs; // f16x3 copy of a plain GEMM's A operand [n*T][512] and its per-32-row unscale factors
  int32_t* pidx;
  size_t bytes;
};

static bool use_wino4(const crnn_handle* h, int n) { return (h->wino4_mask || h->wino6) && n >= h->wino4_min_n; }

Plan make_plan(const crnn_handle* h, const Geo& g, char* base) {
  Plan p{};
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* r = (float*)(base + off);
    off += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  const size_t n = g.n;
  p.x = take(n * 32 * g.W);
  p.p0 = take(n * 16 * g.W1 * 64);
  p.c1 = take(n * 16 * g.W1 * 128);
  p.p1 = take(n * 8 * g.W2 * 128);
  p.c2 = take(n * 8 * g.W2 * 256);
  p.c3 = take(n * 8 * g.W2 * 256);
  p.p2 = take(n * 4 * g.W3 * 256);
  p.c4 = take(n * 4 * g.W3 * 512);
  p.c5 = take(n * 4 * g.W3 * 512);
  p.p3 = take(n * 2 * g.W4 * 512);
  p.seq = take(n * g.T * 512);
  p.xp = take(n * g.T * 2048);
  p.rec0 = take(n * g.T * 512);
  p.mid = take(n * g.T * 256);
  p.rec1 = take(n * g.T * 512);
  p.logits = take(n * g.T * (size_t)h->cfg.nclass);
  p.a16 = (uint16_t*)take(n * g.T * 256);   // 512 bf16 per row
  p.anorm = take(n * g.T);
  p.wl = (int32_t*)take(8 * n);
  p.raw = (int32_t*)take(n * g.T);
  const size_t ntn = 4 * (((size_t)h->cfg.nclass + 127) / 128);   // lin_gemm_kernel: one partial per 32 columns
  p.pmax = take(n * g.T * ntn);
  p.pidx = (int32_t*)take(n * g.T * ntn);
  {
    size_t vmax = 0, mmax = 0;
    if (h->wino6 && use_wino4(h, (int)n)) {
      const size_t tp = (size_t)wino4_pad((long)n * ((g.T + 3) / 4));
      vmax = std::max(vmax, 5 * tp * 1024);
      mmax = std::max(mmax, 5 * tp * 512);
    }
    if (use_wino4(h, (int)n)) {
      const int hh[6] = {0, 16, 8, 8, 4, 4}, ww[6] = {0, g.W1, g.W2, g.W2, g.W3, g.W3};
      for (int i = 1; i <= 5; i++)
        if ((h->wino4_mask >> i) & 1u) {
          const size_t tp = (size_t)wino4_pad(wino4_tiles((int)n, hh[i], ww[i]));
          vmax = std::max(vmax, 36 * tp * NM[i - 1]);
          mmax = std::max(mmax, 36 * tp * NM[i]);
        }
    }
    p.wv = take(vmax);
    p.wm = take(mmax);
    size_t tmax = 0;
    if (use_wino4(h, (int)n)) {
      const int hh[6] = {0, 16, 8, 8, 4, 4}, ww[6] = {0, g.W1, g.W2, g.W2, g.W3, g.W3};
      for (int i = 1; i <= 5; i++) tmax = std::max(tmax, (size_t)wino4_pad(wino4_tiles((int)n, hh[i], ww[i])));
      tmax = std::max(tmax, (size_t)wino4_pad((long)n * ((g.T + 3) / 4)));
    }
    p.vs = take(tmax);
    p.vs2 = take(tmax);
    p.a3 = take(h->h3 && h->cfg.lstm ? n * g.T * 512 : 0);
    p.rs = take(h->h3 && h->cfg.lstm ? (n * g.T + 127) / 128 * 4 : 0);   // one per 32 rows, whole 128-row units
  }
  p.bytes = off;
  return p;
}

// Per-launch profiling (crnn_set_profiling): an event after every launch; name, algorithmic
// FLOPs (2*MACs) and algorithmic bytes (compulsory reads + writes) of the launch it closes.
void kmark(crnn_handle* h, const char* name, double flop, double bytes) {
  if (!h->kprof || tl_cur->nk >= MAXK) return;
  tl_cur->kname[tl_cur->nk] = name; tl_cur->kflop[tl_cur->nk] = flop; tl_cur->kbytes[tl_cur->nk] = bytes;
  tl_cur->nk++;
  (void)hipEventRecord(tl_cur->kev[tl_cur->nk], tl_cur->stream);
}

// fp32 GEMMs with at least 256 units' worth of columns x rows... go through the persistent direct-to-LDS kernel
// (CRNN_LIN_GEMM=0: all through conv_gemm_kernel).  The arg-max partial layout depends on the kernel, so the
// choice for the classifier must be the same in gemm() and in the reduction that follows: lin_partials().
static bool use_lin_gemm(const crnn_handle* h, const ConvGemm& p) {
  // plain GEMMs from 256 columns (emb0: 284 units), the arg-max form from 512 (what the tests cover)
  return h->lin_gemm && p.N >= (p.pmax ? 512 : 256) && lin_gemm_ok(p);
}

// out_bf16 only matters on the bf16 path (which GEMM outputs feed another GEMM)
void gemm(crnn_handle* h, const char* name, const void* A, const float* Wt, const uint16_t* Wt16,
          const float* bias, void* C, int M, int N, int K, bool out_bf16, float* pmax = nullptr,
          int32_t* pidx = nullptr, const uint16_t* W3 = nullptr, float us3 = 1.f, float* a3 = nullptr, float* rs = nullptr,
          bool a3_ready = false) {
  const bool b16 = h->cfg.precision == CRNN_BF16;
  ConvGemm p{};
  if (!b16 && h->h3 && W3 && a3 && rs && !pmax && C && (K == 256 || K == 512) && N >= 256 && h->lin_gemm) {
    // f16x3: convert the rows of A (one scale per 32-row block), then the GEMM on the fp16 MFMA
    ConvGemm q{};
    q.A = a3; q.Wt = (const float*)W3; q.bias = bias; q.C = (float*)C; q.M = M; q.N = N; q.K = K; q.ldc = N;
    q.Hin = q.Win = q.Hout = q.Wout = 1; q.Cin = K; q.ks = 1; q.h3 = 1; q.rs = rs; q.us = us3;
    if (lin_gemm_ok(q) && (a3_ready || launch_rows_to_f16x3((const float*)A, a3, rs, M, K, tl_cur->stream))) {
      static const char* cname[3] = {"lstm0_ih_split", "emb0_split", "lstm1_ih_split"};
      if (!a3_ready) kmark(h, cname[name[0] == 'e' ? 1 : (name[4] == '0' ? 0 : 2)], 0, 8.0 * M * K);
      if (lin_gemm_pp_ok(M, N, K)) {   // whole 256-row units (18,176 rows at 256 lines): the ping-pong GEMM, 8 % quicker
        Wino4Gemm w{};
        w.V = a3; w.U = (const float*)W3; w.M = (float*)C; w.ntile_pad = M; w.N = N; w.Cin = K; w.ncomp = 1; w.h3 = 1;
        w.lin_bias = bias; w.lin_rs = rs; w.lin_us = us3;
        launch_lin_gemm_pp(w, tl_cur->stream);
        static const char* ppname[3] = {"lstm0_ih_pp", "emb0_pp", "lstm1_ih_pp"};   // (the launch names tell the kernel)
        name = ppname[name[0] == 'e' ? 1 : (name[4] == '0' ? 0 : 2)];
      } else
        launch_lin_gemm(q, tl_cur->stream);
      kmark(h, name, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
      return;
    }
  }
  p.A = (const float*)A; p.Wt = b16 ? (const float*)Wt16 : Wt; p.bias = bias; p.C = (float*)C; p.wvalid = nullptr;
  p.M = M; p.N = N; p.K = K; p.ldc = N;
  p.Hin = p.Win = p.Hout = p.Wout = 1; p.Cin = K; p.ks = 1; p.pad = 0; p.act = 0;
  p.pmax = pmax; p.pidx = pidx;
  if (b16) launch_conv_gemm_bf16(p, out_bf16, tl_cur->stream);
  else if (use_lin_gemm(h, p)) launch_lin_gemm(p, tl_cur->stream);
  else launch_conv_gemm(p, tl_cur->stream);
  const double es = b16 ? 2.0 : 4.0;
  kmark(h, name, 2.0 * M * N * K,
        es * ((double)M * K + (double)N * K) + (C ? ((b16 && out_bf16) ? 2.0 : 4.0) * M * N : 8.0 * M * ((N + 127) / 128)));
}

// conv1 takes the f16x3 F(4x4) path for a batch of n lines (the condition conv() applies): then conv0 + pool0 can write
// conv1's V directly (conv0_win4_h3_kernel)
static bool conv1_on_wino4_h3(const crnn_handle* h, int n) {
  return h->cfg.precision != CRNN_BF16 && h->h3 && ((h->wino4_mask >> 1) & 1u) && use_wino4(h, n) && h->d_wino4h[1] != nullptr;
}

// pool_out: where a following MaxPool 2x2/2 would write; the fused Winograd kernel can apply it
// itself (one tile == one pool window).  Returns true when it did, so the caller skips the pool.
// chain_src >= 0 (with vs_next): instead of the output transform, wino4_out_in_h3_kernel may write the NEXT layer's V and
// scales (0: this layer's map feeds it directly, 1: through pool2, 2: through pool3 into conv6's F(4,2)); returns 2 then,
// and the caller passes v_ready there.
static bool conv_on_wino4_h3(const crnn_handle* h, int layer, int n, int Cin) {
  return h->cfg.precision != CRNN_BF16 && h->h3 && layer >= 1 && layer <= 5 && ((h->wino4_mask >> layer) & 1u) &&
         use_wino4(h, n) && h->d_wino4h[layer] && NM[layer] % 128 == 0 && (Cin & (Cin - 1)) == 0 && Cin >= 64 && Cin <= 512;
}
int conv(crnn_handle* h, int layer, const void* A, void* C, const int32_t* wvalid, int n, int Hin, int Win,
         int Cin, int Hout, int Wout, float* wino_v = nullptr, float* wino_m = nullptr, float* pool_out = nullptr,
         float* pool21_out = nullptr, float* vs = nullptr, bool v_ready = false, int chain_src = -1,
         float* vs_next = nullptr, float* split_a3 = nullptr, float* split_rs = nullptr, bool split_only = false) {
  static const char* names[7] = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6"};
  const bool b16 = h->cfg.precision == CRNN_BF16;
  // (with f16x3 weights only the split form of U exists: a shape that form does not take falls through to the direct conv)
  const bool h3_ok = h->h3 && layer >= 1 && layer <= 5 && h->d_wino4h[layer] && vs && NM[layer] % 128 == 0 &&
                     (Cin & (Cin - 1)) == 0 && Cin >= 64 && Cin <= 512;
  if (!b16 && ((h->wino4_mask >> layer) & 1u) && wino_v && wino_m && use_wino4(h, n) && (h3_ok || h->d_wino4[layer])) {
    // Winograd F(4x4,3x3): input transform, 36 GEMMs (K = Cin) in one persistent launch, output transform
    static const char* wn4[3][7] = {{"", "conv1_win4", "conv2_win4", "conv3_win4", "conv4_win4", "conv5_win4", ""},
                                    {"", "conv1_f4", "conv2_f4", "conv3_f4", "conv4_f4", "conv5_f4", ""},
                                    {"", "conv1_wout4", "conv2_wout4", "conv3_wout4", "conv4_wout4", "conv5_wout4", ""}};
    const int Co = NM[layer];
    const long tiles = wino4_tiles(n, Hin, Win), tp = wino4_pad(tiles);
    hipStream_t s = tl_cur->stream;
    const bool h3 = h3_ok;
    if (!v_ready) {   // (conv1: conv0_win4_h3_kernel may have written V and vs already)
      launch_wino4_in((const float*)A, wino_v, n, Hin, Win, Cin, s, h3 ? vs : nullptr);
      kmark(h, wn4[0][layer], 0, 4.0 * ((double)n * Hin * Win * Cin + 36.0 * tiles * Cin));
    }
    static const bool fuse21 = !(getenv("CRNN_WINO4_POOL21") && atoi(getenv("CRNN_WINO4_POOL21")) == 0);
    // the pooled form gives one thread a segment of a tile row: below ~128 lines there are too few of them and the
    // separate kernels are quicker (n = 1: 0.048 vs 0.02 ms per layer)
    if (!fuse21 || n < 128) pool21_out = nullptr;
    // GEMMs + output transform in one kernel (no product tensor): CRNN_WINO4_FUSE = bit mask of layers.  Its 64 x 64
    // units draw twice the operand bytes per FLOP from L2 of the 128 x 128 GEMM, so it pays where the product tensor
    // dominates: conv1 0.315 -> 0.20 ms, conv2 0.221 -> 0.217 (and 0.68 GB less HBM traffic), conv4 0.28 -> 0.37
    static const unsigned fuse_mask = getenv("CRNN_WINO4_FUSE") ? (unsigned)strtoul(getenv("CRNN_WINO4_FUSE"), nullptr, 0) : 0x6u;
    if (h3 && ((fuse_mask >> layer) & 1u) && Co % 64 == 0 && !pool21_out) {
      static const char* wf4[7] = {"", "conv1_fused4", "conv2_fused4", "conv3_fused4", "conv4_fused4", "conv5_fused4", ""};
      Wino4Fused f{};
      f.V = wino_v; f.U = (const float*)h->d_wino4h[layer]; f.vs = vs; f.bias = h->d_conv_b[layer];
      f.y = pool_out ? pool_out : (float*)C; f.wvalid = wvalid; f.us = h->wino4_us[layer];
      f.B = n; f.H = Hin; f.W = Win; f.Cin = Cin; f.N = Co; f.act = h->cfg.leaky_relu ? 2 : 1; f.pool = pool_out ? 1 : 0;
      f.ntile_pad = (int)tp; f.ntile = (int)tiles;
      launch_wino4_fused(f, s);
      const double px = pool_out ? (double)n * (Hin / 2) * (Win / 2) : (double)n * Hin * Win;
      kmark(h, wf4[layer], 2.0 * 36 * tiles * Co * Cin, 4.0 * (36.0 * ((double)tiles * Cin + (double)Co * Cin) + px * Co));
      return pool_out != nullptr ? 1 : 0;
    }
    Wino4Gemm w{};
    w.V = wino_v; w.U = h3 ? (const float*)h->d_wino4h[layer] : h->d_wino4[layer]; w.M = wino_m; w.ntile_pad = (int)tp;
    w.N = Co; w.Cin = Cin; w.ncomp = 36; w.h3 = h3;
    launch_wino4_gemm(w, s);
    static const char* wn4pp[7] = {"", "conv1_f4pp", "conv2_f4pp", "conv3_f4pp", "conv4_f4pp", "conv5_f4pp", ""};
    kmark(h, wino4_gemm_uses_pp(w) ? wn4pp[layer] : wn4[1][layer], 2.0 * 36 * tiles * Co * Cin, 4.0 * 36 * ((double)tiles * Cin + (double)Co * Cin + (double)tiles * Co));
    // CRNN_WINO4_CHAIN (bit mask of layers, default conv3 and conv4; read per call): this layer's output transform and
    // the next layer's input transform as one kernel -- the activation between them is never written
    // (as for the pooled output transform: one thread walks a row of tiles, too few of them below ~128 lines)
    const bool next_ok = chain_src == 2 ? (h->wino6 && h->d_wino4h[6] != nullptr && Co == 512)   // (conv()'s own test for conv6)
                                        : conv_on_wino4_h3(h, layer + 1, n, Co);
    if (chain_src >= 0 && h3 && vs_next && !pool_out && n >= 128 && (chain_src >= 1) == (pool21_out != nullptr) &&
        next_ok && wino4_out_in_ok(n, Hin, Win, Co, chain_src)) {
      const char* e = getenv("CRNN_WINO4_CHAIN");
      const unsigned mask = e ? (unsigned)strtoul(e, nullptr, 0) : 0x38u;
      if ((mask >> layer) & 1u) {
        launch_wino4_out_in(wino_m, h->d_conv_b[layer], vs, h->wino4_us[layer], wvalid, wino_v, vs_next, n, Hin, Win, Co,
                            h->cfg.leaky_relu ? 2 : 1, chain_src, s);
        const int Wn = chain_src ? Win + 1 : Win;
        const double vnext = chain_src == 2 ? 5.0 * (double)n * ((Wn - 1 + 3) / 4) * 2 * Co : 36.0 * (double)wino4_tiles(n, 4, Wn) * Co;
        kmark(h, layer == 3 ? "conv3_wout4_pool2_win4" : (layer == 4 ? "conv4_wout4_win4" : "conv5_wout4_pool3_win4"), 0,
              4.0 * (36.0 * (double)tiles * Co + vnext));
        return 2;
      }
    }
    float* dst = pool_out ? pool_out : (pool21_out ? pool21_out : (float*)C);
    launch_wino4_out(wino_m, h->d_conv_b[layer], dst, wvalid, n, Hin, Win, Co, h->cfg.leaky_relu ? 2 : 1,
                     pool_out ? 1 : (pool21_out ? 2 : 0), s, h3 ? vs : nullptr, h->wino4_us[layer]);
    const double outpx = pool_out ? (double)n * (Hin / 2) * (Win / 2)
                                  : (pool21_out ? (double)n * (Hin / 2) * (Win + 1) : (double)n * Hin * Win);
    kmark(h, pool21_out ? (layer == 3 ? "conv3_wout4_pool2" : "conv5_wout4_pool3") : wn4[2][layer], 0,
          4.0 * (36.0 * tiles * Co + outpx * Co));
    return pool_out != nullptr || pool21_out != nullptr ? 1 : 0;
  }
  const bool h3_ok6 = h->h3 && layer == 6 && h->d_wino4h[6] && vs && Cin == 512;
  if (!b16 && layer == 6 && h->wino6 && wino_v && wino_m && use_wino4(h, n) && (h3_ok6 || h->d_wino4[6])) {
    // conv6 as 1-D Winograd F(4,2) along the width: 5 GEMMs of K = 2 Cin instead of one of K = 4 Cin
    const int Co = NM[6];
    const long tiles = (long)n * ((Wout + 3) / 4), tp = wino4_pad(tiles);
    hipStream_t s = tl_cur->stream;
    const bool h3 = h3_ok6;
    if (!v_ready) {   // (wino4_out_in_h3_kernel<2> may have written V and vs from conv5's products)
      launch_wino42_in((const float*)A, wino_v, n, Win, Cin, s, h3 ? vs : nullptr);
      kmark(h, "conv6_win4", 0, 4.0 * ((double)n * 2 * Win * Cin + 5.0 * tiles * 2 * Cin));
    }
    Wino4Gemm w{};
    w.V = wino_v; w.U = h3 ? (const float*)h->d_wino4h[6] : h->d_wino4[6]; w.M = wino_m; w.ntile_pad = (int)tp; w.N = Co;
    w.Cin = 2 * Cin; w.ncomp = 5; w.h3 = h3;
    launch_wino4_gemm(w, s);
    kmark(h, wino4_gemm_uses_pp(w) ? "conv6_f4pp" : "conv6_f4", 2.0 * 5 * tiles * Co * 2 * Cin, 4.0 * 5 * ((double)tiles * 2 * Cin + (double)Co * 2 * Cin + (double)tiles * Co));
    if (h3 && split_a3 && split_rs && Co == 512) {
      // conv6's output transform writes the first LSTM projection's f16x3 operand itself (split_only: and no fp32 rows)
      launch_wino42_out_split(wino_m, h->d_conv_b[6], split_only ? nullptr : (float*)C, split_a3, split_rs, n, Wout,
                              h->cfg.leaky_relu ? 2 : 1, s, vs, h->wino4_us[6]);
      kmark(h, "conv6_wout4_split", 0, 4.0 * (5.0 * tiles * Co + (split_only ? 1.0 : 2.0) * n * Wout * Co));
      return 3;
    }
    launch_wino42_out(wino_m, h->d_conv_b[6], (float*)C, n, Wout, Co, h->cfg.leaky_relu ? 2 : 1, s, h3 ? vs : nullptr,
                      h->wino4_us[6]);
    kmark(h, "conv6_wout4", 0, 4.0 * (5.0 * tiles * Co + (double)n * Wout * Co));
    return 0;
  }
  ConvGemm p{};
  p.A = (const float*)A; p.Wt = b16 ? (const float*)h->db_conv_w[layer] : h->d_conv_w[layer];
  p.bias = h->d_conv_b[layer]; p.C = (float*)C; p.wvalid = wvalid;
  p.M = n * Hout * Wout; p.N = NM[layer]; p.K = KS[layer] * KS[layer] * Cin; p.ldc = NM[layer];
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hout = Hout; p.Wout = Wout;
  p.ks = KS[layer]; p.pad = KS[layer] == 3 ? 1 : 0;
  p.act = h->cfg.leaky_relu ? 2 : 1;
  if (b16) launch_conv_gemm_bf16(p, true, tl_cur->stream);
  else launch_conv_gemm(p, tl_cur->stream);
  const double es = b16 ? 2.0 : 4.0;
  kmark(h, names[layer], 2.0 * p.M * p.N * p.K,
        es * ((double)n * Hin * Win * Cin + (double)p.N * p.K + (double)p.M * p.N));
  return 0;
}

// Runs the network on the normalised input already in plan.x.  widths: host
// per-line valid widths or nullptr.  Outputs (device): plan.raw plus d_lab/d_len.
// allow_split: the caller synchronises afterwards and checks split_timed_out() (host paths only)
int run_network(crnn_handle* h, const Geo& g, const Plan& p, const int32_t* widths, int32_t* d_lab,
                int32_t* d_len, bool want_logits = false, bool allow_split = false, bool may_alloc = true) {
  const int n = g.n;
  const int act = h->cfg.leaky_relu ? 2 : 1;
  const int32_t *m_p0 = nullptr, *m_c1 = nullptr, *m_w2 = nullptr, *m_w3 = nullptr, *tlen = nullptr;
  if (widths) {
    std::vector<int32_t> wl((size_t)8 * n);
    for (int i = 0; i < n; i++) {
      const int w = widths[i], w1 = w / 2, w2 = w1 / 2, w3 = w2 + 1;
      wl[0 * n + i] = w1;            // valid pooled width after conv0+pool0
      wl[1 * n + i] = 2 * w2;        // conv1 output columns that reach pool1
      wl[2 * n + i] = w2;            // conv2, conv3
      wl[3 * n + i] = w3;            // conv4, conv5 ; also T_i
    }
    HIPCHK(h, hipMemcpyAsync(p.wl, wl.data(), sizeof(int32_t) * 4 * n, hipMemcpyHostToDevice, tl_cur->stream));
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));  // wl is a stack temporary
    m_p0 = p.wl; m_c1 = p.wl + n; m_w2 = p.wl + 2 * n; m_w3 = p.wl + 3 * n; tlen = m_w3;
  }
  hipStream_t s = tl_cur->stream;
  HIPCHK(h, hipEventRecord(tl_cur->ev[1], s));
  set_persistent_blocks(tl_cur->pblocks);
  tl_cur->pblocks = 256;
  const double dn = n;
  const bool b16 = h->cfg.precision == CRNN_BF16;
  const double es = b16 ? 2.0 : 4.0;
  // conv0 + pool0 fused with conv1's input transform (CRNN_CONV0_FUSED=0: separate kernels; crnn_debug_stage(.., 0)
  // asks for the pooled conv0 output itself and gets the separate kernels too)
  static const bool c0f = !(getenv("CRNN_CONV0_FUSED") && atoi(getenv("CRNN_CONV0_FUSED")) == 0);
  const bool fuse0 = c0f && !tl_cur->want_p0 && conv1_on_wino4_h3(h, n) && p.wv && p.wm && p.vs && g.W % 2 == 0;
  if (fuse0) {
    launch_conv0_win4(p.x, h->d_conv_w[0], h->d_conv_b[0], p.wv, p.vs, m_p0, n, g.W, act, s, h->d_conv0_h3, h->conv0_us3);
    kmark(h, "conv0_pool0_win4", 2.0 * dn * 32 * g.W * 64 * 9, 4.0 * dn * 32 * g.W + 4.0 * 36.0 * wino4_tiles(n, 16, g.W1) * 64);
  } else {
    if (b16) launch_conv0_pool_bf16(p.x, h->d_conv_w[0], h->d_conv_b[0], p.p0, m_p0, n, g.W, act, s);
    else launch_conv0_pool(p.x, h->d_conv_w[0], h->d_conv_b[0], p.p0, m_p0, n, g.W, act, s);
    kmark(h, "conv0_pool0", 2.0 * dn * 32 * g.W * 64 * 9, 4.0 * dn * 32 * g.W + es * dn * 16 * g.W1 * 64);
  }
  if (!conv(h, 1, p.p0, p.c1, m_c1, n, 16, g.W1, 64, 16, g.W1, p.wv, p.wm, p.p1, nullptr, p.vs, fuse0)) {
    if (b16) launch_pool_2x2s2_bf16(p.c1, p.p1, n, 16, g.W1, 128, s);
    else launch_pool_2x2s2(p.c1, p.p1, n, 16, g.W1, 128, s);
    kmark(h, "pool1", 0, es * (dn * 16 * g.W1 * 128 + dn * 8 * g.W2 * 128));
  }
  conv(h, 2, p.p1, p.c2, m_w2, n, 8, g.W2, 128, 8, g.W2, p.wv, p.wm, nullptr, nullptr, p.vs);
  // conv3 -> pool2 -> conv4 and conv4 -> conv5: output transform + next input transform as one kernel where conv() can
  // (the scales of the two layers alternate between vs and vs2; a debug request for the stage in between gets it written)
  float *vs_a = p.vs, *vs_b = p.vs2;
  const int st_want = tl_cur->want_stage;
  int r3 = conv(h, 3, p.c2, p.c3, m_w2, n, 8, g.W2, 256, 8, g.W2, p.wv, p.wm, nullptr, p.p2, vs_a, false,
                st_want == 3 ? -1 : 1, vs_b);
  if (r3 == 2) std::swap(vs_a, vs_b);
  else if (!r3) {
    if (b16) launch_pool_2x2s21p01_bf16(p.c3, p.p2, n, 8, g.W2, 256, s);
    else launch_pool_2x2s21p01(p.c3, p.p2, n, 8, g.W2, 256, s);
    kmark(h, "pool2", 0, es * (dn * 8 * g.W2 * 256 + dn * 4 * g.W3 * 256));
  }
  const int r4 = conv(h, 4, p.p2, p.c4, m_w3, n, 4, g.W3, 256, 4, g.W3, p.wv, p.wm, nullptr, nullptr, vs_a, r3 == 2,
                      st_want == 4 ? -1 : 0, vs_b);
  if (r4 == 2) std::swap(vs_a, vs_b);
  const int r5 = conv(h, 5, p.c4, p.c5, m_w3, n, 4, g.W3, 512, 4, g.W3, p.wv, p.wm, nullptr, p.p3, vs_a, r4 == 2,
                      st_want == 5 ? -1 : 2, vs_b);
  if (r5 == 2) std::swap(vs_a, vs_b);
  else if (!r5) {
    if (b16) launch_pool_2x2s21p01_bf16(p.c5, p.p3, n, 4, g.W3, 512, s);
    else launch_pool_2x2s21p01(p.c5, p.p3, n, 4, g.W3, 512, s);
    kmark(h, "pool3", 0, es * (dn * 4 * g.W3 * 512 + dn * 2 * g.W4 * 512));
  }
  // conv6's output transform can write the first LSTM projection's f16x3 operand itself (CRNN_CONV6_SPLIT=0: separate
  // kernels); the fp32 sequence tensor is then written only for a debug request (stage 6)
  bool seq_split = false;
  {
    const char* c6e = getenv("CRNN_CONV6_SPLIT");   // (read per call)
    const bool c6s = !(c6e && atoi(c6e) == 0);
    ConvGemm q{};
    q.M = n * g.T; q.N = 2048; q.K = 512; q.ks = 1;
    const bool lin_h3 = !b16 && h->h3 && h->cfg.lstm && h->d_wih_h3[0] && p.a3 && p.rs && h->lin_gemm && lin_gemm_ok(q);
    const int r6 = conv(h, 6, p.p3, p.seq, nullptr, n, 2, g.W4, 512, 1, g.T, p.wv, p.wm, nullptr, nullptr, vs_a, r5 == 2, -1,
                        nullptr, c6s && lin_h3 ? p.a3 : nullptr, p.rs, st_want != 6);
    seq_split = r6 == 3;
  }
  HIPCHK(h, hipEventRecord(tl_cur->ev[2], s));
  const int M = n * g.T;
  const double recflop = 2.0 * M * 2 * 1024 * 256;
  const double recbytes = 4.0 * M * 2048 + es * ((double)M * 512 + 2 * 1024 * 256);
  // asynchronous device path (several lanes in flight): the hidden-split form there too (round 3; CRNN_PIPE_SPLIT=0: the
  // one-workgroup kernel), with the one-workgroup kernel queued behind it and gated on the time-out word, since nobody on
  // the host looks at that word in this path.  Measured (bench.py, 256 lines per step): 104.8 k -> 109.3 k lines/s at 100
  // steps and 98-101 k -> 107 k at 20 steps, with TWO lanes instead of four (a batch alone takes 2.9 instead of 3.9 ms)
  static const bool pipe_split = !(getenv("CRNN_PIPE_SPLIT") && atoi(getenv("CRNN_PIPE_SPLIT")) == 0);
  // (may_alloc == false: crnn_forward_device_noalloc -- the split form needs its sync buffer and the chain's event)
  const bool gated = !allow_split && may_alloc && pipe_split && h->d_whh3[0] != nullptr;
  // (the split form exists for the f16x3 recurrence only: CRNN_F16X3=0 / CRNN_REC_H3=0 keep the one-workgroup f32 kernel)
  const bool split = (allow_split || gated) && !b16 && h->cfg.lstm && h->d_whh3[0] != nullptr && n <= h->split_max &&
                     (double)M * 2048.0 < 2.0e9;
  unsigned* d_sync = nullptr;
  if (split) {
    const bool fresh = tl_cur->sync.p == nullptr;
    int rc = ensure_buf(h, tl_cur->sync, 4096);
    if (rc) return rc;
    d_sync = (unsigned*)tl_cur->sync.p;                       // [0] the time-out word, [1] repeats done on the device (gated)
    HIPCHK(h, hipMemsetAsync(d_sync, 0, fresh ? 4096 : 4, s));
  }
  tl_cur->split_used = split;
  auto rec = [&](int l, float* out) {
    if (b16) launch_lstm_rec_bf16(p.xp, h->db_whhp[l], out, tlen, n, g.T, s);
    else if (split) {
      SplitChain& ch = g_split[h->cfg.device >= 0 && h->cfg.device < MAX_DEVICES ? h->cfg.device : 0];
      std::lock_guard<std::mutex> lk(ch.mu);
      if (ch.pid != getpid()) { ch.ev = nullptr; ch.pid = getpid(); }     // a forked child starts its own chain
      if (!ch.ev) (void)hipEventCreateWithFlags(&ch.ev, hipEventDisableTiming);
      if (ch.ev) (void)hipStreamWaitEvent(s, ch.ev, 0);
      // two row tiles (32 lines) per workgroup where the recurrence shares the chip with another batch's kernels (the
      // asynchronous path, from 128 lines): half the workgroups, steps a little longer (the kernel).  CRNN_SPLIT_NT=1 / 2
      // forces one form (read per call)
      const char* nte = getenv("CRNN_SPLIT_NT");
      const int nt = nte && atoi(nte) > 0 ? (atoi(nte) >= 2 ? 2 : 1) : (gated && n >= 128 ? 2 : 1);
      launch_lstm_rec_split_h3(p.xp, h->d_whh3[l], out, tlen, n, g.T, h->whh3_unscale[l], d_sync, s, nt);
      if (ch.ev) (void)hipEventRecord(ch.ev, s);
      if (gated) launch_lstm_rec_h3(p.xp, h->d_whh3[l], out, tlen, n, g.T, h->whh3_unscale[l], s, d_sync);
    }
    else if (h->d_whh3[l]) launch_lstm_rec_h3(p.xp, h->d_whh3[l], out, tlen, n, g.T, h->whh3_unscale[l], s);
    else launch_lstm_rec(p.xp, h->d_whhp[l], out, tlen, n, g.T, s);
    kmark(h, l ? "lstm1_rec" : "lstm0_rec", recflop, recbytes);
  };
  // classifier + arg-max.  Logits not requested (every product path): certified bf16 pre-filter + exact fp32
  // scoring of the candidates (crnn_kernels.hip); plan.logits is the filter's scratch.  Otherwise the fp32 GEMM
  // with the fused arg-max partials.
  bool filtered = false;
  auto classify = [&](const float* in, const float* Wf, const uint16_t* W16, const float* bias) {
    const int N = h->cfg.nclass;
    if (!b16 && h->cls_filter && !want_logits && N >= 512 && N <= 8192 && h->d_cls_wnorm && W16) {
      const int ldl = (N + 7) & ~7;   // bf16 rows of the approximate logits: 16-byte aligned
      launch_rows_to_bf16(in, p.a16, p.anorm, M, 512, s, h->cls_f16);
      kmark(h, "cls_to_bf16", 0, 6.0 * M * 512);
      ConvGemm q{};
      q.A = (const float*)p.a16; q.Wt = (const float*)W16; q.bias = bias; q.C = p.logits;
      q.M = M; q.N = N; q.K = 512; q.ldc = ldl;
      q.Hin = q.Win = q.Hout = q.Wout = 1; q.Cin = 512; q.ks = 1; q.pad = 0; q.act = 0;
      static const bool g16 = !(getenv("CRNN_GEMM16_DMA") && atoi(getenv("CRNN_GEMM16_DMA")) == 0);
      if (g16 && gemm16_dma_ok(q)) launch_gemm16_dma(q, h->cls_f16 ? 1 : 0, h->cls_f16 ? 2 : 1, s);   // 256 x 256 units, direct-to-LDS
      else launch_conv_gemm_bf16(q, h->cls_f16 ? 2 : 1, s);    // approximate logits stored as bf16 / fp16 in plan.logits
      // bf16 MFMA work: not counted with the f32 MFMA FLOPs (bench.py's executed-FLOP rate is against the f32 peak)
      kmark(h, "cls_filter_bf16", 0, 2.0 * ((double)M * 512 + (double)N * 512) + 2.0 * M * N);
      launch_cls_select((const uint16_t*)p.logits, in, Wf, bias, p.anorm, h->d_cls_wnorm, p.raw, M, N, 512, ldl, s, h->cls_f16, h->cls_wnmax);
      kmark(h, "cls_select", 0, 2.0 * M * N);
      return true;
    }
    gemm(h, "classifier", in, Wf, W16, bias, want_logits ? p.logits : nullptr, M, N, 512, false, p.pmax, p.pidx);
    return false;
  };
  if (h->cfg.lstm) {
    gemm(h, "lstm0_ih", p.seq, h->d_wih[0], h->db_wih[0], h->d_bih[0], p.xp, M, 2048, 512, false, nullptr, nullptr,
         h->d_wih_h3[0], h->wih_us[0], p.a3, p.rs, seq_split);
    rec(0, p.rec0);
    gemm(h, "emb0", p.rec0, h->d_emb_w[0], h->db_emb_w[0], h->d_emb_b[0], p.mid, M, 256, 512, true, nullptr, nullptr,
         h->d_emb0_h3, h->emb0_us, p.a3, p.rs);
    gemm(h, "lstm1_ih", p.mid, h->d_wih[1], h->db_wih[1], h->d_bih[1], p.xp, M, 2048, 256, false, nullptr, nullptr,
         h->d_wih_h3[1], h->wih_us[1], p.a3, p.rs);
    rec(1, p.rec1);
    filtered = classify(p.rec1, h->d_emb_w[1], h->db_emb_w[1], h->d_emb_b[1]);
  } else {
    filtered = classify(p.seq, h->d_lin_w, h->db_lin_w, h->d_lin_b);
  }
  HIPCHK(h, hipEventRecord(tl_cur->ev[3], s));
  // the classifier's epilogue left per-tile (max, index) partials: finish the arg-max over tiles
  // partials per row: one per 128 columns from conv_gemm_kernel, one per 32 from lin_gemm_kernel
  int ntn = (h->cfg.nclass + 127) / 128;
  {
    ConvGemm q{};
    q.M = M; q.N = h->cfg.nclass; q.K = 512; q.ks = 1; q.pmax = p.pmax;   // the same question gemm() asked
    if (!b16 && use_lin_gemm(h, q)) ntn *= 4;
  }
  if (!filtered) {
    launch_argmax_partials(p.pmax, p.pidx, p.raw, M, ntn, s);
    kmark(h, "argmax", 0, 8.0 * M * ntn + 4.0 * M);
  }
  launch_ctc(p.raw, tlen, d_lab, d_len, n, g.T, s);
  kmark(h, "ctc", 0, 4.0 * (2.0 * M + n));
  HIPCHK(h, hipEventRecord(tl_cur->ev[4], s));
  tl_cur->ev_valid = true;
  h->last.store(tl_cur, std::memory_order_release);
  HIPCHK(h, hipGetLastError());
  // remember where the stages live
  float* st[11] = {p.p0, p.p1, p.c2, p.p2, p.c4, p.p3, p.seq, p.mid, p.rec0, p.rec1, p.logits};
  size_t sz[11] = {(size_t)n * 16 * g.W1 * 64, (size_t)n * 8 * g.W2 * 128, (size_t)n * 8 * g.W2 * 256,
                   (size_t)n * 4 * g.W3 * 256, (size_t)n * 4 * g.W3 * 512, (size_t)n * 2 * g.W4 * 512,
                   (size_t)M * 512, (size_t)M * 256, (size_t)M * 512, (size_t)M * 512,
                   (size_t)M * h->cfg.nclass};
  for (int i = 0; i < 11; i++) { tl_cur->st[i] = st[i]; tl_cur->st_floats[i] = sz[i]; }
  return 0;
}

int prepare_plan(crnn_handle* h, const Geo& g, Plan& p) {
  // conv_gemm_kernel addresses its operands with 32-bit byte offsets (raw buffer loads)
  const double big = 4.0 * g.n * std::max(std::max(16.0 * g.W1 * 64, 8.0 * g.W2 * 256), 4.0 * g.W3 * 512);
  if (big >= 2147000000.0)
    return fail(h, CRNN_ERR_ARG, "batch too large (an activation tensor exceeds 2 GiB): split the batch");
  Plan probe = make_plan(h, g, nullptr);
  int rc = ensure_buf(h, tl_cur->arena, probe.bytes);
  if (rc) return rc;
  p = make_plan(h, g, (char*)tl_cur->arena.p);
  return 0;
}

// Gray conversion + Pillow-exact resize + normalise for a set of host images,
// into d_x [nb,32,Wp].  sel lists the image indices to process, dw their
// resized widths.
// With rect_off != nullptr the sources are rectangles of the lane's page already in HBM: rect_off[i]
// is the byte offset of rectangle i's first pixel, rows are page rows, nothing is uploaded but the
// coefficient tables.
int run_preprocess(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights, const int32_t* widths,
                   const int32_t* row_strides, const int32_t* channels, const std::vector<int>& sel,
                   const std::vector<int>& dw, int Wp, float* d_x, const int64_t* rect_off = nullptr) {
  const int nb = (int)sel.size();
  std::vector<ResizeJob> jobs(nb);
  std::vector<int32_t> coef;
  std::vector<int32_t> bh, kh, bv, kv;
  size_t src_bytes = 0, tmp_bytes = 0;
  int max_th = 1, max_dw = 1;
  for (int j = 0; j < nb; j++) {
    const int i = sel[j];
    ResizeJob& J = jobs[j];
    const int sh = heights[i], sw = widths[i], ch = channels[i];
    J.sh = sh; J.sw = sw; J.ch = ch; J.dw = dw[j];
    if (rect_off) {
      J.stride = tl_cur->page_w * tl_cur->page_ch;
      J.src_off = rect_off[i];
    } else {
      J.stride = sw * ch;  // repacked tightly on upload
      J.src_off = (int64_t)src_bytes;
      src_bytes += ((size_t)sh * J.stride + 15) / 16 * 16;
    }
    J.ksh = resample_coeffs(sw, J.dw, bh, kh);
    J.ksv = resample_coeffs(sh, 32, bv, kv);
    J.ybox_first = bv[0];
    const int ybox_last = bv[2 * 31] + bv[2 * 31 + 1];
    J.th = ybox_last - J.ybox_first;
    for (int y = 0; y < 32; y++) bv[2 * y] -= J.ybox_first;
    J.tmp_off = (int64_t)tmp_bytes;
    tmp_bytes += ((size_t)J.th * J.dw + 15) / 16 * 16;
    J.coef_off = (int64_t)coef.size();
    coef.insert(coef.end(), bh.begin(), bh.end());
    coef.insert(coef.end(), kh.begin(), kh.end());
    coef.insert(coef.end(), bv.begin(), bv.end());
    coef.insert(coef.end(), kv.begin(), kv.end());
    if (J.th > max_th) max_th = J.th;
    if (J.dw > max_dw) max_dw = J.dw;
  }
  // pack the sources (rows may be strided views of a page: ocr_context.py:307, vat_detect.py:318)
  int rc = ensure_host(h, tl_cur->stage, src_bytes + 16);
  if (rc) return rc;
  uint8_t* src = (uint8_t*)tl_cur->stage.p;
  for (int j = 0; j < nb && !rect_off; j++) {
    const int i = sel[j];
    const size_t rowb = (size_t)widths[i] * channels[i];
    for (int y = 0; y < heights[i]; y++)
      memcpy(src + jobs[j].src_off + y * rowb, imgs[i] + (size_t)y * row_strides[i], rowb);
  }
  const size_t a = (src_bytes + 255) / 256 * 256, b = (tmp_bytes + 255) / 256 * 256;
  const size_t c = (coef.size() * 4 + 255) / 256 * 256, d = (jobs.size() * sizeof(ResizeJob) + 255) / 256 * 256;
  rc = ensure_buf(h, tl_cur->pre, a + b + c + d);
  if (rc) return rc;
  uint8_t* d_src = (uint8_t*)tl_cur->pre.p;
  uint8_t* d_tmp = d_src + a;
  if (rect_off) d_src = (uint8_t*)tl_cur->page.p;
  int32_t* d_coef = (int32_t*)(d_tmp + b);
  ResizeJob* d_jobs = (ResizeJob*)((char*)d_coef + c);
  hipStream_t s = tl_cur->stream;
  if (!rect_off && src_bytes) HIPCHK(h, hipMemcpyAsync(d_src, src, src_bytes, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(ResizeJob), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipEventRecord(tl_cur->ev[0], s));
  tl_cur->nk = 0;
  if (h->kprof) HIPCHK(h, hipEventRecord(tl_cur->kev[0], s));
  double src_px = 0, tmp_px = 0;   // algorithmic bytes: sources read, intermediate written + read, f32 written
  for (const ResizeJob& J : jobs) { src_px += (double)J.sh * J.sw * J.ch; tmp_px += (double)J.th * J.dw; }
  int max_sw = 1;
  for (const ResizeJob& J : jobs) max_sw = std::max(max_sw, (int)J.sw);
  launch_resize_h(d_src, d_jobs, d_coef, d_tmp, nb, max_th, max_dw, max_sw, s);
  kmark(h, "resize_h", 0, src_px + tmp_px);
  launch_resize_v(d_src, d_jobs, d_coef, d_tmp, d_x, nb, Wp, s);
  kmark(h, "resize_v", 0, tmp_px + 4.0 * nb * 32 * Wp);
  HIPCHK(h, hipStreamSynchronize(s));  // host staging vectors go out of scope
  return 0;
}

int check_images(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights, const int32_t* widths,
                 const int32_t* row_strides, const int32_t* channels, int n) {
  if (!imgs || !heights || !widths || !row_strides || !channels) return fail(h, CRNN_ERR_ARG, "null image array");
  for (int i = 0; i < n; i++) {
    if (channels[i] != 1 && channels[i] != 3) return fail(h, CRNN_ERR_ARG, "channels must be 1 (gray) or 3 (BGR)");
    if (heights[i] < 0 || widths[i] < 0) return fail(h, CRNN_ERR_ARG, "negative image size");
    if (heights[i] == 0) return fail(h, CRNN_ERR_ZERO_H, "image height 0 (ZeroDivisionError in resizeNormalize)");
    if (resized_width(widths[i], heights[i], 32) <= 0)
      return fail(h, CRNN_ERR_ZERO_W, "resized width 0 (height and width must be > 0)");
    if (!imgs[i]) return fail(h, CRNN_ERR_ARG, "null image pointer");
    if (row_strides[i] < widths[i] * channels[i]) return fail(h, CRNN_ERR_ARG, "row stride smaller than a row");
  }
  return 0;
}

// crnn_debug_stage reports the reference's activations: stages 0..6 are conv0..conv6 outputs (channels innermost), stage 7
// the first embedding's; the equalisation factors of their channels (exact powers of two) are divided out on the host.
void unscale_stage(const crnn_handle* h, int stage, float* out, size_t count) {
  if (stage > 7) return;
  const std::vector<float>& cs = h->chan_scale[stage];
  const size_t C = cs.size();
  if (!C) return;
  bool any = false;
  for (float v : cs) any |= v != 1.f;
  if (!any) return;
  for (size_t i = 0; i < count; i++) out[i] /= cs[i % C];
}

// Load-time channel equalisation.  Between two layers joined only by ReLU / LeakyReLU and max-pooling, output channel c
// of the producer may be multiplied by any s_c > 0 if the consumer's weights for input channel c are divided by it: the
// network computes the same function (network_torch.py:53-78: conv -> [BN] -> ReLU -> [pool] -> conv; the first
// embedding -> the second LSTM's input projection is plain linear, :26-32).  With s_c a power of two the change is exact
// in binary floating point, so an fp32 evaluation of the rescaled network gives the same bits as before (no overflow or
// underflow at these magnitudes).  A checkpoint uses that freedom arbitrarily -- a BatchNorm gain of 1e-3 in one channel and
// 1e3 in its neighbour, made up for in the next layer's columns -- and that is what hurts the split-fp16 GEMM operands,
// whose scale is one power of two per weight tensor / per activation tile: an element 2^-20 below the block maximum keeps 19
// significant bits instead of 22.  So the freedom is spent here, once, on the host: s_c = sqrt(column range of the consumer
// / row range of the producer), rounded to a power of two, makes the two ranges of every hidden channel equal (Nagel et al.
// 2019, "cross-layer equalization", there for int8 inference); a few sweeps over the chain conv0 .. conv6 -> W_ih (layer 0)
// or the dense head, and emb0 -> W_ih (layer 1).  Channels that are dead on either side are left alone.
// h->chan_scale keeps the accumulated factors (crnn_debug_stage reports the reference's activations).
void equalize_channels(crnn_handle* h, std::vector<double>* fw, std::vector<double>* fb) {
  const int nh = 256;
  auto p2 = [](double x) { return std::exp2(std::nearbyint(std::log2(x))); };
  for (int sweep = 0; sweep < 16; sweep++) {
    bool changed = false;
    for (int i = 0; i < 8; i++) {
      // producer rows: conv i (i < 7) or emb0 (i == 7); consumer columns: conv i+1, W_ih[0] / linear (after conv6), W_ih[1]
      if (i == 7 && !h->cfg.lstm) continue;
      const int C = i < 7 ? NM[i] : nh;
      std::vector<double> r1(C, 0.0), r2(C, 0.0);
      const int cin = i == 0 ? 1 : (i < 7 ? NM[i - 1] : 2 * nh);
      const size_t per = i < 7 ? (size_t)cin * KS[i] * KS[i] : (size_t)2 * nh;
      for (int c = 0; c < C; c++)
        for (size_t j = 0; j < per; j++)
          r1[c] = std::max(r1[c], i < 7 ? std::fabs(fw[i][(size_t)c * per + j]) : (double)std::fabs(h->h_emb_w[0][(size_t)c * per + j]));
      std::vector<float>* lin = nullptr;      // a [rows][C] consumer
      if (i < 6) {
        const int co = NM[i + 1], kk = KS[i + 1] * KS[i + 1];
        for (int o = 0; o < co; o++)
          for (int c = 0; c < C; c++)
            for (int t = 0; t < kk; t++) r2[c] = std::max(r2[c], std::fabs(fw[i + 1][((size_t)o * C + c) * kk + t]));
      } else {
        lin = i == 7 ? &h->h_wih[1] : (h->cfg.lstm ? &h->h_wih[0] : &h->h_lin_w);
        const size_t rows = lin->size() / C;
        for (size_t o = 0; o < rows; o++)
          for (int c = 0; c < C; c++) r2[c] = std::max(r2[c], (double)std::fabs((*lin)[o * C + c]));
      }
      std::vector<double> sc(C, 1.0);
      for (int c = 0; c < C; c++) {
        if (!(r1[c] > 0.0) || !(r2[c] > 0.0) || !std::isfinite(r1[c]) || !std::isfinite(r2[c])) continue;
        double v = p2(std::sqrt(r2[c] / r1[c]));
        const double tot = v * (double)h->chan_scale[i][c];
        if (!(tot > 0x1p-60 && tot < 0x1p60)) v = 1.0;       // keep the accumulated factor in a safe range
        sc[c] = v;
        if (v != 1.0) changed = true;
      }
      for (int c = 0; c < C; c++) {
        if (sc[c] == 1.0) continue;
        h->chan_scale[i][c] = (float)((double)h->chan_scale[i][c] * sc[c]);
        if (i < 7) {
          for (size_t j = 0; j < per; j++) fw[i][(size_t)c * per + j] *= sc[c];
          fb[i][c] *= sc[c];
        } else {
          for (size_t j = 0; j < per; j++) h->h_emb_w[0][(size_t)c * per + j] = (float)((double)h->h_emb_w[0][(size_t)c * per + j] * sc[c]);
          h->h_emb_b[0][c] = (float)((double)h->h_emb_b[0][c] * sc[c]);
        }
      }
      if (i < 6) {
        const int co = NM[i + 1], kk = KS[i + 1] * KS[i + 1];
        for (int o = 0; o < co; o++)
          for (int c = 0; c < C; c++)
            if (sc[c] != 1.0)
              for (int t = 0; t < kk; t++) fw[i + 1][((size_t)o * C + c) * kk + t] /= sc[c];
      } else {
        const size_t rows = lin->size() / C;
        for (size_t o = 0; o < rows; o++)
          for (int c = 0; c < C; c++)
            if (sc[c] != 1.0) (*lin)[o * C + c] = (float)((double)(*lin)[o * C + c] / sc[c]);
      }
    }
    if (!changed) break;
  }
}

}  // namespace

extern "C" {

int crnn_create(const crnn_config* cfg, crnn_handle** out) {
  if (!cfg || !out) return CRNN_ERR_ARG;
  *out = nullptr;
  if (cfg->img_h != 32 || cfg->nc != 1 || cfg->nh != 256 || cfg->nclass < 2) return CRNN_ERR_ARG;
  if (cfg->precision != CRNN_FP32 && cfg->precision != CRNN_BF16) return CRNN_ERR_ARG;
  crnn_handle* h = new (std::nothrow) crnn_handle();
  if (!h) return CRNN_ERR_NOMEM;
  h->cfg = *cfg;
  h->nlanes = cfg->lanes <= 0 ? 1 : (cfg->lanes > MAX_LANES ? MAX_LANES : cfg->lanes);
  // The lanes are HIP streams, which the runtime multiplexes over GPU_MAX_HW_QUEUES hardware queues (default 4, read
  // when HIP initialises).  Four or more lanes on four queues are SLOWER than three (two lanes serialise on one queue:
  // 7.25 against 5.9 ms per 256-line step), so without at least 8 queues the handle keeps three lanes.
  // CRNN_EFFECTIVE_HW_QUEUES: what the runtime really uses when the variable was changed after HIP came up
  // (ocr_amd/crnn/_capi.py sets it in that case).
  {
    const char* e = getenv("CRNN_EFFECTIVE_HW_QUEUES");
    if (!e) e = getenv("GPU_MAX_HW_QUEUES");
    const int q = e ? atoi(e) : 4;
    if (q < 8 && h->nlanes > 3 && !getenv("CRNN_KEEP_LANES")) h->nlanes = 3;
  }
  // F(4x4) layers (bit i = conv i): default conv1..5; a layer outside the mask runs as direct implicit GEMM.  With the
  // textbook points 0, +-1, +-2 conv1's error was too large (its input has the widest dynamic range), with the points
  // the kernels use now it is not.  CRNN_WINOGRAD_F4=<mask> overrides (0 = off), CRNN_WINO4_MIN_N=<lines>
  h->wino4_mask = cfg->precision == CRNN_FP32 ? 0x3Eu : 0u;
  if (const char* e = getenv("CRNN_WINOGRAD_F4")) h->wino4_mask = cfg->precision == CRNN_FP32 ? ((unsigned)strtoul(e, nullptr, 0) & 0x3Eu) : 0u;
  if (const char* e = getenv("CRNN_WINO4_MIN_N")) h->wino4_min_n = atoi(e);
  if (const char* e = getenv("CRNN_LIN_GEMM")) h->lin_gemm = atoi(e) != 0;
  if (const char* e = getenv("CRNN_WINO6")) h->wino6 = atoi(e) != 0;
  if (const char* e = getenv("CRNN_F16X3")) h->h3 = atoi(e) != 0;
  if (const char* e = getenv("CRNN_REC_H3")) h->rec_h3 = atoi(e) != 0;
  if (const char* e = getenv("CRNN_EQUALIZE")) h->equalize = atoi(e) != 0;
  if (cfg->precision != CRNN_FP32) h->h3 = false;
  if (const char* e = getenv("CRNN_CLS_FILTER")) h->cls_filter = atoi(e) != 0;
  if (cfg->precision != CRNN_FP32) h->cls_filter = false;
  if (cfg->precision != CRNN_FP32) h->wino6 = false;
  if (const char* e = getenv("CRNN_LSTM_SPLIT")) h->split_max = atoi(e);
  for (int v = 0; v < 256; v++) h->lut[v] = (float)(((double)v / 255.0 - 0.5) / 0.5);
  *out = h;
  return 0;
}

int crnn_load_weights(crnn_handle* h, const crnn_tensor* tensors, int32_t n) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  if (!tensors || n <= 0) return fail(h, CRNN_ERR_ARG, "no tensors");
  if (h->weights_on_dev) return fail(h, CRNN_ERR_STATE, "weights already on the device; create a new handle");
  std::map<std::string, const crnn_tensor*> m;
  for (int i = 0; i < n; i++) {
    if (!tensors[i].name) return fail(h, CRNN_ERR_ARG, "tensor without a name");
    std::string nm = tensors[i].name;
    size_t pos;
    while ((pos = nm.find("module.")) != std::string::npos) nm.erase(pos, 7);  // network_torch.py:115
    m[nm] = &tensors[i];
  }
  std::string err;
  const int nh = 256, nclass = h->cfg.nclass;
  // ---- phase A: stage.  Conv weights with BatchNorm2d(eval) folded -- y = (x - mean) / sqrt(var + eps) * gamma + beta --
  // in float64, reference layout [co][cin][ky][kx] (fw) + bias (fb); LSTM / linear weights as they come.
  std::vector<double> fw[7], fb[7];
  for (int i = 0; i < 7; i++) {
    const int cin = i == 0 ? 1 : NM[i - 1], co = NM[i], k = KS[i];
    const std::string pc = "cnn.conv" + std::to_string(i) + ".";
    const float* w = find_tensor(m, pc + "weight", {co, cin, k, k}, err);
    const float* b = w ? find_tensor(m, pc + "bias", {co}, err) : nullptr;
    const float *g = nullptr, *bt = nullptr, *mu = nullptr, *var = nullptr;
    if (b && HAS_BN[i]) {
      const std::string pb = "cnn.batchnorm" + std::to_string(i) + ".";
      g = find_tensor(m, pb + "weight", {co}, err);
      bt = g ? find_tensor(m, pb + "bias", {co}, err) : nullptr;
      mu = bt ? find_tensor(m, pb + "running_mean", {co}, err) : nullptr;
      var = mu ? find_tensor(m, pb + "running_var", {co}, err) : nullptr;
      if (!var) b = nullptr;
    }
    if (!b) return fail(h, CRNN_ERR_ARG, err);
    const size_t per = (size_t)cin * k * k;
    fw[i].resize((size_t)co * per);
    fb[i].resize(co);
    for (int o = 0; o < co; o++) {
      double sc = 1.0, sh = 0.0;
      if (g) {
        sc = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
        sh = (double)bt[o] - (double)mu[o] * sc;
      }
      fb[i][o] = (double)b[o] * sc + sh;
      for (size_t j = 0; j < per; j++) fw[i][(size_t)o * per + j] = (double)w[(size_t)o * per + j] * sc;
    }
  }
  const float* whh_src[2][2] = {};
  if (h->cfg.lstm) {
    for (int l = 0; l < 2; l++) {
      const int nin = l == 0 ? 512 : nh, nout = l == 0 ? nh : nclass;
      const std::string pr = "rnn." + std::to_string(l) + ".rnn.";
      const float *wih[2], *bih[2], *bhh[2];
      for (int d = 0; d < 2; d++) {
        const std::string sfx = d ? "_reverse" : "";
        wih[d] = find_tensor(m, pr + "weight_ih_l0" + sfx, {4 * nh, nin}, err);
        whh_src[l][d] = wih[d] ? find_tensor(m, pr + "weight_hh_l0" + sfx, {4 * nh, nh}, err) : nullptr;
        bih[d] = whh_src[l][d] ? find_tensor(m, pr + "bias_ih_l0" + sfx, {4 * nh}, err) : nullptr;
        bhh[d] = bih[d] ? find_tensor(m, pr + "bias_hh_l0" + sfx, {4 * nh}, err) : nullptr;
        if (!bhh[d]) return fail(h, CRNN_ERR_ARG, err);
      }
      h->h_wih[l].resize((size_t)2 * 4 * nh * nin);
      h->h_bih[l].resize((size_t)2 * 4 * nh);
      for (int d = 0; d < 2; d++) {
        memcpy(h->h_wih[l].data() + (size_t)d * 4 * nh * nin, wih[d], sizeof(float) * 4 * nh * nin);
        for (int j = 0; j < 4 * nh; j++) h->h_bih[l][(size_t)d * 4 * nh + j] = bih[d][j] + bhh[d][j];
      }
      const std::string pe = "rnn." + std::to_string(l) + ".embedding.";
      const float* ew = find_tensor(m, pe + "weight", {nout, 2 * nh}, err);
      const float* eb = ew ? find_tensor(m, pe + "bias", {nout}, err) : nullptr;
      if (!eb) return fail(h, CRNN_ERR_ARG, err);
      h->h_emb_w[l].assign(ew, ew + (size_t)nout * 2 * nh);
      h->h_emb_b[l].assign(eb, eb + nout);
    }
  } else {
    const float* lw = find_tensor(m, "linear.weight", {nclass, 2 * nh}, err);
    const float* lb = lw ? find_tensor(m, "linear.bias", {nclass}, err) : nullptr;
    if (!lb) return fail(h, CRNN_ERR_ARG, err);
    h->h_lin_w.assign(lw, lw + (size_t)nclass * 2 * nh);
    h->h_lin_b.assign(lb, lb + nclass);
  }
  // ---- phase B: equalise the hidden channels (exact power-of-two reparametrisation; see equalize_channels)
  for (int i = 0; i < 7; i++) h->chan_scale[i].assign(NM[i], 1.f);
  h->chan_scale[7].assign(nh, 1.f);
  if (h->equalize) equalize_channels(h, fw, fb);
  // ---- phase C: the layouts the kernels read
  for (int i = 0; i < 7; i++) {
    const int cin = i == 0 ? 1 : NM[i - 1], co = NM[i], k = KS[i];
    h->h_conv_w[i].assign((size_t)co * k * k * cin, 0.f);
    h->h_conv_b[i].assign(co, 0.f);
    for (int o = 0; o < co; o++) {
      h->h_conv_b[i][o] = (float)fb[i][o];
      for (int c = 0; c < cin; c++)
        for (int ky = 0; ky < k; ky++)
          for (int kx = 0; kx < k; kx++) {
            const float v = (float)fw[i][(((size_t)o * cin + c) * k + ky) * k + kx];
            if (i == 0) {
              h->h_conv_w[0][(size_t)(ky * 3 + kx) * 64 + o] = v;  // [tap][co] for the direct kernel
            } else {
              // K order (channel chunk, ky, kx, channel in chunk): the taps of one chunk are
              // consecutive K-tiles, so the 3x3 neighbourhood re-reads of a chunk hit L1/L2
              // instead of going back to the fabric (chunk = one K-tile: 32 fp32 / 64 bf16).
              const int ch = 32;
              h->h_conv_w[i][(size_t)o * k * k * cin + ((size_t)(c / ch) * k * k + ky * k + kx) * ch + c % ch] = v;
              if (h->cfg.precision == CRNN_BF16) {
                const int cb = 64;
                if (h->hb_conv_w[i].empty()) h->hb_conv_w[i].assign((size_t)co * k * k * cin, 0);
                h->hb_conv_w[i][(size_t)o * k * k * cin + ((size_t)(c / cb) * k * k + ky * k + kx) * cb + c % cb] =
                    f32_to_bf16_host(v);
              }
            }
          }
    }
  }
  // F(4x4,3x3) weights (points 0, 1, -1, 2, -1/2, inf): U = G g G^T,
  // G = [1 0 0; -1/3 -1/3 -1/3; 1/3 -1/3 1/3; 1/15 2/15 4/15; -16/15 8/15 -4/15; 0 0 1]
  for (int i = 1; i <= 5; i++) {
    if (!((h->wino4_mask >> i) & 1u)) continue;
    const int cin = NM[i - 1], co = NM[i];
    const std::vector<double>& w = fw[i];
    static const double G[6][3] = {{1, 0, 0},          {-1.0 / 3, -1.0 / 3, -1.0 / 3}, {1.0 / 3, -1.0 / 3, 1.0 / 3},
                                   {1.0 / 15, 2.0 / 15, 4.0 / 15}, {-16.0 / 15, 8.0 / 15, -4.0 / 15}, {0, 0, 1}};
    h->h_wino4[i].assign((size_t)36 * co * cin, 0.f);
    for (int o = 0; o < co; o++) {
      for (int c = 0; c < cin; c++) {
        double gg[3][3], t[6][3];
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) gg[a][b] = w[(((size_t)o * cin + c) * 3 + a) * 3 + b];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 3; b++) t[a][b] = G[a][0] * gg[0][b] + G[a][1] * gg[1][b] + G[a][2] * gg[2][b];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 6; b++)
            h->h_wino4[i][((size_t)(a * 6 + b) * co + o) * cin + c] =
                (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
      }
    }
  }
  // conv6 as F(4,2) along the width: U6[b][o][ky * 512 + c] = sum_l G[b][l] w[o][c][ky][l]
  if (h->wino6) {
    const int cin = NM[5], co = NM[6];
    const std::vector<double>& w = fw[6];
    static const double G[5][2] = {{0.5, 0}, {-0.5, -0.5}, {-1.0 / 6, 1.0 / 6}, {1.0 / 6, 1.0 / 3}, {0, 1}};
    h->h_wino4[6].assign((size_t)5 * co * 2 * cin, 0.f);
    for (int o = 0; o < co; o++) {
      for (int c = 0; c < cin; c++)
        for (int ky = 0; ky < 2; ky++) {
          const double w0 = w[(((size_t)o * cin + c) * 2 + ky) * 2 + 0];
          const double w1 = w[(((size_t)o * cin + c) * 2 + ky) * 2 + 1];
          for (int b = 0; b < 5; b++)
            h->h_wino4[6][((size_t)b * co + o) * 2 * cin + (size_t)ky * cin + c] = (float)(G[b][0] * w0 + G[b][1] * w1);
        }
    }
  }
  if (h->h3)
    for (int i = 1; i <= 6; i++) {
      if (h->h_wino4[i].empty()) continue;
      h->h_wino4h[i].resize(h->h_wino4[i].size() * 2);
      h->wino4_us[i] = to_f16x3_host(h->h_wino4[i].data(), h->h_wino4[i].size(), h->h_wino4h[i].data());
      std::vector<float>().swap(h->h_wino4[i]);     // only the split form is used
    }
  h->h_conv0_h3.clear();
  {
    static const bool c0m = !(getenv("CRNN_CONV0_MFMA") && atoi(getenv("CRNN_CONV0_MFMA")) == 0);
    if (h->h3 && c0m) {
      std::vector<float> w16((size_t)64 * 16, 0.f);   // [channel][tap], taps 9 .. 15 zero (one K = 16 MFMA step)
      for (int o = 0; o < 64; o++)
        for (int t = 0; t < 9; t++) w16[(size_t)o * 16 + t] = h->h_conv_w[0][(size_t)t * 64 + o];
      h->h_conv0_h3.resize(w16.size() * 2);
      h->conv0_us3 = to_f16x3_host(w16.data(), w16.size(), h->h_conv0_h3.data()) * (1.f / 16384.f);
    }
  }
  if (h->cfg.lstm) {
    for (int l = 0; l < 2; l++) {
      h->h_whhp[l].resize((size_t)2 * 4 * nh * nh);
      pack_whh(whh_src[l][0], whh_src[l][1], h->h_whhp[l].data());
      if (h->h3 && h->rec_h3) {
        h->h_whh3[l].resize((size_t)2 * 4 * nh * nh * 2);
        h->whh3_unscale[l] = pack_whh_h3(whh_src[l][0], whh_src[l][1], h->h_whh3[l].data());
      }
    }
  }
  if (h->h3 && h->cfg.lstm) {
    for (int l = 0; l < 2; l++) {
      h->h_wih_h3[l].resize(h->h_wih[l].size() * 2);
      h->wih_us[l] = to_f16x3_host(h->h_wih[l].data(), h->h_wih[l].size(), h->h_wih_h3[l].data());
    }
    h->h_emb0_h3.resize(h->h_emb_w[0].size() * 2);
    h->emb0_us = to_f16x3_host(h->h_emb_w[0].data(), h->h_emb_w[0].size(), h->h_emb0_h3.data());
  }
  if (h->cls_filter) {
    const std::vector<float>& wsrc = h->cfg.lstm ? h->h_emb_w[1] : h->h_lin_w;
    std::vector<uint16_t>& wdst = h->cfg.lstm ? h->hb_emb_w[1] : h->hb_lin_w;
    const int kk = 2 * nh;
    wdst.resize(wsrc.size());
    // fp16 (8x tighter bound, so far fewer candidates to score) when the operands cannot leave its range: the LSTM head's
    // input is h (|h| < 1) and the weights are checked here; the dense head's input is unbounded: bf16.  CRNN_CLS_F16=0: bf16
    float wmax = 0.f;
    for (float v : wsrc) wmax = std::max(wmax, std::fabs(v));
    h->cls_f16 = h->cfg.lstm && wmax < 30000.f && !(getenv("CRNN_CLS_F16") && atoi(getenv("CRNN_CLS_F16")) == 0);
    for (size_t i = 0; i < wsrc.size(); i++) wdst[i] = h->cls_f16 ? f32_to_f16_host(wsrc[i]) : f32_to_bf16_host(wsrc[i]);
    h->h_cls_wnorm.assign(((size_t)nclass + 7) / 8 * 8, 0.f);   // padded: the select kernel loads eight at a time
    h->cls_wnmax = 0.f;
    for (int c = 0; c < nclass; c++) {
      double ss = 0;
      for (int k = 0; k < kk; k++) ss += (double)wsrc[(size_t)c * kk + k] * wsrc[(size_t)c * kk + k];
      h->h_cls_wnorm[c] = (float)(std::sqrt(ss) * 1.000001);
      h->cls_wnmax = std::max(h->cls_wnmax, h->h_cls_wnorm[c]);
    }
  }
  if (h->cfg.precision == CRNN_BF16) {
    auto cvt = [](const std::vector<float>& src, std::vector<uint16_t>& dst) {
      dst.resize(src.size());
      for (size_t i = 0; i < src.size(); i++) dst[i] = f32_to_bf16_host(src[i]);
    };
    if (h->cfg.lstm) {
      for (int l = 0; l < 2; l++) {
        cvt(h->h_wih[l], h->hb_wih[l]);
        cvt(h->h_emb_w[l], h->hb_emb_w[l]);
        const std::string pr = "rnn." + std::to_string(l) + ".rnn.";
        h->hb_whhp[l].resize((size_t)2 * 4 * nh * nh);
        pack_whh_bf16(m[pr + "weight_hh_l0"]->data, m[pr + "weight_hh_l0_reverse"]->data, h->hb_whhp[l].data());
      }
    } else {
      cvt(h->h_lin_w, h->hb_lin_w);
    }
  }
  h->loaded = true;
  return 0;
}

int crnn_resized_width(const crnn_handle* h, int32_t src_w, int32_t src_h) {
  if (src_h == 0) return fail(h, CRNN_ERR_ZERO_H, "image height 0 (ZeroDivisionError in resizeNormalize)");
  if (src_h < 0 || src_w < 0) return fail(h, CRNN_ERR_ARG, "negative image size");
  return resized_width(src_w, src_h, 32);
}

int crnn_out_len(int32_t w) { return w < 4 ? 0 : Geo(1, w).T; }

int crnn_preprocess(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights, const int32_t* widths,
                    const int32_t* row_strides, const int32_t* channels, int32_t n, int32_t wp, float* out_host,
                    int32_t* out_w) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  tl_cur = &h->lanes[0];
  std::lock_guard<std::mutex> lane_lk(h->lanes[0].mu);
  if (n <= 0 || !out_host || !out_w) return fail(h, CRNN_ERR_ARG, "bad arguments");
  int rc = check_images(h, imgs, heights, widths, row_strides, channels, n);
  if (rc) return rc;
  std::vector<int> sel(n), dw(n);
  for (int i = 0; i < n; i++) {
    sel[i] = i;
    dw[i] = resized_width(widths[i], heights[i], 32);
    out_w[i] = dw[i];
    if (dw[i] > wp) return fail(h, CRNN_ERR_ARG, "wp smaller than a resized width");
  }
  if ((rc = ensure_device(h, false))) return rc;
  const size_t bytes = (size_t)n * 32 * wp * sizeof(float);
  if ((rc = ensure_buf(h, tl_cur->arena, bytes))) return rc;
  if ((rc = run_preprocess(h, imgs, heights, widths, row_strides, channels, sel, dw, wp, (float*)tl_cur->arena.p)))
    return rc;
  HIPCHK(h, hipMemcpy(out_host, tl_cur->arena.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

// crnn_predict_batch / crnn_predict_rects after argument checks (lock held, tl_cur set)
static int predict_core(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights, const int32_t* widths,
                        const int32_t* row_strides, const int32_t* channels, const int64_t* rect_off, int32_t n,
                        int32_t pad_mode, int32_t tmax, int32_t* out_raw, int32_t* out_lab, int32_t* out_len,
                        int32_t* out_t, float* out_logits) {
  int rc;
  std::vector<int> sel, dw;
  int Wp = 0;
  for (int i = 0; i < n; i++) {
    const int w = resized_width(widths[i], heights[i], 32);
    out_len[i] = 0;
    if (out_t) out_t[i] = 0;
    if (pad_mode == CRNN_PAD_PER_LINE && w < 8) continue;  // network_torch.py:141-142 -> ''
    sel.push_back(i);
    dw.push_back(w);
    if (w > Wp) Wp = w;
  }
  if (sel.empty()) return 0;
  if (Wp < 4) return fail(h, CRNN_ERR_ARG, "padded width < 4: the conv stack has no output");
  const Geo g((int)sel.size(), Wp);
  if (tmax < g.T) return fail(h, CRNN_ERR_ARG, "tmax smaller than the number of timesteps");
  if ((rc = ensure_device(h))) return rc;
  Plan p;
  if ((rc = prepare_plan(h, g, p))) return rc;
  const size_t nT = (size_t)g.n * g.T;
  if ((rc = ensure_buf(h, tl_cur->io, (nT + g.n) * sizeof(int32_t) + 512))) return rc;
  int32_t* d_lab = (int32_t*)tl_cur->io.p;
  int32_t* d_len = d_lab + ((nT + 63) / 64) * 64;
  if ((rc = run_preprocess(h, imgs, heights, widths, row_strides, channels, sel, dw, Wp, p.x, rect_off))) return rc;
  std::vector<int32_t> wv(dw.begin(), dw.end());
  const bool uniform = pad_mode == CRNN_PAD_REFERENCE;
  if ((rc = run_network(h, g, p, uniform ? nullptr : wv.data(), d_lab, d_len, out_logits != nullptr, true))) return rc;
  if (tl_cur->split_used) {
    // the hidden-split recurrence gives up (bounded spins) if its four workgroups per group do not all become
    // resident in time, e.g. under heavy load from other lanes: repeat with the one-workgroup kernel then
    unsigned tmo = 0;
    HIPCHK(h, hipMemcpyAsync(&tmo, tl_cur->sync.p, 4, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
    if (tmo) {
      h->split_timeouts++;
      if ((rc = run_network(h, g, p, uniform ? nullptr : wv.data(), d_lab, d_len, out_logits != nullptr, false))) return rc;
    }
  }
  std::vector<int32_t> lab(nT), raw(nT), len(g.n);
  HIPCHK(h, hipMemcpyAsync(lab.data(), d_lab, nT * 4, hipMemcpyDeviceToHost, tl_cur->stream));
  HIPCHK(h, hipMemcpyAsync(raw.data(), p.raw, nT * 4, hipMemcpyDeviceToHost, tl_cur->stream));
  HIPCHK(h, hipMemcpyAsync(len.data(), d_len, (size_t)g.n * 4, hipMemcpyDeviceToHost, tl_cur->stream));
  HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
  std::vector<float> lg;
  if (out_logits) {
    lg.resize(nT * (size_t)h->cfg.nclass);
    HIPCHK(h, hipMemcpy(lg.data(), p.logits, lg.size() * 4, hipMemcpyDeviceToHost));
  }
  for (int j = 0; j < g.n; j++) {
    const int i = sel[j];
    const int Ti = uniform ? g.T : Geo(1, dw[j]).T;
    out_len[i] = len[j];
    if (out_t) out_t[i] = Ti;
    memcpy(out_lab + (size_t)i * tmax, lab.data() + (size_t)j * g.T, sizeof(int32_t) * len[j]);
    if (out_raw) memcpy(out_raw + (size_t)i * tmax, raw.data() + (size_t)j * g.T, sizeof(int32_t) * Ti);
    if (out_logits)
      memcpy(out_logits + (size_t)i * tmax * h->cfg.nclass, lg.data() + (size_t)j * g.T * h->cfg.nclass,
             sizeof(float) * (size_t)Ti * h->cfg.nclass);
  }
  return 0;
}

// Concurrent callers (threads of one worker) each take a lane of their own: the handle mutex is held only
// to bring the device up and to pick the lane, the lane mutex (returned locked) for the run itself.
static int pick_lane(crnn_handle* h, std::unique_lock<std::mutex>& lane_lk) {
  {
    std::lock_guard<std::mutex> lk(h->mu);
    tl_cur = &h->lanes[0];
    int rc = ensure_device(h);
    if (rc) return rc;
    int lane = -1;
    for (int i = 0; i < h->nlanes && lane < 0; i++) {
      std::unique_lock<std::mutex> t(h->lanes[i].mu, std::try_to_lock);
      if (t.owns_lock()) { lane = i; lane_lk = std::move(t); }
    }
    if (lane < 0) {   // all busy: queue behind one of them, round robin
      lane = h->next_host_lane;
      h->next_host_lane = (h->next_host_lane + 1) % h->nlanes;
    }
    tl_cur = &h->lanes[lane];
  }
  if (!lane_lk.owns_lock()) lane_lk = std::unique_lock<std::mutex>(tl_cur->mu);
  return 0;
}

int crnn_predict_batch(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights, const int32_t* widths,
                       const int32_t* row_strides, const int32_t* channels, int32_t n, int32_t pad_mode,
                       int32_t tmax, int32_t* out_raw, int32_t* out_lab, int32_t* out_len, int32_t* out_t,
                       float* out_logits) {
  if (!h) return CRNN_ERR_ARG;
  if (n <= 0 || !out_lab || !out_len) return fail(h, CRNN_ERR_ARG, "bad arguments");
  if (pad_mode != CRNN_PAD_PER_LINE && pad_mode != CRNN_PAD_REFERENCE) return fail(h, CRNN_ERR_ARG, "bad pad_mode");
  int rc = check_images(h, imgs, heights, widths, row_strides, channels, n);
  if (rc) return rc;
  std::unique_lock<std::mutex> lane_lk;
  if ((rc = pick_lane(h, lane_lk))) return rc;
  return predict_core(h, imgs, heights, widths, row_strides, channels, nullptr, n, pad_mode, tmax, out_raw, out_lab,
                      out_len, out_t, out_logits);
}

// page -> the current lane's page buffer (through its pinned stage)
static int upload_page(crnn_handle* h, const uint8_t* page, int32_t height, int32_t width, int32_t row_stride,
                       int32_t channels) {
  if (!page || height <= 0 || width <= 0) return fail(h, CRNN_ERR_ARG, "empty page");
  if (channels != 1 && channels != 3) return fail(h, CRNN_ERR_ARG, "channels must be 1 (gray) or 3 (BGR)");
  if (row_stride < width * channels) return fail(h, CRNN_ERR_ARG, "row stride smaller than a row");
  Lane& L = *tl_cur;
  const size_t rowb = (size_t)width * channels, bytes = rowb * height;
  HIPCHK(h, hipStreamSynchronize(L.stream));   // the lane may still be reading its previous page
  int rc;
  if ((rc = ensure_buf(h, L.page, bytes + 256))) return rc;
  if ((rc = ensure_host(h, L.stage, bytes))) return rc;
  uint8_t* st = (uint8_t*)L.stage.p;
  if ((size_t)row_stride == rowb) memcpy(st, page, bytes);
  else for (int y = 0; y < height; y++) memcpy(st + (size_t)y * rowb, page + (size_t)y * row_stride, rowb);
  HIPCHK(h, hipMemcpyAsync(L.page.p, st, bytes, hipMemcpyHostToDevice, L.stream));
  HIPCHK(h, hipStreamSynchronize(L.stream));   // the stage is reused by the next upload
  L.page_h = height; L.page_w = width; L.page_ch = channels;
  return 0;
}

// rectangles of the current lane's page through the network
static int rects_core(crnn_handle* h, const int32_t* rects, int32_t n, int32_t pad_mode, int32_t tmax,
                      int32_t* out_raw, int32_t* out_lab, int32_t* out_len, int32_t* out_t, float* out_logits) {
  const Lane& L = *tl_cur;
  if (!L.page.p || L.page_h <= 0) return fail(h, CRNN_ERR_STATE, "no page: call crnn_set_page first");
  // im0[y1:y2, x1:x2] (vat_detect.py:318): ends are clipped to the page like a numpy slice; negative
  // coordinates (which numpy would wrap) are refused
  std::vector<int32_t> hh(n), ww(n), st(n), cc(n);
  std::vector<int64_t> off(n);
  for (int i = 0; i < n; i++) {
    const int x1 = rects[4 * i], y1 = rects[4 * i + 1];
    const int x2 = std::min(rects[4 * i + 2], L.page_w), y2 = std::min(rects[4 * i + 3], L.page_h);
    if (x1 < 0 || y1 < 0 || rects[4 * i + 2] < 0 || rects[4 * i + 3] < 0)
      return fail(h, CRNN_ERR_ARG, "negative rectangle coordinate");
    hh[i] = std::max(0, y2 - y1); ww[i] = std::max(0, x2 - x1);
    st[i] = L.page_w * L.page_ch; cc[i] = L.page_ch;
    off[i] = ((int64_t)y1 * L.page_w + x1) * L.page_ch;
    if (hh[i] == 0) return fail(h, CRNN_ERR_ZERO_H, "rectangle height 0 (ZeroDivisionError in resizeNormalize)");
    if (resized_width(ww[i], hh[i], 32) <= 0)
      return fail(h, CRNN_ERR_ZERO_W, "resized width 0 (height and width must be > 0)");
  }
  return predict_core(h, nullptr, hh.data(), ww.data(), st.data(), cc.data(), off.data(), n, pad_mode, tmax, out_raw,
                      out_lab, out_len, out_t, out_logits);
}

int crnn_set_page(crnn_handle* h, const uint8_t* page, int32_t height, int32_t width, int32_t row_stride,
                  int32_t channels) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  tl_cur = &h->lanes[0];
  std::lock_guard<std::mutex> lane_lk(h->lanes[0].mu);
  int rc = ensure_device(h, false);
  if (rc) return rc;
  return upload_page(h, page, height, width, row_stride, channels);
}

int crnn_predict_rects(crnn_handle* h, const int32_t* rects, int32_t n, int32_t pad_mode, int32_t tmax,
                       int32_t* out_raw, int32_t* out_lab, int32_t* out_len, int32_t* out_t, float* out_logits) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  tl_cur = &h->lanes[0];
  std::lock_guard<std::mutex> lane_lk(h->lanes[0].mu);
  if (n <= 0 || !rects || !out_lab || !out_len) return fail(h, CRNN_ERR_ARG, "bad arguments");
  if (pad_mode != CRNN_PAD_PER_LINE && pad_mode != CRNN_PAD_REFERENCE) return fail(h, CRNN_ERR_ARG, "bad pad_mode");
  return rects_core(h, rects, n, pad_mode, tmax, out_raw, out_lab, out_len, out_t, out_logits);
}

int crnn_predict_page(crnn_handle* h, const uint8_t* page, int32_t height, int32_t width, int32_t row_stride,
                      int32_t channels, const int32_t* rects, int32_t n, int32_t pad_mode, int32_t tmax,
                      int32_t* out_raw, int32_t* out_lab, int32_t* out_len, int32_t* out_t, float* out_logits) {
  if (!h) return CRNN_ERR_ARG;
  if (n <= 0 || !rects || !out_lab || !out_len) return fail(h, CRNN_ERR_ARG, "bad arguments");
  if (pad_mode != CRNN_PAD_PER_LINE && pad_mode != CRNN_PAD_REFERENCE) return fail(h, CRNN_ERR_ARG, "bad pad_mode");
  int rc;
  std::unique_lock<std::mutex> lane_lk;
  if ((rc = pick_lane(h, lane_lk))) return rc;
  if ((rc = upload_page(h, page, height, width, row_stride, channels))) return rc;
  return rects_core(h, rects, n, pad_mode, tmax, out_raw, out_lab, out_len, out_t, out_logits);
}

int crnn_forward_device(crnn_handle* h, const uint8_t* d_gray, int32_t n, int32_t w, const int32_t* widths,
                        int32_t* d_raw, int32_t* d_lab, int32_t* d_len, int32_t sync) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  if (!d_gray || n <= 0 || w < 8 || !d_lab || !d_len) return fail(h, CRNN_ERR_ARG, "bad arguments");
  if (widths)
    for (int i = 0; i < n; i++)
      if (widths[i] < 8 || widths[i] > w) return fail(h, CRNN_ERR_ARG, "line width outside [8, w]");
  const int lane = h->next_lane;
  h->next_lane = (h->next_lane + 1) % h->nlanes;
  tl_cur = &h->lanes[lane];
  std::lock_guard<std::mutex> lane_lk(h->lanes[lane].mu);
  int rc = ensure_device(h);
  if (rc) return rc;
  const Geo g(n, w);
  Plan p;
  if ((rc = prepare_plan(h, g, p))) return rc;
  int32_t* d_w = nullptr;
  if (widths) {
    d_w = p.wl + 4 * (size_t)n;
    HIPCHK(h, hipMemcpyAsync(d_w, widths, sizeof(int32_t) * n, hipMemcpyHostToDevice, tl_cur->stream));
  }
  HIPCHK(h, hipEventRecord(tl_cur->ev[0], tl_cur->stream));
  tl_cur->nk = 0;
  if (h->kprof) HIPCHK(h, hipEventRecord(tl_cur->kev[0], tl_cur->stream));
  launch_normalize(d_gray, p.x, d_w, n, w, w, tl_cur->stream);
  kmark(h, "normalize", 0, (double)n * 32 * w * 5.0);
  // several lanes in flight (no sync, not the one-batch-at-a-time profiling mode): leave CUs to the other lanes' kernels
  static const int pipe_blocks = getenv("CRNN_PIPELINE_BLOCKS") ? atoi(getenv("CRNN_PIPELINE_BLOCKS")) : 192;
  tl_cur->pblocks = (!sync && !h->kprof && h->nlanes > 1) ? pipe_blocks : 256;
  // sync callers have nothing else in flight on this lane and wait for the result: they get the latency form of the
  // recurrence (four workgroups per group, W_hh resident; 0.45 instead of 0.78 ms per layer at 256 lines), with the same
  // give-up-and-repeat handling as the host paths
  if ((rc = run_network(h, g, p, widths, d_lab, d_len, false, sync != 0))) return rc;
  if (sync && tl_cur->split_used) {
    unsigned tmo = 0;
    HIPCHK(h, hipMemcpyAsync(&tmo, tl_cur->sync.p, 4, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
    if (tmo) {
      h->split_timeouts++;
      if ((rc = run_network(h, g, p, widths, d_lab, d_len, false, false))) return rc;
    }
  }
  if (d_raw)
    HIPCHK(h, hipMemcpyAsync(d_raw, p.raw, sizeof(int32_t) * (size_t)n * g.T, hipMemcpyDeviceToDevice, tl_cur->stream));
  if (sync) HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
  return lane;
}

int64_t crnn_noalloc_workspace_bytes(const crnn_handle* h, int32_t n, int32_t w) {
  if (!h || n <= 0 || w < 8) return CRNN_ERR_ARG;
  return (int64_t)make_plan(h, Geo(n, w), nullptr).bytes;
}

// crnn_forward_device on the fixed workspace: no hipMalloc / hipFree / hipHostMalloc, no stream or event creation.
// Everything this function touches on the device was created by an earlier ensure_device(); the plan is carved
// out of h->scratch_ptr (never out of the array's name) and the kernels get their pointers from the plan as usual.
int crnn_forward_device_noalloc(crnn_handle* h, const uint8_t* d_gray, int32_t n, int32_t w, const int32_t* widths,
                                int32_t* d_raw, int32_t* d_lab, int32_t* d_len, int32_t sync) {
  if (!h) return CRNN_ERR_ARG;
  // argument and state checks under the handle mutex; it is released before the call waits for lane 0, so that a
  // no-alloc caller queued behind a long host-path call on lane 0 does not hold up the handle's other entry points
  std::unique_lock<std::mutex> lk(h->mu);
  if (!d_gray || n <= 0 || w < 8 || !d_lab || !d_len) return fail(h, CRNN_ERR_ARG, "bad arguments");
  if (widths)
    for (int i = 0; i < n; i++)
      if (widths[i] < 8 || widths[i] > w) return fail(h, CRNN_ERR_ARG, "line width outside [8, w]");
  if (!h->dev_ready || h->dev_pid != getpid() || !h->weights_on_dev || !h->scratch_ptr)
    return fail(h, CRNN_ERR_STATE,
                "crnn_forward_device_noalloc: the device state of this handle does not exist yet in this process and "
                "this entry point may not create it (it allocates): call crnn_stream(h, 0) after crnn_load_weights, or "
                "any other compute entry point, once first");
  const Geo g(n, w);
  const double big = 4.0 * g.n * std::max(std::max(16.0 * g.W1 * 64, 8.0 * g.W2 * 256), 4.0 * g.W3 * 512);
  const size_t need = big < 2147000000.0 ? make_plan(h, g, nullptr).bytes : (size_t)-1;
  if (need > (size_t)CRNN_NOALLOC_SCRATCH_BYTES) {
    char msg[256];
    snprintf(msg, sizeof msg,
             "crnn_forward_device_noalloc: %d lines of width %d need a workspace of %.1f MiB, the fixed one holds %.0f MiB: "
             "use a smaller batch or crnn_forward_device", n, w, (double)need / 1048576.0,
             (double)CRNN_NOALLOC_SCRATCH_BYTES / 1048576.0);
    return fail(h, CRNN_ERR_ARG, msg);
  }
  lk.unlock();
  tl_cur = &h->lanes[0];                                   // every no-alloc call is ordered on lane 0's stream
  std::lock_guard<std::mutex> lane_lk(h->lanes[0].mu);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const Plan p = make_plan(h, g, (char*)h->scratch_ptr);
  hipStream_t s = tl_cur->stream;
  int32_t* d_w = nullptr;
  if (widths) {
    d_w = p.wl + 4 * (size_t)n;
    HIPCHK(h, hipMemcpyAsync(d_w, widths, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  }
  HIPCHK(h, hipEventRecord(tl_cur->ev[0], s));
  tl_cur->nk = 0;
  if (h->kprof) HIPCHK(h, hipEventRecord(tl_cur->kev[0], s));
  launch_normalize(d_gray, p.x, d_w, n, w, w, s);
  kmark(h, "normalize", 0, (double)n * 32 * w * 5.0);
  int rc;
  if ((rc = run_network(h, g, p, widths, d_lab, d_len, false, false, false))) return rc;
  if (d_raw) HIPCHK(h, hipMemcpyAsync(d_raw, p.raw, sizeof(int32_t) * (size_t)n * g.T, hipMemcpyDeviceToDevice, s));
  if (sync) HIPCHK(h, hipStreamSynchronize(s));
  return 0;
}

int crnn_channel_scales(const crnn_handle* h, int32_t layer, float* out, int32_t cap) {
  if (!h || layer < 0 || layer > 7 || (cap > 0 && !out)) return CRNN_ERR_ARG;
  if (!h->loaded) return fail(h, CRNN_ERR_STATE, "no weights loaded");
  const std::vector<float>& cs = h->chan_scale[layer];
  for (int i = 0; i < cap && i < (int)cs.size(); i++) out[i] = cs[i];
  return (int)cs.size();
}

int crnn_debug_stage(crnn_handle* h, const float* x, int32_t n, int32_t w, const int32_t* widths, int32_t stage,
                     float* out_host, int64_t out_floats) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  tl_cur = &h->lanes[0];
  std::lock_guard<std::mutex> lane_lk(h->lanes[0].mu);
  if (!x || n <= 0 || w < 4 || stage < 0 || stage > 10 || !out_host) return fail(h, CRNN_ERR_ARG, "bad arguments");
  int rc = ensure_device(h);
  if (rc) return rc;
  const Geo g(n, w);
  Plan p;
  if ((rc = prepare_plan(h, g, p))) return rc;
  const size_t nT = (size_t)g.n * g.T;
  if ((rc = ensure_buf(h, tl_cur->io, (nT + g.n) * sizeof(int32_t) + 512))) return rc;
  int32_t* d_lab = (int32_t*)tl_cur->io.p;
  int32_t* d_len = d_lab + ((nT + 63) / 64) * 64;
  HIPCHK(h, hipMemcpyAsync(p.x, x, sizeof(float) * (size_t)n * 32 * w, hipMemcpyHostToDevice, tl_cur->stream));
  HIPCHK(h, hipEventRecord(tl_cur->ev[0], tl_cur->stream));
  tl_cur->want_p0 = stage == 0;
  tl_cur->want_stage = stage;
  rc = run_network(h, g, p, widths, d_lab, d_len, true);
  tl_cur->want_p0 = false;
  tl_cur->want_stage = -1;
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(tl_cur->stream));
  if ((size_t)out_floats < tl_cur->st_floats[stage]) return fail(h, CRNN_ERR_ARG, "out_host too small for this stage");
  if (h->cfg.precision == CRNN_BF16 && stage != 10) {   // stored as bf16 on the device: widen for the caller
    std::vector<uint16_t> tmp(tl_cur->st_floats[stage]);
    HIPCHK(h, hipMemcpy(tmp.data(), tl_cur->st[stage], tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); i++) {
      const uint32_t u = (uint32_t)tmp[i] << 16;
      memcpy(&out_host[i], &u, 4);
    }
    unscale_stage(h, stage, out_host, tmp.size());
    return 0;
  }
  HIPCHK(h, hipMemcpy(out_host, tl_cur->st[stage], tl_cur->st_floats[stage] * sizeof(float), hipMemcpyDeviceToHost));
  unscale_stage(h, stage, out_host, tl_cur->st_floats[stage]);
  return 0;
}

int crnn_synchronize(crnn_handle* h) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->dev_ready) return 0;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  for (int l = 0; l < h->nlanes; l++) HIPCHK(h, hipStreamSynchronize(h->lanes[l].stream));
  return 0;
}

void* crnn_stream(crnn_handle* h, int32_t lane) {
  if (!h || lane < 0) return nullptr;
  std::lock_guard<std::mutex> lk(h->mu);
  if (lane >= h->nlanes) return nullptr;
  tl_cur = &h->lanes[0];
  if (ensure_device(h, h->loaded)) return nullptr;   // with the weights, if there are any: after this call nothing is left to allocate
  return (void*)h->lanes[lane].stream;
}

int crnn_lanes(const crnn_handle* h) { return h ? h->nlanes : CRNN_ERR_ARG; }

int crnn_next_lane(crnn_handle* h) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  return h->next_lane;
}

int64_t crnn_split_timeouts(const crnn_handle* h) {
  if (!h) return 0;
  int64_t n = h->split_timeouts.load();
  // repeats that the asynchronous device path did by itself (the gated one-workgroup kernel counts them in the lane's
  // sync buffer); read without locks: a diagnostic, exact once the lanes are idle
  if (!h->dev_ready || h->dev_pid != getpid()) return n;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(h->cfg.device) != hipSuccess) return n;
  for (int l = 0; l < h->nlanes; l++) {
    const unsigned* d = (const unsigned*)h->lanes[l].sync.p;
    unsigned c = 0;
    if (d && hipMemcpy(&c, d + 1, 4, hipMemcpyDeviceToHost) == hipSuccess) n += c;
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  return n;
}

int crnn_last_timing(crnn_handle* h, float ms[5]) {
  if (!h || !ms) return CRNN_ERR_ARG;
  Lane& L = *h->last.load(std::memory_order_acquire);      // (no handle mutex: waiting for a busy lane must not hold it)
  std::lock_guard<std::mutex> lane_lk(L.mu);
  if (!L.ev_valid) return fail(h, CRNN_ERR_STATE, "nothing has run yet");
  HIPCHK(h, hipEventSynchronize(L.ev[4]));
  for (int i = 0; i < 4; i++) HIPCHK(h, hipEventElapsedTime(&ms[i], L.ev[i], L.ev[i + 1]));
  HIPCHK(h, hipEventElapsedTime(&ms[4], L.ev[0], L.ev[4]));
  return 0;
}

int crnn_set_profiling(crnn_handle* h, int32_t on) {
  if (!h) return CRNN_ERR_ARG;
  std::lock_guard<std::mutex> lk(h->mu);
  h->kprof = on != 0;
  for (Lane& L : h->lanes) L.nk = 0;
  return 0;
}

int crnn_last_kernel_times(crnn_handle* h, int32_t cap, const char** names, float* ms, double* flops,
                           double* bytes) {
  if (!h) return CRNN_ERR_ARG;
  Lane& L = *h->last.load(std::memory_order_acquire);
  std::lock_guard<std::mutex> lane_lk(L.mu);
  if (!h->kprof || L.nk == 0) return 0;
  HIPCHK(h, hipEventSynchronize(L.kev[L.nk]));
  int n = L.nk < cap ? L.nk : cap;
  for (int i = 0; i < n; i++) {
    if (names) names[i] = L.kname[i];
    if (ms) HIPCHK(h, hipEventElapsedTime(&ms[i], L.kev[i], L.kev[i + 1]));
    if (flops) flops[i] = L.kflop[i];
    if (bytes) bytes[i] = L.kbytes[i];
  }
  return n;
}

const char* crnn_last_error(const crnn_handle* h) { return h ? tl_err.c_str() : "null handle"; }

void crnn_destroy(crnn_handle* h) {
  if (!h) return;
  if (h->dev_ready && h->dev_pid == getpid()) {
    (void)hipSetDevice(h->cfg.device);
    for (int l = 0; l < h->nlanes; l++) (void)hipStreamSynchronize(h->lanes[l].stream);
    for (int i = 0; i < 7; i++) { (void)hipFree(h->d_conv_w[i]); (void)hipFree(h->d_conv_b[i]); }
    for (int l = 0; l < 2; l++) {
      (void)hipFree(h->d_wih[l]); (void)hipFree(h->d_bih[l]); (void)hipFree(h->d_whhp[l]);
      (void)hipFree(h->d_emb_w[l]); (void)hipFree(h->d_emb_b[l]);
    }
    (void)hipFree(h->d_lin_w); (void)hipFree(h->d_lin_b); (void)hipFree(h->db_lin_w); (void)hipFree(h->d_zero_bias);
    for (int i = 0; i < 7; i++) (void)hipFree(h->d_wino4[i]);
    for (int i = 0; i < 7; i++) (void)hipFree(h->d_wino4h[i]);
    for (int l = 0; l < 2; l++) (void)hipFree(h->d_wih_h3[l]);
    for (int l = 0; l < 2; l++) (void)hipFree(h->d_whh3[l]);
    (void)hipFree(h->d_emb0_h3);
    (void)hipFree(h->d_conv0_h3);
    (void)hipFree(h->d_cls_wnorm);
    for (int i = 0; i < 7; i++) (void)hipFree(h->db_conv_w[i]);
    for (int l = 0; l < 2; l++) { (void)hipFree(h->db_wih[l]); (void)hipFree(h->db_whhp[l]); (void)hipFree(h->db_emb_w[l]); }
    for (int l = 0; l < h->nlanes; l++) {
      Lane& L = h->lanes[l];
      (void)hipFree(L.arena.p); (void)hipFree(L.pre.p); (void)hipFree(L.io.p); (void)hipHostFree(L.stage.p);
      (void)hipFree(L.page.p); (void)hipFree(L.sync.p);
      for (int i = 0; i < 5; i++) (void)hipEventDestroy(L.ev[i]);
      for (int i = 0; i <= MAXK; i++) (void)hipEventDestroy(L.kev[i]);
      (void)hipStreamDestroy(L.stream);
    }
  }
  delete h;
}

}  // extern "C"
