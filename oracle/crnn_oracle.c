/*
 * crnn_oracle.c -- CPU restatement of the reference's CRNN text-line path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * product in ocr_amd/csrc; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product never calls into it.
 *
 * Parity pinning: the reference holds no tests, golden vectors or fixtures for
 * this path.  The oracle is pinned instead against
 * outputs of the reference itself (crnn/network_torch.py imported on CPU in
 * the build container) captured by tests/golden/make_golden.py and committed
 * under tests/golden/.
 *
 * What it follows (paths relative to the reference root):
 *   crnn/network_torch.py:122-128  CRNN.ocr        BGR -> 'L'
 *   crnn/util.py:6-13              resizeNormalize  Pillow BILINEAR + normalise
 *   crnn/network_torch.py:44-86    layer list       conv/BN/ReLU/pool, BiLSTM x2
 *   crnn/network_torch.py:88-108   forward
 *   crnn/network_torch.py:130-147  predict          argmax over classes
 *   crnn/util.py:16-22             strLabelConverter greedy CTC collapse
 *
 * Third-party arithmetic restated here because the dependency is not part of
 * the reference tree:
 *   - Pillow (requirements.txt pins 12.0.0; container has 12.2.0):
 *     Image.convert('L') (libImaging/Convert.c, L24 macro) and
 *     Image.resize(BILINEAR) (libImaging/Resample.c: precompute_coeffs,
 *     normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc,
 *     ImagingResampleInner).  Checked bit-exact against the installed Pillow
 *     in tests/test_oracle_golden.py.
 *   - PyTorch nn.Conv2d / BatchNorm2d(eval) / MaxPool2d / LSTM / Linear
 *     semantics (gate order i,f,g,o; -inf pooling pad), fp32 arithmetic.
 *
 * Plain C, fp32, NCHW like the reference.  OpenMP only splits independent
 * output planes / rows; every output element is one sequential fp32 sum.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* preprocessing                                                      */
/* ------------------------------------------------------------------ */

/* network_torch.py:124 -- cv2 BGR->RGB swap, then PIL convert('L'):
 * L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16  (Pillow Convert.c L24). */
ORACLE_API void oracle_bgr_to_gray(const uint8_t *bgr, int h, int w, long row_stride,
                                   uint8_t *gray /* [h,w] */) {
  for (int y = 0; y < h; y++) {
    const uint8_t *p = bgr + (long)y * row_stride;
    for (int x = 0; x < w; x++) {
      uint32_t b = p[3 * x + 0], g = p[3 * x + 1], r = p[3 * x + 2];
      gray[(long)y * w + x] = (uint8_t)((19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16);
    }
  }
}

/* util.py:7-9 -- scale = H*1.0/imgH; w = int(W/scale) */
__attribute__((optimize("fp-contract=off")))
ORACLE_API int oracle_resized_width(int src_w, int src_h, int img_h) {
  double scale = (double)src_h * 1.0 / (double)img_h;
  double w = (double)src_w / scale;
  return (int)w;
}

#define PRECISION_BITS (32 - 8 - 2)

__attribute__((optimize("fp-contract=off")))
static double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

/* Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc (BILINEAR,
 * support 1.0, box = whole image).  Returns ksize; allocates bounds[2*out]
 * and kk[out*ksize] (22-bit fixed point). */
/* no FMA contraction: Pillow's build evaluates these as separate double operations */
__attribute__((optimize("fp-contract=off")))
ORACLE_API int oracle_resample_coeffs(int in_size, int out_size, int **bounds_out, int32_t **kk_out) {
  double support, scale, filterscale, center, ww, ss;
  int xx, x, ksize, xmin, xmax;
  float in0 = 0.0f, in1 = (float)in_size;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  support = 1.0 * filterscale;
  ksize = (int)ceil(support) * 2 + 1;
  double *prekk = (double *)malloc(sizeof(double) * (size_t)out_size * ksize);
  int *bounds = (int *)malloc(sizeof(int) * 2 * (size_t)out_size);
  int32_t *kk = (int32_t *)malloc(sizeof(int32_t) * (size_t)out_size * ksize);
  for (xx = 0; xx < out_size; xx++) {
    center = in0 + (xx + 0.5) * scale;
    ww = 0.0;
    ss = 1.0 / filterscale;
    xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double *k = &prekk[(size_t)xx * ksize];
    for (x = 0; x < xmax; x++) {
      double w = bilinear_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; x++)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; x++) k[x] = 0;
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  for (x = 0; x < out_size * ksize; x++) {
    if (prekk[x] < 0)
      kk[x] = (int32_t)(-0.5 + prekk[x] * (1 << PRECISION_BITS));
    else
      kk[x] = (int32_t)(0.5 + prekk[x] * (1 << PRECISION_BITS));
  }
  free(prekk);
  *bounds_out = bounds;
  *kk_out = kk;
  return ksize;
}

ORACLE_API void oracle_free(void *p) { free(p); }

static inline uint8_t clip8(int32_t v) {
  v >>= PRECISION_BITS;
  if (v < 0) return 0;
  if (v > 255) return 255;
  return (uint8_t)v;
}

/* Pillow Image.resize((dw,dh), BILINEAR) on an 8-bit single-band image:
 * horizontal pass over the rows the vertical pass needs, uint8 in between,
 * then vertical pass; a pass is skipped when that dimension is unchanged. */
ORACLE_API int oracle_resize_bilinear_u8(const uint8_t *src, int sw, int sh, long src_stride,
                                         uint8_t *dst, int dw, int dh) {
  if (dw <= 0 || dh <= 0 || sw <= 0 || sh <= 0) return -1;
  int need_h = (dw != sw), need_v = (dh != sh);
  if (!need_h && !need_v) {
    for (int y = 0; y < sh; y++) memcpy(dst + (long)y * dw, src + (long)y * src_stride, (size_t)sw);
    return 0;
  }
  int *bh = NULL, *bv = NULL;
  int32_t *kh = NULL, *kv = NULL;
  int ksh = 0, ksv = 0;
  ksh = oracle_resample_coeffs(sw, dw, &bh, &kh);
  ksv = oracle_resample_coeffs(sh, dh, &bv, &kv);
  int ybox_first = bv[0];
  int ybox_last = bv[dh * 2 - 2] + bv[dh * 2 - 1];
  const uint8_t *in = src;
  long in_stride = src_stride;
  uint8_t *tmp = NULL;
  if (need_h) {
    for (int i = 0; i < dh; i++) bv[i * 2] -= ybox_first;
    int th = ybox_last - ybox_first;
    uint8_t *out = need_v ? (tmp = (uint8_t *)malloc((size_t)dw * th)) : dst;
    for (int yy = 0; yy < th; yy++) {
      const uint8_t *row = src + (long)(yy + ybox_first) * src_stride;
      for (int xx = 0; xx < dw; xx++) {
        int xmin = bh[xx * 2], xmax = bh[xx * 2 + 1];
        const int32_t *k = &kh[(size_t)xx * ksh];
        int32_t ss0 = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < xmax; x++) ss0 += (int32_t)row[x + xmin] * k[x];
        out[(long)yy * dw + xx] = clip8(ss0);
      }
    }
    in = out;
    in_stride = dw;
  }
  if (need_v) {
    for (int yy = 0; yy < dh; yy++) {
      int ymin = bv[yy * 2], ymax = bv[yy * 2 + 1];
      const int32_t *k = &kv[(size_t)yy * ksv];
      for (int xx = 0; xx < dw; xx++) {
        int32_t ss0 = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < ymax; y++) ss0 += (int32_t)in[(long)(y + ymin) * in_stride + xx] * k[y];
        dst[(long)yy * dw + xx] = clip8(ss0);
      }
    }
  }
  free(tmp);
  free(bh); free(bv); free(kh); free(kv);
  return 0;
}

/* util.py:12 -- (x/255.0-0.5)/0.5 in float64, cast to float32 by the caller
 * (network_torch.py:132). */
__attribute__((optimize("fp-contract=off")))
ORACLE_API void oracle_normalize_u8(const uint8_t *src, long n, float *dst) {
  for (long i = 0; i < n; i++) dst[i] = (float)(((double)src[i] / 255.0 - 0.5) / 0.5);
}

/* util.py:6-13 on an 8-bit gray image.  dst must hold 32*oracle_resized_width
 * floats.  Returns the resized width, or -1 when Pillow would raise. */
ORACLE_API int oracle_resize_normalize(const uint8_t *gray, int sw, int sh, long stride,
                                       int img_h, float *dst) {
  if (sh <= 0) return -2; /* ZeroDivisionError in util.py:8 */
  int w = oracle_resized_width(sw, sh, img_h);
  if (w <= 0) return -1;
  uint8_t *tmp = (uint8_t *)malloc((size_t)w * img_h);
  int rc = oracle_resize_bilinear_u8(gray, sw, sh, stride, tmp, w, img_h);
  if (rc == 0) oracle_normalize_u8(tmp, (long)w * img_h, dst);
  free(tmp);
  return rc == 0 ? w : rc;
}

/* ------------------------------------------------------------------ */
/* network                                                            */
/* ------------------------------------------------------------------ */

typedef struct {
  int32_t nclass, nh, lstm, leaky;
  const float *conv_w[7], *conv_b[7];
  const float *bn_w[7], *bn_b[7], *bn_mean[7], *bn_var[7]; /* NULL where no BN */
  const float *w_ih[2][2], *w_hh[2][2], *b_ih[2][2], *b_hh[2][2]; /* [layer][dir] */
  const float *emb_w[2], *emb_b[2];
  const float *lin_w, *lin_b; /* lstmFlag=False variant */
} oracle_weights;

static const int NM[7] = {64, 128, 256, 256, 512, 512, 512};
static const int KS[7] = {3, 3, 3, 3, 3, 3, 2};
static const int PS[7] = {1, 1, 1, 1, 1, 1, 0};

/* nn.Conv2d stride 1: in [B,Ci,H,W] -> out [B,Co,Ho,Wo] */
static void conv2d(const float *in, int B, int Ci, int H, int W, const float *wt, const float *bias,
                   int Co, int k, int p, float *out) {
  int Ho = H + 2 * p - k + 1, Wo = W + 2 * p - k + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int co = 0; co < Co; co++) {
      float *o = out + ((long)b * Co + co) * Ho * Wo;
      for (long i = 0; i < (long)Ho * Wo; i++) o[i] = bias[co];
      for (int ci = 0; ci < Ci; ci++) {
        const float *ip = in + ((long)b * Ci + ci) * H * W;
        const float *wp = wt + ((long)co * Ci + ci) * k * k;
        for (int ky = 0; ky < k; ky++)
          for (int kx = 0; kx < k; kx++) {
            float w = wp[ky * k + kx];
            for (int y = 0; y < Ho; y++) {
              int iy = y + ky - p;
              if (iy < 0 || iy >= H) continue;
              int x0 = p - kx; if (x0 < 0) x0 = 0;
              int x1 = W + p - kx; if (x1 > Wo) x1 = Wo;
              const float *irow = ip + (long)iy * W + (kx - p);
              float *orow = o + (long)y * Wo;
              for (int x = x0; x < x1; x++) orow[x] += w * irow[x];
            }
          }
      }
    }
}

/* nn.BatchNorm2d in eval mode, eps 1e-5, then ReLU / LeakyReLU(0.2) */
static void bn_act(float *x, int B, int C, long HW, const float *g, const float *bt,
                   const float *mean, const float *var, int leaky) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int c = 0; c < C; c++) {
      float *p = x + ((long)b * C + c) * HW;
      if (g) {
        float inv = 1.0f / sqrtf(var[c] + 1e-5f);
        for (long i = 0; i < HW; i++) p[i] = (p[i] - mean[c]) * inv * g[c] + bt[c];
      }
      if (leaky) {
        for (long i = 0; i < HW; i++) p[i] = p[i] > 0.0f ? p[i] : 0.2f * p[i];
      } else {
        for (long i = 0; i < HW; i++) p[i] = p[i] > 0.0f ? p[i] : 0.0f;
      }
    }
}

/* nn.MaxPool2d((2,2),(sy,sx),(0,px)), -inf padding, floor mode */
static void maxpool(const float *in, int B, int C, int H, int W, int sy, int sx, int px, float *out,
                    int *Ho_, int *Wo_) {
  int Ho = (H - 2) / sy + 1, Wo = (W + 2 * px - 2) / sx + 1;
#pragma omp parallel for schedule(static)
  for (long bc = 0; bc < (long)B * C; bc++) {
    const float *ip = in + bc * H * W;
    float *op = out + bc * Ho * Wo;
    for (int y = 0; y < Ho; y++)
      for (int x = 0; x < Wo; x++) {
        float m = -INFINITY;
        for (int dy = 0; dy < 2; dy++)
          for (int dx = 0; dx < 2; dx++) {
            int iy = y * sy + dy, ix = x * sx + dx - px;
            if (ix < 0 || ix >= W || iy >= H) continue;
            float v = ip[(long)iy * W + ix];
            if (v > m) m = v;
          }
        op[(long)y * Wo + x] = m;
      }
  }
  *Ho_ = Ho; *Wo_ = Wo;
}

/* nn.Linear: y[r,o] = b[o] + sum_i x[r,i] w[o,i] */
static void linear(const float *x, long R, int I, const float *w, const float *b, int O, float *y) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; r++) {
    const float *xr = x + r * I;
    for (int o = 0; o < O; o++) {
      const float *wo = w + (long)o * I;
      float s = 0.0f;
      for (int i = 0; i < I; i++) s += xr[i] * wo[i];
      y[r * O + o] = s + b[o];
    }
  }
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one direction of nn.LSTM(I, nh): x [T,B,I] -> out[T,B,2nh] at column dir*nh.
 * Gate order i,f,g,o (rows of W_ih / W_hh in blocks of nh). */
static void lstm_dir(const float *x, int T, int B, int I, int nh, const float *w_ih, const float *w_hh,
                     const float *b_ih, const float *b_hh, int reverse, float *out, int dir) {
  float *xp = (float *)malloc(sizeof(float) * (size_t)T * B * 4 * nh);
  linear(x, (long)T * B, I, w_ih, b_ih, 4 * nh, xp);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; b++) {
    float *h = (float *)calloc((size_t)nh, sizeof(float));
    float *c = (float *)calloc((size_t)nh, sizeof(float));
    float *g = (float *)malloc(sizeof(float) * 4 * (size_t)nh);
    for (int s = 0; s < T; s++) {
      int t = reverse ? T - 1 - s : s;
      const float *xg = xp + ((long)t * B + b) * 4 * nh;
      for (int j = 0; j < 4 * nh; j++) {
        const float *wr = w_hh + (long)j * nh;
        float a = 0.0f;
        for (int k = 0; k < nh; k++) a += h[k] * wr[k];
        g[j] = xg[j] + (a + b_hh[j]);
      }
      float *o = out + ((long)t * B + b) * 2 * nh + (long)dir * nh;
      for (int j = 0; j < nh; j++) {
        float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[nh + j]);
        float gg = tanhf(g[2 * nh + j]), og = sigmoidf_(g[3 * nh + j]);
        c[j] = fg * c[j] + ig * gg;
        h[j] = og * tanhf(c[j]);
        o[j] = h[j];
      }
    }
    free(h); free(c); free(g);
  }
  free(xp);
}

/* T for an input of width W: conv stack of network_torch.py:66-78 */
ORACLE_API int oracle_out_T(int W) { return (W / 2) / 2 + 1; }

/* stage ids for the optional debug taps (NCHW float buffers owned by the
 * caller, sized by the caller from the shapes in SURVEY.md section 8a):
 * 0 pool0, 1 pool1, 2 relu2, 3 pool2, 4 relu4, 5 pool3, 6 relu6 ([B,512,1,T]),
 * 7 lstm0 output after its embedding [T,B,nh], 8 lstm0 raw recurrent [T,B,2nh],
 * 9 lstm1 raw recurrent [T,B,2nh]. */
#define ORACLE_NSTAGE 10

/* network_torch.py:88-108 forward.  x [B,1,32,W] -> logits [T,B,nclass]. */
ORACLE_API int oracle_forward(const oracle_weights *wt, const float *x, int B, int W, float *logits,
                              float *const *stage /* NULL or ORACLE_NSTAGE pointers (each may be NULL) */) {
  if (W < 4) return -1;
  int H = 32, C = 1;
  size_t maxel = (size_t)B * 64 * 32 * W;
  float *a = (float *)malloc(sizeof(float) * maxel);
  float *bbuf = (float *)malloc(sizeof(float) * maxel);
  if (!a || !bbuf) { free(a); free(bbuf); return -2; }
  memcpy(a, x, sizeof(float) * (size_t)B * 32 * W);
  int sid = 0;
  for (int i = 0; i < 7; i++) {
    int Ho = H + 2 * PS[i] - KS[i] + 1, Wo = W + 2 * PS[i] - KS[i] + 1;
    conv2d(a, B, C, H, W, wt->conv_w[i], wt->conv_b[i], NM[i], KS[i], PS[i], bbuf);
    C = NM[i]; H = Ho; W = Wo;
    bn_act(bbuf, B, C, (long)H * W, wt->bn_w[i], wt->bn_b[i], wt->bn_mean[i], wt->bn_var[i], wt->leaky);
    if (i == 0 || i == 1) {
      maxpool(bbuf, B, C, H, W, 2, 2, 0, a, &H, &W);
    } else if (i == 3 || i == 5) {
      maxpool(bbuf, B, C, H, W, 2, 1, 1, a, &H, &W);
    } else {
      float *t = a; a = bbuf; bbuf = t;
    }
    if (stage && stage[sid]) memcpy(stage[sid], a, sizeof(float) * (size_t)B * C * H * W);
    sid++;
  }
  if (H != 1) { free(a); free(bbuf); return -3; } /* network_torch.py:93 assert */
  int T = W, nh = wt->nh;
  /* squeeze(2).permute(2,0,1): [B,512,T] -> [T,B,512] */
  float *seq = bbuf;
  for (int t = 0; t < T; t++)
    for (int b = 0; b < B; b++)
      for (int c = 0; c < 512; c++) seq[((long)t * B + b) * 512 + c] = a[((long)b * 512 + c) * T + t];
  if (!wt->lstm) {
    linear(seq, (long)T * B, 512, wt->lin_w, wt->lin_b, wt->nclass, logits);
    free(a); free(bbuf);
    return T;
  }
  float *rec = (float *)malloc(sizeof(float) * (size_t)T * B * 2 * nh);
  float *mid = (float *)malloc(sizeof(float) * (size_t)T * B * nh);
  for (int d = 0; d < 2; d++)
    lstm_dir(seq, T, B, 512, nh, wt->w_ih[0][d], wt->w_hh[0][d], wt->b_ih[0][d], wt->b_hh[0][d], d, rec, d);
  if (stage && stage[8]) memcpy(stage[8], rec, sizeof(float) * (size_t)T * B * 2 * nh);
  linear(rec, (long)T * B, 2 * nh, wt->emb_w[0], wt->emb_b[0], nh, mid);
  if (stage && stage[7]) memcpy(stage[7], mid, sizeof(float) * (size_t)T * B * nh);
  for (int d = 0; d < 2; d++)
    lstm_dir(mid, T, B, nh, nh, wt->w_ih[1][d], wt->w_hh[1][d], wt->b_ih[1][d], wt->b_hh[1][d], d, rec, d);
  if (stage && stage[9]) memcpy(stage[9], rec, sizeof(float) * (size_t)T * B * 2 * nh);
  linear(rec, (long)T * B, 2 * nh, wt->emb_w[1], wt->emb_b[1], wt->nclass, logits);
  free(rec); free(mid); free(a); free(bbuf);
  return T;
}

/* preds.max(2) (network_torch.py:144): first maximal index wins on ties.
 * logits [T,B,nclass] -> idx [B,T] (the transpose of :145).  margin (optional)
 * receives top1-top2 per (b,t). */
ORACLE_API void oracle_argmax(const float *logits, int T, int B, int nclass, int32_t *idx, float *margin) {
  for (int t = 0; t < T; t++)
    for (int b = 0; b < B; b++) {
      const float *p = logits + ((long)t * B + b) * nclass;
      int best = 0;
      float bv = p[0], second = -INFINITY;
      for (int k = 1; k < nclass; k++) {
        if (p[k] > bv) { second = bv; bv = p[k]; best = k; }
        else if (p[k] > second) second = p[k];
      }
      idx[(long)b * T + t] = best;
      if (margin) margin[(long)b * T + t] = bv - second;
    }
}

/* util.py:16-22 strLabelConverter on class indices: emit k-1 (alphabet
 * position) when k != 0 and k != previous raw index.  Returns the count. */
ORACLE_API int oracle_ctc_collapse(const int32_t *idx, int T, int32_t *out) {
  int n = 0;
  for (int i = 0; i < T; i++)
    if (idx[i] != 0 && !(i > 0 && idx[i - 1] == idx[i])) out[n++] = idx[i] - 1;
  return n;
}
