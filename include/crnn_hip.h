/*
 * crnn_hip.h -- C ABI of libcrnn_hip.so, the MI355X (gfx950) implementation of
 * the reference's CRNN text-line recognition path.
 *
 * The reference (ClassesOver/ocr) has no native code: the path is Python +
 * torch (crnn/network_torch.py, crnn/util.py).  Each entry point below names
 * the reference interface it replaces; the Python class that mirrors
 * `crnn.network_torch.CRNN` on top of this ABI is ocr_amd/crnn/network_hip.py
 * and INTEGRATION.md shows the binding a maintainer adds to the reference.
 *
 * Conventions: plain pointers and sizes, no torch types.  Every function
 * returns 0 on success or a negative crnn_status; nothing throws across the
 * boundary.  The caller owns every buffer.  A handle is bound to one device
 * and owns up to 8 lanes (a HIP stream plus a workspace each); it is internally
 * mutex-guarded, so the two gunicorn threads of a worker (gunicorn.conf.py:24)
 * may share it and run on different lanes at the same time.  Device state is
 * created lazily on the first call that needs it, so a handle may be created
 * before fork() (gunicorn preload_app, gunicorn.conf.py:57) and used after.
 *
 * Stream ordering of the device-pointer entry points (crnn_forward_device and
 * its _noalloc variant): the library queues its work on one of ITS OWN streams
 * (created non-blocking: no implicit ordering with the NULL stream or with a
 * framework's current stream).  Every input buffer must be completely written,
 * and every output buffer free to be overwritten, before the call is made --
 * either synchronise the producing stream first, or ask crnn_next_lane() which
 * lane the next call will take and make crnn_stream(h, lane) wait on an event of
 * the producer (hipStreamWaitEvent).  Results are complete when that lane's
 * stream reaches the point after the call (sync != 0, crnn_synchronize, or an
 * event recorded on crnn_stream(h, returned lane)).
 */
#ifndef CRNN_HIP_H
#define CRNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRNN_API __attribute__((visibility("default")))

typedef struct crnn_handle crnn_handle;

typedef enum {
  CRNN_OK = 0,
  CRNN_ERR_ARG = -1,      /* bad argument (null pointer, bad shape, missing tensor) */
  CRNN_ERR_HIP = -2,      /* a HIP runtime call failed; see crnn_last_error */
  CRNN_ERR_STATE = -3,    /* weights not loaded */
  CRNN_ERR_ZERO_H = -4,   /* source height 0: ZeroDivisionError in crnn/util.py:7-8 */
  CRNN_ERR_ZERO_W = -5,   /* resized width 0: Pillow's resize raises ValueError (crnn/util.py:10) */
  CRNN_ERR_NOMEM = -6
} crnn_status;

#define CRNN_FP32 0
#define CRNN_BF16 1

/* Constructor arguments of CRNN(imgH, nc, nclass, nh, leakyRelu, lstmFlag, GPU, alphabet)
 * -- crnn/network_torch.py:37; built at settings.py:45 as CRNN(32,1,nclass,256,False,LSTMFLAG,GPU,alphabet). */
typedef struct {
  int32_t img_h;      /* 32 */
  int32_t nc;         /* 1 */
  int32_t nclass;     /* len(alphabet)+1 */
  int32_t nh;         /* 256 */
  int32_t leaky_relu; /* leakyRelu */
  int32_t lstm;       /* lstmFlag */
  int32_t device;     /* HIP device ordinal (config.GPUID) */
  int32_t lanes;      /* independent (stream, workspace) lanes crnn_forward_device rotates over; 0 -> 1, max 8 */
  int32_t precision;  /* CRNN_FP32 (the parity path: fp32-accurate arithmetic -- GEMM operands as (hi, lo) fp16 pairs on
                         the fp16 MFMA with fp32 accumulation, or the f32 MFMA with CRNN_F16X3=0) or CRNN_BF16 (bf16 MFMA
                         inputs, fp32 accumulate; reports character agreement, not held to the 1e-3 logit bound) */
  int32_t reserved;
} crnn_config;

/* One entry of the flat state dict CRNN.load_weights reads (network_torch.py:110-117):
 * names as in the reference module ("cnn.conv0.weight", "rnn.0.rnn.weight_ih_l0_reverse", ...;
 * a leading "module." is stripped like :115), fp32, row-major, torch shapes. */
typedef struct {
  const char* name;
  const float* data;
  int32_t ndim;
  int64_t shape[4];
} crnn_tensor;

/* CRNN.__init__ (network_torch.py:37-86).  Allocates host state only. */
CRNN_API int crnn_create(const crnn_config* cfg, crnn_handle** out);

/* CRNN.load_weights (network_torch.py:110-120): takes the named host tensors (borrowed for
 * the call), folds BatchNorm (eval mode, eps 1e-5) into the preceding conv, reorders conv
 * weights to (ky,kx,ci), packs W_hh for the recurrent kernel; uploaded to the device lazily. */
CRNN_API int crnn_load_weights(crnn_handle* h, const crnn_tensor* tensors, int32_t n);

/* crnn/util.py:6-9: scale = H/imgH; w = int(W/scale).  Returns w, or a negative status. */
CRNN_API int crnn_resized_width(const crnn_handle* h, int32_t src_w, int32_t src_h);

/* T = number of CTC timesteps for an input of width w (conv stack of network_torch.py:66-78). */
CRNN_API int crnn_out_len(int32_t w);

/* CRNN.ocr's colour conversion (network_torch.py:124) + resizeNormalize (util.py:6-13) for n
 * host images (uint8; channels[i] = 3 for BGR rows as the detectors pass them, 1 for PIL 'L').
 * Writes normalised fp32 [n,32,wp] (zero right padding) to out_host and the resized widths to
 * out_w.  wp must be >= every resized width.  Debug/parity entry point. */
CRNN_API int crnn_preprocess(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights,
                    const int32_t* widths, const int32_t* row_strides, const int32_t* channels,
                    int32_t n, int32_t wp, float* out_host, int32_t* out_w);

/* pad_mode for crnn_predict_batch */
#define CRNN_PAD_PER_LINE 0 /* every line decoded exactly as CRNN.predict would decode it alone */
#define CRNN_PAD_REFERENCE 1 /* CRNN.predict_batch semantics (network_torch.py:179-196): zero
                                right-padding to the widest line, all T of the padded width decoded */

/* CRNN.predict / predict_job / predict_batch (network_torch.py:130-200) for n host images.
 *   out_raw   [n,tmax] int32  arg-max class per timestep (preds.max(2), :144); may be NULL
 *   out_lab   [n,tmax] int32  greedy-CTC-collapsed alphabet positions (class-1), util.py:16-22
 *   out_len   [n]      int32  number of labels per line; lines with resized width < 8 get 0
 *                             (the '' of :141-142)
 *   out_t     [n]      int32  timesteps decoded per line; may be NULL
 *   out_logits [n,tmax,nclass] fp32 or NULL (debug: the forward() output, batch-major)
 * tmax must be >= crnn_out_len(widest resized line).  Lines may have different sizes. */
CRNN_API int crnn_predict_batch(crnn_handle* h, const uint8_t* const* imgs, const int32_t* heights,
                       const int32_t* widths, const int32_t* row_strides, const int32_t* channels,
                       int32_t n, int32_t pad_mode, int32_t tmax, int32_t* out_raw, int32_t* out_lab,
                       int32_t* out_len, int32_t* out_t, float* out_logits);

/* Crops that are rectangles of ONE page.  The detectors slice every crop out of the
 * page they were given -- im0[y1:y2, x1:x2] at obj_det/vat_detect.py:291,318, stock_detect_v2.py:134,
 * stock_detect.py:128, bill_detect.py:115 -- and TextOcrModel hands the slices to ocr_predict one by
 * one (ocr_context.py:119,307).  crnn_set_page uploads the page once (HxWx3 BGR or HxW gray uint8,
 * rows row_stride bytes apart) and keeps it in device memory until the next crnn_set_page;
 * crnn_predict_rects then runs crnn_predict_batch's path on n rectangles rects[n][4] =
 * (x1, y1, x2, y2), x2/y2 exclusive and clipped to the page like a numpy slice, with the gray
 * conversion and resize reading straight from the resident page.  Outputs, pad_mode and error codes
 * as crnn_predict_batch (CRNN_ERR_ZERO_H for an empty rectangle, CRNN_ERR_STATE without a page). */
CRNN_API int crnn_set_page(crnn_handle* h, const uint8_t* page, int32_t height, int32_t width,
                  int32_t row_stride, int32_t channels);
CRNN_API int crnn_predict_rects(crnn_handle* h, const int32_t* rects, int32_t n, int32_t pad_mode,
                       int32_t tmax, int32_t* out_raw, int32_t* out_lab, int32_t* out_len, int32_t* out_t,
                       float* out_logits);
/* Both steps in one call, for concurrent callers (the two request threads of a gunicorn worker,
 * gunicorn.conf.py:24, each with its own page): takes any free lane of the handle, uploads the page into
 * that lane's page buffer and recognises the rectangles there, so two pages are in flight at once.
 * crnn_set_page / crnn_predict_rects above always use lane 0 and serialise. */
CRNN_API int crnn_predict_page(crnn_handle* h, const uint8_t* page, int32_t height, int32_t width,
                      int32_t row_stride, int32_t channels, const int32_t* rects, int32_t n,
                      int32_t pad_mode, int32_t tmax, int32_t* out_raw, int32_t* out_lab, int32_t* out_len,
                      int32_t* out_t, float* out_logits);

/* The same path for lines already resident in device memory at height 32 (the layout the
 * width-bucketing scheduler and bench.py use): d_gray uint8 [n,32,w] device pointer; widths
 * host int32[n] (valid width of each line, <= w) or NULL when every line is w wide.
 * Outputs are DEVICE pointers: d_lab [n,T] int32, d_len [n] int32, d_raw [n,T] int32 or NULL,
 * with T = crnn_out_len(w).  Runs asynchronously unless sync != 0.  Up to 512 lines the LSTM recurrence runs in its
 * hidden-split form either way, like the host paths (synchronous calls: 16 lines per group of four workgroups, the
 * lowest latency; asynchronous calls from 128 lines: 32 lines per group, half the workgroups, and a repeat with the
 * one-workgroup kernel queued behind it that only runs if a hand-off timed out): the same results bit for bit.
 * Successive calls rotate over
 * the handle's lanes (crnn_config.lanes): the call returns the lane index (>= 0) it was queued
 * on, whose stream is crnn_stream(h, lane); work on different lanes overlaps on the device.
 * No reference counterpart (the reference never keeps data on a device). */
CRNN_API int crnn_forward_device(crnn_handle* h, const uint8_t* d_gray, int32_t n, int32_t w,
                        const int32_t* widths, int32_t* d_raw, int32_t* d_lab, int32_t* d_len,
                        int32_t sync);

/* Lane (>= 0) the next crnn_forward_device call on this handle will be queued on, so that a single producer
 * thread can order its own stream before that lane's stream (see "Stream ordering" above).  With several
 * threads calling crnn_forward_device on one handle the answer can be stale by the time it is used. */
CRNN_API int crnn_next_lane(crnn_handle* h);

/* crnn_forward_device for callers that must not allocate (tight serving loops, harnesses that forbid hipMalloc, code
 * that runs while a stream capture in global mode is active in the process: allocation calls are illegal then; the work
 * itself is queued on the library's own stream and is NOT part of the capture; with widths != NULL the call also
 * synchronises its stream and makes a pageable host-to-device copy, so pass widths == NULL there).
 * It has no counterpart in the reference.  Same arguments, outputs and
 * results as crnn_forward_device, but the whole activation workspace lives in a fixed array of the code object
 * (CRNN_NOALLOC_SCRATCH_BYTES per GPU, held from module load until the process ends) and the call makes no
 * hipMalloc / hipFree / hipHostMalloc and creates no stream or event: the handle's device state and weights must
 * already be on the device (any earlier compute call on this handle in this process, or crnn_stream(h, 0) after
 * crnn_load_weights); otherwise CRNN_ERR_STATE.  A batch whose workspace does not fit (about 4 lines of width
 * 280 with the 5530-class head; crnn_noalloc_workspace_bytes tells) is refused with CRNN_ERR_ARG and a message
 * naming the sizes.  All calls on one GPU share that GPU's array and are queued on lane 0's stream of their
 * handle: calls through ONE handle are ordered by that stream; calls through different handles (or processes'
 * threads using different handles) on the same GPU that overlap in time race on the array and must be
 * serialised by the caller.  Nothing in the array has to start at zero (every kernel writes what it later reads),
 * so a call is independent of what the previous one left there.  Returns 0 (the lane) or a negative status. */
#define CRNN_NOALLOC_SCRATCH_BYTES (64u << 20)
CRNN_API int crnn_forward_device_noalloc(crnn_handle* h, const uint8_t* d_gray, int32_t n, int32_t w,
                                const int32_t* widths, int32_t* d_raw, int32_t* d_lab, int32_t* d_len,
                                int32_t sync);
/* Workspace bytes a batch of n lines of padded width w needs (what _noalloc compares with the limit). */
CRNN_API int64_t crnn_noalloc_workspace_bytes(const crnn_handle* h, int32_t n, int32_t w);

/* Number of times the hidden-split recurrence (batches of up to 512 lines) gave up waiting for its partner
 * workgroups on this handle and the layer was repeated with the one-workgroup kernel, repeats made on the device
 * by asynchronous calls included (a latency cliff a service wants to see; 0 in normal operation).  No reference
 * counterpart. */
CRNN_API int64_t crnn_split_timeouts(const crnn_handle* h);

/* Diagnostic: the load-time channel equalisation's factors (crnn_load_weights rescales every hidden channel by an exact
 * power of two on the producing side and its inverse on the consuming side -- the network function is unchanged).
 * layer 0..6 = conv0..conv6 output channels, 7 = the first embedding's outputs.  Copies min(cap, channels) factors to out
 * and returns the number of channels, or a negative status (CRNN_ERR_STATE before crnn_load_weights).  Host-only, needs no
 * GPU.  No reference counterpart (the reference has no reduced-precision path to protect). */
CRNN_API int crnn_channel_scales(const crnn_handle* h, int32_t layer, float* out, int32_t cap);

/* Debug taps for layer-by-layer parity: runs crnn_forward_device's network on normalised fp32
 * host input x [n,32,w] and copies stage `stage` (0 pool0, 1 pool1, 2 relu2, 3 pool2, 4 relu4,
 * 5 pool3, 6 relu6/sequence, 7 lstm0 out, 8 lstm0 recurrent, 9 lstm1 recurrent, 10 logits),
 * NHWC / batch-major as stored on the device, to out_host (capacity out_floats); the channel equalisation's factors are
 * divided out, so the values are the reference's activations. */
CRNN_API int crnn_debug_stage(crnn_handle* h, const float* x, int32_t n, int32_t w, const int32_t* widths,
                     int32_t stage, float* out_host, int64_t out_floats);

/* Blocks until everything queued on every lane of the handle has finished. */
CRNN_API int crnn_synchronize(crnn_handle* h);

/* The HIP stream (hipStream_t as void*) of one lane, for callers that time with HIP events or
 * order their own work (e.g. the copy-back of results) against it.  crnn_lanes: lane count. */
CRNN_API void* crnn_stream(crnn_handle* h, int32_t lane);
CRNN_API int crnn_lanes(const crnn_handle* h);

/* Device time of the most recent crnn_forward_device / crnn_predict_batch in ms (HIP events
 * on the handle's stream), split by stage: [0] preprocess, [1] conv stack, [2] lstm+linear,
 * [3] argmax+ctc, [4] total.  Valid after crnn_synchronize. */
CRNN_API int crnn_last_timing(crnn_handle* h, float ms[5]);

/* Per-launch profiling of crnn_forward_device: when on, a HIP event is recorded on the handle's
 * stream after every kernel launch.  crnn_last_kernel_times returns, for the most recent
 * crnn_forward_device, up to `cap` launches in order: a static name, the time between the
 * events that bracket the launch (ms), its algorithmic FLOPs (2*MACs) and algorithmic bytes
 * (compulsory reads + writes).  Returns the number of launches written.  Measurement aid for
 * bench.py's roofline line; no reference counterpart. */
CRNN_API int crnn_set_profiling(crnn_handle* h, int32_t on);
CRNN_API int crnn_last_kernel_times(crnn_handle* h, int32_t cap, const char** names, float* ms,
                                    double* flops, double* bytes);

/* Message of the last failing call on this handle (never NULL). */
CRNN_API const char* crnn_last_error(const crnn_handle* h);

CRNN_API void crnn_destroy(crnn_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CRNN_HIP_H */
