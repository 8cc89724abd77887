"""MI355X drop-in for the reference's ``crnn/network_torch.py``.

``CRNN`` keeps the reference class's constructor and method signatures
(``network_torch.py:37,110,122,130,149,157``) so that ``settings.py:45-51``
(``crnn = CRNN(32, 1, nclass, 256, leakyRelu=False, lstmFlag=LSTMFLAG, GPU=GPU,
alphabet=alphabet)``; ``ocr_predict = crnn.ocr``) and every caller above it
(``obj_det/ocr_context.py:28``) work unchanged.  All arithmetic runs in the HIP
library behind ``include/crnn_hip.h``; this file only marshals buffers and
turns label indices into text.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from .util import labels_to_text


def _as_gray_array(image):
    """PIL 'L' image (what the reference's predict receives) or a uint8 HxW array."""
    if hasattr(image, "mode") and hasattr(image, "size"):
        if image.mode != "L":
            # the reference's view(1,1,H,W) at network_torch.py:139 fails for multi-band images
            raise ValueError("CRNN.predict expects a PIL image of mode 'L', got %r" % (image.mode,))
        if image.size[1] == 0:
            raise ZeroDivisionError("float division by zero")  # util.py:7-8
        if image.size[0] == 0:
            raise ValueError("height and width must be > 0")
        return np.asarray(image)
    a = np.asarray(image)
    if a.ndim != 2 or a.dtype != np.uint8:
        raise ValueError("expected a PIL 'L' image or a uint8 HxW array")
    return a


class CRNN(object):
    """Same surface as ``crnn.network_torch.CRNN``; inference only."""

    def __init__(self, imgH, nc, nclass, nh, leakyRelu=False, lstmFlag=True, GPU=False, alphabet=None, device=None,
                 lanes=2, precision='fp32'):
        assert imgH % 16 == 0, 'imgH has to be a multiple of 16'
        if imgH != 32 or nc != 1 or nh != 256:
            raise ValueError("the HIP path implements the shipped configuration: imgH=32, nc=1, nh=256")
        self.lstmFlag = lstmFlag
        self.leakyRelu = bool(leakyRelu)
        self.GPU = GPU
        self.alphabet = alphabet
        self.nclass = nclass
        self.precision = precision
        if device is None:
            device = int(os.environ.get("CRNN_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.device = device
        self._lib = _capi.lib()
        self._page_shape = None
        cfg = _capi.Config(imgH, nc, nclass, nh, int(bool(leakyRelu)), int(bool(lstmFlag)), device, lanes,
                           {'fp32': _capi.FP32, 'bf16': _capi.BF16}[precision], 0)
        h = C.c_void_p()
        rc = self._lib.crnn_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise ValueError("crnn_create failed (status %d)" % rc)
        self._h = h
        self._alpha_cp = None
        if alphabet is not None:
            self._alpha_cp = np.frombuffer(alphabet.encode("utf-32-le"), dtype="<u4")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.crnn_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict):
        """Flat name -> tensor mapping with the reference's keys (numpy arrays or torch tensors)."""
        keep, arr = [], (_capi.Tensor * len(state_dict))()
        n = 0
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(a)
            arr[n].name = k.encode("utf-8")
            arr[n].data = a.ctypes.data_as(C.POINTER(C.c_float))
            arr[n].ndim = a.ndim
            for i, s in enumerate(a.shape):
                arr[n].shape[i] = s
            n += 1
        _capi.check(self._h, self._lib.crnn_load_weights(self._h, arr, n))
        return self

    def load_weights(self, path):
        """network_torch.py:110-120: a flat ``.pth`` state dict, keys optionally prefixed ``module.``."""
        import torch  # tensor plumbing only: reads the checkpoint
        sd = torch.load(path, map_location="cpu", weights_only=True)
        self.load_state_dict({k: v for k, v in sd.items()})

    def eval(self):
        return self

    def forward(self, input):
        """network_torch.py:88-108: normalised input ``[b, 1, 32, w]`` (torch tensor or array, float32) ->
        logits ``[T, b, nclass]``, time-major like the reference's ``forward`` (``conv.permute(2, 0, 1)``, :95).
        A torch tensor in gives a torch tensor (CPU) out.  Every sample is taken at the full width w, as the
        reference does (no per-line masks)."""
        is_torch = hasattr(input, "detach")
        x = input.detach().cpu().numpy() if is_torch else np.asarray(input)
        if x.ndim != 4 or x.shape[1] != 1 or x.shape[2] != 32:
            raise ValueError("expected input of shape [b, 1, 32, w], got %s" % (x.shape,))
        out = self.debug_stage(np.ascontiguousarray(x[:, 0], dtype=np.float32), 10)      # [b, T, nclass]
        out = np.ascontiguousarray(np.transpose(out, (1, 0, 2)))
        if is_torch:
            import torch
            return torch.from_numpy(out)
        return out

    __call__ = forward

    # --------------------------------------------------------------- inference
    def _run(self, images, pad_mode, want_raw=False, want_logits=False):
        il = _capi.ImageList(images)
        n = il.n
        wmax = 0
        for i in range(n):
            w = _capi.check(self._h, self._lib.crnn_resized_width(self._h, il.w[i], il.h[i]))
            if w <= 0:
                raise ValueError("height and width must be > 0")
            wmax = max(wmax, w)
        tmax = max(1, self._lib.crnn_out_len(wmax))
        lab = np.zeros((n, tmax), np.int32)
        ln = np.zeros(n, np.int32)
        tt = np.zeros(n, np.int32)
        raw = np.zeros((n, tmax), np.int32) if want_raw else None
        logits = np.zeros((n, tmax, self.nclass), np.float32) if want_logits else None
        rc = self._lib.crnn_predict_batch(
            self._h, il.ptrs, il.h, il.w, il.stride, il.ch, n, pad_mode, tmax,
            raw.ctypes.data if want_raw else None, lab.ctypes.data, ln.ctypes.data, tt.ctypes.data,
            logits.ctypes.data if want_logits else None)
        _capi.check(self._h, rc)
        return lab, ln, tt, raw, logits

    def _texts(self, lab, ln):
        return [labels_to_text(lab[i, :ln[i]], self.alphabet, self._alpha_cp) for i in range(len(ln))]

    def ocr(self, image):
        """network_torch.py:122-128: HxWx3 uint8 BGR crop (a view into the page is fine) -> text."""
        a = np.asarray(image)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8 or a.size == 0:
            # cv2.cvtColor(image, COLOR_BGR2RGB) raises on anything else (:124)
            raise ValueError("CRNN.ocr expects a non-empty HxWx3 uint8 BGR array, got %s %s" % (a.dtype, a.shape))
        return self.predict_lines([a])[0]

    def predict(self, image):
        """network_torch.py:130-147: PIL 'L' image -> text ('' when the resized width is < 8)."""
        return self.predict_lines([_as_gray_array(image)])[0]

    def predict_job(self, boxes):
        """network_torch.py:149-155: per-box predict; results are what predict gives each box alone."""
        if len(boxes) == 0:
            return boxes
        texts = self.predict_lines([b['img'] for b in boxes])
        for b, t in zip(boxes, texts):
            b['text'] = t
        return boxes

    def predict_batch(self, boxes, batch_size=1):
        """network_torch.py:157-200 with its exact semantics: chunks of ``batch_size`` boxes, each
        chunk zero-right-padded to its widest line and all timesteps of the padded width decoded."""
        N = len(boxes)
        res = []
        for i in range(0, N, batch_size):
            imgs = [_as_gray_array(b['img']) for b in boxes[i:i + batch_size]]
            lab, ln, _, _, _ = self._run(imgs, _capi.PAD_REFERENCE)
            res.extend(self._texts(lab, ln))
        for i in range(N):
            boxes[i]['text'] = res[i]
        return boxes

    # ------------------------------------------------ batched, predict-exact API
    def predict_lines(self, images, max_batch=256, width_spread=1.25, min_cols=24576):
        """Texts of many lines, each identical to ``predict`` on that line alone.

        Lines are sorted by resized width and cut into buckets (at most ``max_batch`` lines, widest
        / narrowest <= ``width_spread``, not enforced while a bucket has fewer than ``min_cols`` padded
        columns -- small batches are latency-bound); each bucket is one padded device batch with per-line
        width masks.  ``images``: PIL 'L' images, uint8 HxW arrays or HxWx3 BGR arrays."""
        arrs = []
        for im in images:
            a = np.asarray(im) if not (hasattr(im, "mode") and hasattr(im, "size")) else _as_gray_array(im)
            arrs.append(a)
        from .batching import width_buckets
        widths = []
        for a in arrs:
            if a.ndim < 2 or a.shape[0] == 0:
                raise ZeroDivisionError("float division by zero")
            widths.append(_capi.check(self._h, self._lib.crnn_resized_width(self._h, a.shape[1], a.shape[0])))
        out = [None] * len(arrs)
        pad_mode = _capi.PAD_PER_LINE
        if self.leakyRelu:
            # Per-line width masks rely on post-ReLU zeros standing in for the -inf pooling pad;
            # with LeakyReLU activations can be negative, so only lines of IDENTICAL width share a
            # batch (then no mask is needed and the result is exact).
            width_spread = 1.0
            narrow = [i for i, w in enumerate(widths) if w < 8]
            for i in narrow:
                out[i] = ''
            keep = [i for i, w in enumerate(widths) if w >= 8]
            arrs_k, widths_k = [arrs[i] for i in keep], [widths[i] for i in keep]
            for bucket in width_buckets(widths_k, max_batch, 1.0):
                lab, ln, _, _, _ = self._run([arrs_k[i] for i in bucket], _capi.PAD_REFERENCE)
                for i, t in zip(bucket, self._texts(lab, ln)):
                    out[keep[i]] = t
            return out
        buckets = width_buckets(widths, max_batch, width_spread, min_cols)

        def one(bucket):
            lab, ln, _, _, _ = self._run([arrs[i] for i in bucket], pad_mode)
            return self._texts(lab, ln)
        nthreads = min(len(buckets), self.lanes, 4)
        if nthreads > 1:
            # buckets run concurrently on different lanes of the handle (ctypes releases the GIL during
            # the call): one bucket's packing / upload / copy-back overlaps another's device work
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nthreads) as ex:
                results = list(ex.map(one, buckets))
        else:
            results = [one(b) for b in buckets]
        for bucket, texts in zip(buckets, results):
            for i, t in zip(bucket, texts):
                out[i] = t
        return out

    # ------------------------------------------- crops as rectangles of one page
    @staticmethod
    def _page_array(page):
        a = np.asarray(page)
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3) or a.size == 0:
            raise ValueError("expected a non-empty HxW or HxWx3 uint8 page, got %s %s" % (a.dtype, a.shape))
        ch = 1 if a.ndim == 2 else 3
        if a.strides[-1] != 1 or (a.ndim == 3 and a.strides[1] != 3) or a.strides[0] < a.shape[1] * ch:
            a = np.ascontiguousarray(a)
        return a, ch

    def set_page(self, page):
        """Upload a page (HxWx3 uint8 BGR as cv2.imread gives it, or HxW gray; any row stride) once;
        it stays in device memory (lane 0) until the next set_page."""
        a, ch = self._page_array(page)
        self._page_shape = a.shape[:2]
        _capi.check(self._h, self._lib.crnn_set_page(self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0], ch))

    def _rect_plan(self, rects, shape, max_batch, width_spread, min_cols):
        r = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
        if (r < 0).any():
            raise ValueError("negative rectangle coordinate")
        ph, pw = shape
        hh = np.minimum(r[:, 3], ph) - r[:, 1]
        ww = np.minimum(r[:, 2], pw) - r[:, 0]
        if (hh <= 0).any():
            raise ZeroDivisionError("float division by zero")       # util.py:8 on an empty crop
        widths = [_capi.check(self._h, self._lib.crnn_resized_width(self._h, int(max(w, 0)), int(h)))
                  for w, h in zip(ww, hh)]
        if min(widths) <= 0:
            raise ValueError("height and width must be > 0")
        from .batching import width_buckets
        exact = 1.0 if self.leakyRelu else width_spread
        return r, widths, width_buckets(widths, max_batch, exact, 0 if self.leakyRelu else min_cols)

    def _rect_bucket(self, r, widths, bucket, call):
        """One bucket of rectangles through ``call(rects_ptr, n, mode, tmax, lab_ptr, len_ptr)``."""
        if self.leakyRelu and widths[bucket[0]] < 8:
            return [''] * len(bucket)
        rb = np.ascontiguousarray(r[bucket])
        tmax = max(1, self._lib.crnn_out_len(max(widths[i] for i in bucket)))
        lab = np.zeros((len(bucket), tmax), np.int32)
        ln = np.zeros(len(bucket), np.int32)
        mode = _capi.PAD_REFERENCE if self.leakyRelu else _capi.PAD_PER_LINE
        _capi.check(self._h, call(rb.ctypes.data, len(bucket), mode, tmax, lab.ctypes.data, ln.ctypes.data))
        return self._texts(lab, ln)

    def predict_rects(self, rects, max_batch=256, width_spread=1.25, min_cols=24576):
        """Texts of the crops page[y1:y2, x1:x2] for rects = [(x1, y1, x2, y2), ...] of the page given to
        set_page -- what ``[self.ocr(page[y1:y2, x1:x2]) for ...]`` returns (ocr_context.py:119,307 fed by
        vat_detect.py:318), without cutting or uploading the crops: the resize reads the resident page.
        Width-bucketed like predict_lines; ends beyond the page are clipped as numpy slicing does."""
        if len(rects) == 0:
            return []
        if self._page_shape is None:
            raise RuntimeError("no page: call set_page first")
        r, widths, buckets = self._rect_plan(rects, self._page_shape, max_batch, width_spread, min_cols)
        out = [None] * len(r)
        for bucket in buckets:
            texts = self._rect_bucket(r, widths, bucket, lambda rp, n, mode, tmax, lab, ln: self._lib.crnn_predict_rects(
                self._h, rp, n, mode, tmax, None, lab, ln, None, None))
            for i, t in zip(bucket, texts):
                out[i] = t
        return out

    def ocr_page(self, page, rects, max_batch=256, width_spread=1.25, min_cols=24576):
        """Texts of page[y1:y2, x1:x2] for every rectangle: ``[self.ocr(page[y1:y2, x1:x2]) for ...]`` with one
        upload of the page.  When the rectangles make one device batch (the usual case for a page's worth of
        crops) this is ONE library call that takes any free lane, so concurrent callers with different pages
        overlap on the device; otherwise set_page + predict_rects (lane 0)."""
        if len(rects) == 0:
            return []
        a, ch = self._page_array(page)
        r, widths, buckets = self._rect_plan(rects, a.shape[:2], max_batch, width_spread, min_cols)
        if len(buckets) != 1:
            self.set_page(a)
            return self.predict_rects(rects, max_batch, width_spread, min_cols)
        texts = self._rect_bucket(r, widths, buckets[0], lambda rp, n, mode, tmax, lab, ln: self._lib.crnn_predict_page(
            self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0], ch, rp, n, mode, tmax, None, lab, ln, None, None))
        out = [None] * len(r)
        for i, t in zip(buckets[0], texts):      # the bucket is in width order
            out[i] = t
        return out

    # ------------------------------------------------------------- debug / test
    def forward_logits(self, images, pad_mode=_capi.PAD_PER_LINE):
        """(raw argmax [n,tmax], T per line, logits [n,tmax,nclass]) for parity tests."""
        lab, ln, tt, raw, logits = self._run(images, pad_mode, want_raw=True, want_logits=True)
        return raw, tt, logits, lab, ln

    def preprocess(self, images):
        """resizeNormalize of each image through the HIP kernels: list of float32 [32,w] arrays."""
        il = _capi.ImageList(images)
        n = il.n
        ws = [_capi.check(self._h, self._lib.crnn_resized_width(self._h, il.w[i], il.h[i])) for i in range(n)]
        if min(ws) <= 0:
            raise ValueError("height and width must be > 0")
        wp = max(ws)
        out = np.empty((n, 32, wp), np.float32)
        ow = (C.c_int32 * n)()
        _capi.check(self._h, self._lib.crnn_preprocess(self._h, il.ptrs, il.h, il.w, il.stride, il.ch, n, wp,
                                                       out.ctypes.data, ow))
        return [out[i, :, :ow[i]].copy() for i in range(n)]

    def debug_stage(self, x, stage, widths=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, hh, w = x.shape
        assert hh == 32
        w1, w2 = w // 2, w // 2 // 2
        T = w2 + 1
        shapes = [(n, 16, w1, 64), (n, 8, w2, 128), (n, 8, w2, 256), (n, 4, w2 + 1, 256), (n, 4, w2 + 1, 512),
                  (n, 2, w2 + 2, 512), (n, T, 512), (n, T, 256), (n, T, 512), (n, T, 512), (n, T, self.nclass)]
        out = np.empty(shapes[stage], np.float32)
        wv = None
        if widths is not None:
            wv = np.ascontiguousarray(widths, dtype=np.int32)
        rc = self._lib.crnn_debug_stage(self._h, x.ctypes.data, n, w, wv.ctypes.data if wv is not None else None,
                                        stage, out.ctypes.data, out.size)
        _capi.check(self._h, rc)
        return out

    # --------------------------------------------------- device-resident batches
    @property
    def lanes(self):
        return self._lib.crnn_lanes(self._h)

    def stream_ptr(self, lane=0):
        """hipStream_t of one lane as an integer (for torch.cuda.ExternalStream / HIP events)."""
        p = self._lib.crnn_stream(self._h, lane)
        if not p:
            _capi.check(self._h, _capi.CRNN_ERR_HIP)
        return p

    def forward_device(self, d_gray_ptr, n, w, d_lab_ptr, d_len_ptr, d_raw_ptr=None, widths=None, sync=False):
        """Lines already in HBM: uint8 [n,32,w] at d_gray_ptr -> collapsed labels [n,T] and lengths [n]
        (device pointers).  Asynchronous unless sync; returns the lane the batch was queued on."""
        wv = None
        if widths is not None:
            wv = np.ascontiguousarray(widths, dtype=np.int32)
        rc = self._lib.crnn_forward_device(self._h, d_gray_ptr, n, w, wv.ctypes.data if wv is not None else None,
                                           d_raw_ptr, d_lab_ptr, d_len_ptr, int(sync))
        return _capi.check(self._h, rc)

    def forward_device_noalloc(self, d_gray_ptr, n, w, d_lab_ptr, d_len_ptr, d_raw_ptr=None, widths=None, sync=False):
        """forward_device on the library's fixed workspace: makes no device allocation (for callers under HIP graph
        capture or in loops that must not allocate).  Small batches only (crnn_noalloc_workspace_bytes); the device
        state must exist already (stream_ptr(0) after the weights are loaded brings it up).  include/crnn_hip.h."""
        wv = None
        if widths is not None:
            wv = np.ascontiguousarray(widths, dtype=np.int32)
        rc = self._lib.crnn_forward_device_noalloc(self._h, d_gray_ptr, n, w, wv.ctypes.data if wv is not None else None,
                                                   d_raw_ptr, d_lab_ptr, d_len_ptr, int(sync))
        return _capi.check(self._h, rc)

    def synchronize(self):
        _capi.check(self._h, self._lib.crnn_synchronize(self._h))

    def next_lane(self):
        """Lane the next forward_device call will be queued on (to order a producer stream before it)."""
        return _capi.check(self._h, self._lib.crnn_next_lane(self._h))

    def split_timeouts(self):
        """How often the hidden-split recurrence gave up and a batch was repeated (0 in normal operation)."""
        return int(self._lib.crnn_split_timeouts(self._h))

    def channel_scales(self, layer):
        """The load-time channel equalisation's power-of-two factors of conv `layer` (0..6) / the first embedding (7)."""
        import ctypes as C
        buf = (C.c_float * 512)()
        n = _capi.check(self._h, self._lib.crnn_channel_scales(self._h, layer, buf, 512))
        return np.array(buf[:n], np.float32)

    def texts_from_labels(self, lab, ln):
        """Host arrays lab [n,T] / ln [n] (as copied back from forward_device) -> list of str."""
        lab = np.asarray(lab)
        ln = np.asarray(ln)
        if self._alpha_cp is None:
            return self._texts(lab, ln)
        T = lab.shape[1]
        mask = np.arange(T)[None, :] < ln[:, None]
        s = self._alpha_cp[lab[mask]].astype("<u4").tobytes().decode("utf-32-le")
        offs = np.concatenate(([0], np.cumsum(ln)))
        return [s[offs[i]:offs[i + 1]] for i in range(len(ln))]

    def set_profiling(self, on):
        _capi.check(self._h, self._lib.crnn_set_profiling(self._h, int(on)))

    def kernel_times(self):
        """[(name, ms, flops, bytes)] per launch of the most recent forward_device (profiling on)."""
        cap = 48
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        n = _capi.check(self._h, self._lib.crnn_last_kernel_times(self._h, cap, names, ms, fl, by))
        return [(names[i].decode(), ms[i], fl[i], by[i]) for i in range(n)]

    def timing(self):
        ms = (C.c_float * 5)()
        _capi.check(self._h, self._lib.crnn_last_timing(self._h, ms))
        return dict(zip(("preprocess", "conv", "lstm_linear", "argmax_ctc", "total"), list(ms)))
