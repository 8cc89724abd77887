"""ctypes binding of oracle/libcrnn_oracle.so (the CPU parity checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under ocr_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcrnn_oracle.so")

NSTAGE = 10
_BN = (2, 4, 6)


class _Weights(C.Structure):
    _fp = C.POINTER(C.c_float)
    _fields_ = [
        ("nclass", C.c_int32), ("nh", C.c_int32), ("lstm", C.c_int32), ("leaky", C.c_int32),
        ("conv_w", _fp * 7), ("conv_b", _fp * 7),
        ("bn_w", _fp * 7), ("bn_b", _fp * 7), ("bn_mean", _fp * 7), ("bn_var", _fp * 7),
        ("w_ih", (_fp * 2) * 2), ("w_hh", (_fp * 2) * 2), ("b_ih", (_fp * 2) * 2), ("b_hh", (_fp * 2) * 2),
        ("emb_w", _fp * 2), ("emb_b", _fp * 2),
        ("lin_w", _fp), ("lin_b", _fp),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "crnn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_forward.restype = C.c_int
        L.oracle_forward.argtypes = [C.POINTER(_Weights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_resize_normalize.restype = C.c_int
        L.oracle_resize_normalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p]
        L.oracle_resize_bilinear_u8.restype = C.c_int
        L.oracle_resize_bilinear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int]
        L.oracle_bgr_to_gray.restype = None
        L.oracle_bgr_to_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p]
        L.oracle_resized_width.restype = C.c_int
        L.oracle_resized_width.argtypes = [C.c_int, C.c_int, C.c_int]
        L.oracle_out_T.restype = C.c_int
        L.oracle_out_T.argtypes = [C.c_int]
        L.oracle_argmax.restype = None
        L.oracle_argmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_ctc_collapse.restype = C.c_int
        L.oracle_ctc_collapse.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_normalize_u8.restype = None
        L.oracle_normalize_u8.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    """CPU CRNN with the reference's state-dict key set (numpy float32 tensors)."""

    def __init__(self, state_dict, nclass, nh=256, lstm=True, leaky=False):
        self._keep = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in state_dict.items()
                      if not k.endswith("num_batches_tracked")}
        sd = self._keep
        w = _Weights()
        w.nclass, w.nh, w.lstm, w.leaky = nclass, nh, int(lstm), int(leaky)
        for i in range(7):
            w.conv_w[i] = _p(sd["cnn.conv%d.weight" % i])
            w.conv_b[i] = _p(sd["cnn.conv%d.bias" % i])
            if i in _BN:
                p = "cnn.batchnorm%d." % i
                w.bn_w[i], w.bn_b[i] = _p(sd[p + "weight"]), _p(sd[p + "bias"])
                w.bn_mean[i], w.bn_var[i] = _p(sd[p + "running_mean"]), _p(sd[p + "running_var"])
        if lstm:
            for layer in range(2):
                for d, sfx in enumerate(("", "_reverse")):
                    p = "rnn.%d.rnn." % layer
                    w.w_ih[layer][d] = _p(sd[p + "weight_ih_l0" + sfx])
                    w.w_hh[layer][d] = _p(sd[p + "weight_hh_l0" + sfx])
                    w.b_ih[layer][d] = _p(sd[p + "bias_ih_l0" + sfx])
                    w.b_hh[layer][d] = _p(sd[p + "bias_hh_l0" + sfx])
                w.emb_w[layer] = _p(sd["rnn.%d.embedding.weight" % layer])
                w.emb_b[layer] = _p(sd["rnn.%d.embedding.bias" % layer])
        else:
            w.lin_w, w.lin_b = _p(sd["linear.weight"]), _p(sd["linear.bias"])
        self.w = w
        self.nclass, self.nh, self.lstm = nclass, nh, lstm

    def forward(self, x, stages=False):
        """x float32 [B,1,32,W] (or [B,32,W]) -> logits [T,B,nclass]; optionally the stage taps."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim == 4:
            x = x[:, 0]
        B, H, W = x.shape
        assert H == 32
        T = lib().oracle_out_T(W)
        logits = np.empty((T, B, self.nclass), np.float32)
        taps = None
        arr = None
        if stages:
            w1, w2 = W // 2, W // 2 // 2
            shapes = [(B, 64, 16, w1), (B, 128, 8, w2), (B, 256, 8, w2), (B, 256, 4, w2 + 1),
                      (B, 512, 4, w2 + 1), (B, 512, 2, w2 + 2), (B, 512, 1, w2 + 1),
                      (T, B, self.nh), (T, B, 2 * self.nh), (T, B, 2 * self.nh)]
            taps = [np.empty(s, np.float32) for s in shapes]
            arr = (C.c_void_p * NSTAGE)(*[t.ctypes.data for t in taps])
        rc = lib().oracle_forward(C.byref(self.w), x.ctypes.data, B, W, logits.ctypes.data, arr)
        if rc < 0:
            raise RuntimeError("oracle_forward failed: %d" % rc)
        assert rc == T
        return (logits, taps) if stages else logits


def bgr_to_gray(bgr):
    bgr = np.asarray(bgr)
    assert bgr.dtype == np.uint8 and bgr.ndim == 3 and bgr.shape[2] == 3
    if bgr.strides[2] != 1 or bgr.strides[1] != 3:
        bgr = np.ascontiguousarray(bgr)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().oracle_bgr_to_gray(bgr.ctypes.data, h, w, bgr.strides[0], out.ctypes.data)
    return out


def resized_width(w, h, img_h=32):
    return lib().oracle_resized_width(w, h, img_h)


def resize_u8(gray, dw, dh):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    sh, sw = gray.shape
    out = np.empty((dh, dw), np.uint8)
    rc = lib().oracle_resize_bilinear_u8(gray.ctypes.data, sw, sh, sw, out.ctypes.data, dw, dh)
    if rc:
        raise ValueError("height and width must be > 0")
    return out


def resize_normalize(gray, img_h=32):
    """util.py:6-13 on a uint8 [H,W] array -> float32 [32,w]."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    sh, sw = gray.shape
    if sh == 0:
        raise ZeroDivisionError("float division by zero")
    w = resized_width(sw, sh, img_h)
    if w <= 0:
        raise ValueError("height and width must be > 0")
    out = np.empty((img_h, w), np.float32)
    rc = lib().oracle_resize_normalize(gray.ctypes.data, sw, sh, sw, img_h, out.ctypes.data)
    assert rc == w, rc
    return out


def argmax(logits, margin=False):
    """logits [T,B,nclass] -> idx int32 [B,T] (+ top1-top2 margins)."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    T, B, K = logits.shape
    idx = np.empty((B, T), np.int32)
    m = np.empty((B, T), np.float32) if margin else None
    lib().oracle_argmax(logits.ctypes.data, T, B, K, idx.ctypes.data, m.ctypes.data if margin else None)
    return (idx, m) if margin else idx


def ctc_collapse(idx):
    """raw class indices [T] -> alphabet positions (class-1) after greedy collapse."""
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(len(idx), np.int32)
    n = lib().oracle_ctc_collapse(idx.ctypes.data, len(idx), out.ctypes.data)
    return out[:n]


def decode(idx, alphabet):
    return "".join(alphabet[k] for k in ctc_collapse(idx))


def predict_gray(oracle, gray, alphabet):
    """CRNN.predict (network_torch.py:130-147) on a uint8 gray [H,W] array."""
    x = resize_normalize(gray)
    if x.shape[-1] < 8:
        return ""
    logits = oracle.forward(x[None])
    return decode(argmax(logits)[0], alphabet)
