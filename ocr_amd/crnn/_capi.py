"""ctypes binding of ocr_amd/libcrnn_hip.so (include/crnn_hip.h).

The library is the product; there is no Python/CPU compute fallback.  If the
shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libcrnn_hip.so")

CRNN_OK = 0
CRNN_ERR_ARG, CRNN_ERR_HIP, CRNN_ERR_STATE, CRNN_ERR_ZERO_H, CRNN_ERR_ZERO_W, CRNN_ERR_NOMEM = -1, -2, -3, -4, -5, -6
PAD_PER_LINE, PAD_REFERENCE = 0, 1
FP32, BF16 = 0, 1

SYMBOLS = ["crnn_create", "crnn_load_weights", "crnn_resized_width", "crnn_out_len", "crnn_preprocess",
           "crnn_predict_batch", "crnn_set_page", "crnn_predict_rects", "crnn_predict_page", "crnn_forward_device", "crnn_debug_stage", "crnn_synchronize", "crnn_stream", "crnn_lanes",
           "crnn_last_timing", "crnn_set_profiling", "crnn_last_kernel_times", "crnn_last_error", "crnn_destroy",
           "crnn_next_lane", "crnn_split_timeouts", "crnn_forward_device_noalloc", "crnn_noalloc_workspace_bytes",
           "crnn_channel_scales"]


class Config(C.Structure):
    _fields_ = [("img_h", C.c_int32), ("nc", C.c_int32), ("nclass", C.c_int32), ("nh", C.c_int32),
                ("leaky_relu", C.c_int32), ("lstm", C.c_int32), ("device", C.c_int32), ("lanes", C.c_int32),
                ("precision", C.c_int32), ("reserved", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


_lib = None


def _more_hw_queues():
    """A handle's lanes are HIP streams; the runtime multiplexes a process's streams over
    GPU_MAX_HW_QUEUES hardware queues (HIP's default, 4), and torch's own streams take some of those.
    With 4 lanes on 4 queues the step is SLOWER than with 3 (lanes sharing a queue serialise); with 8
    queues 4 lanes are worth +7 %.  Raises the setting to 8 when it is unset or lower (never lowers it);
    CRNN_KEEP_HW_QUEUES=1 leaves it alone.  Only effective before the HIP runtime initialises: if torch
    has already brought HIP up in this process the variable is left as it is and the number of queues
    the runtime really has goes into CRNN_EFFECTIVE_HW_QUEUES, which makes crnn_create keep a handle at
    three lanes (the fastest choice on four queues).  Deployments should export GPU_MAX_HW_QUEUES=8
    themselves; this is the fallback for callers that did not."""
    try:
        cur = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
    except ValueError:
        cur = 0
    if os.environ.get("CRNN_KEEP_HW_QUEUES") or cur >= 8:
        return
    if _hip_already_up():
        os.environ.setdefault("CRNN_EFFECTIVE_HW_QUEUES", str(cur or 4))
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"


def _hip_already_up():
    """True when the HIP runtime of this process has (or may have) read GPU_MAX_HW_QUEUES already: torch has
    initialised its device context, or a profiler tool library is loaded -- under `rocprofv3 --pmc -- python3 ...` the
    tool brings HIP up before Python starts, so a later change of the variable would be recorded but not in effect."""
    import sys
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_initialized():
            return True
    except Exception:  # noqa: BLE001
        pass
    if any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return True
    if "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return True
    return False


def effective_hw_queues():
    """Hardware queues the HIP runtime of this process has (as far as this module can tell)."""
    try:
        return int(os.environ.get("CRNN_EFFECTIVE_HW_QUEUES") or os.environ.get("GPU_MAX_HW_QUEUES") or 4)
    except ValueError:
        return 4


_more_hw_queues()   # at import: the earlier the better (must precede HIP initialisation)


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and load it
    by its unversioned name, so it does not unify with the /opt/rocm copy this library links to; two
    HIP/HSA runtimes in one process cannot both open the GPU.  If torch is installed (it is the
    tensor plumbing of bench.py and of callers that hand over device pointers), load ITS runtime
    first: this library's libamdhip64.so.7 dependency then resolves to the same object (same
    SONAME), whichever of the two is imported first.  CRNN_HIP_RUNTIME=<path> overrides."""
    path = os.environ.get("CRNN_HIP_RUNTIME")
    if not path:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
                if os.path.exists(cand):
                    path = cand
        except Exception:  # noqa: BLE001
            path = None
    if path:
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ocr_amd/libcrnn_hip.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ocr_amd/csrc` (hipcc, gfx950).  There is no CPU fallback.")
    _more_hw_queues()
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_int32)
    L.crnn_create.restype = C.c_int
    L.crnn_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.crnn_load_weights.restype = C.c_int
    L.crnn_load_weights.argtypes = [vp, C.POINTER(Tensor), i32]
    L.crnn_resized_width.restype = C.c_int
    L.crnn_resized_width.argtypes = [vp, i32, i32]
    L.crnn_out_len.restype = C.c_int
    L.crnn_out_len.argtypes = [i32]
    L.crnn_preprocess.restype = C.c_int
    L.crnn_preprocess.argtypes = [vp, C.POINTER(vp), i32p, i32p, i32p, i32p, i32, i32, vp, i32p]
    L.crnn_predict_batch.restype = C.c_int
    L.crnn_predict_batch.argtypes = [vp, C.POINTER(vp), i32p, i32p, i32p, i32p, i32, i32, i32, vp, vp, vp, vp, vp]
    L.crnn_set_page.restype = C.c_int
    L.crnn_set_page.argtypes = [vp, vp, i32, i32, i32, i32]
    L.crnn_predict_rects.restype = C.c_int
    L.crnn_predict_rects.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    L.crnn_predict_page.restype = C.c_int
    L.crnn_predict_page.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    L.crnn_forward_device.restype = C.c_int
    L.crnn_forward_device.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32]
    L.crnn_debug_stage.restype = C.c_int
    L.crnn_debug_stage.argtypes = [vp, vp, i32, i32, vp, i32, vp, C.c_int64]
    L.crnn_synchronize.restype = C.c_int
    L.crnn_synchronize.argtypes = [vp]
    L.crnn_stream.restype = vp
    L.crnn_stream.argtypes = [vp, i32]
    L.crnn_lanes.restype = C.c_int
    L.crnn_lanes.argtypes = [vp]
    L.crnn_last_timing.restype = C.c_int
    L.crnn_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.crnn_set_profiling.restype = C.c_int
    L.crnn_set_profiling.argtypes = [vp, i32]
    L.crnn_last_kernel_times.restype = C.c_int
    L.crnn_last_kernel_times.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.crnn_last_error.restype = C.c_char_p
    L.crnn_last_error.argtypes = [vp]
    L.crnn_destroy.restype = None
    L.crnn_destroy.argtypes = [vp]
    L.crnn_next_lane.restype = C.c_int
    L.crnn_next_lane.argtypes = [vp]
    L.crnn_split_timeouts.restype = C.c_int64
    L.crnn_split_timeouts.argtypes = [vp]
    L.crnn_forward_device_noalloc.restype = C.c_int
    L.crnn_forward_device_noalloc.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32]
    L.crnn_noalloc_workspace_bytes.restype = C.c_int64
    L.crnn_noalloc_workspace_bytes.argtypes = [vp, i32, i32]
    L.crnn_channel_scales.restype = C.c_int
    L.crnn_channel_scales.argtypes = [vp, i32, C.POINTER(C.c_float), i32]
    _lib = L
    return L


class CrnnError(RuntimeError):
    pass


def check(handle, rc):
    """Map a crnn_status to the exception the reference would raise at that point."""
    if rc >= 0:
        return rc
    msg = lib().crnn_last_error(handle).decode("utf-8", "replace") if handle else "crnn error"
    if rc == CRNN_ERR_ZERO_H:
        raise ZeroDivisionError("float division by zero")  # util.py:8 on a zero-height image
    if rc == CRNN_ERR_ZERO_W:
        raise ValueError("height and width must be > 0")  # Pillow's resize, util.py:10
    if rc == CRNN_ERR_ARG:
        raise ValueError(msg)
    if rc == CRNN_ERR_NOMEM:
        raise MemoryError(msg)
    raise CrnnError("%s (status %d)" % (msg, rc))


class ImageList:
    """uint8 numpy images (HxW gray or HxWx3 BGR, rows may be strided) as the ABI's parallel arrays."""

    def __init__(self, images):
        self.keep = []
        n = len(images)
        self.n = n
        self.ptrs = (C.c_void_p * n)()
        self.h = (C.c_int32 * n)()
        self.w = (C.c_int32 * n)()
        self.stride = (C.c_int32 * n)()
        self.ch = (C.c_int32 * n)()
        for i, a in enumerate(images):
            a = np.asarray(a)
            if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
                raise ValueError("image %d: expected uint8 HxW (gray) or HxWx3 (BGR), got %s %s" % (i, a.dtype, a.shape))
            ch = 1 if a.ndim == 2 else 3
            ok = a.size == 0 or (a.strides[1] == ch and (ch == 1 or a.strides[2] == 1) and a.strides[0] >= a.shape[1] * ch)
            if not ok:
                a = np.ascontiguousarray(a)
            self.keep.append(a)
            self.ptrs[i] = a.ctypes.data
            self.h[i], self.w[i], self.ch[i] = a.shape[0], a.shape[1], ch
            self.stride[i] = a.strides[0] if a.size else a.shape[1] * ch
