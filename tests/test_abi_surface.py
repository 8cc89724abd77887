"""The C-ABI library loads and exports every symbol include/crnn_hip.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

from ocr_amd.crnn import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "crnn_hip.h")).read()
    declared = sorted(set(re.findall(r"CRNN_API [^;(]*?\b(crnn_[a-z_]+)\(", hdr)))
    assert declared and sorted(_capi.SYMBOLS) == declared
    lib = _capi.lib()
    for s in declared:
        assert getattr(lib, s) is not None


def test_host_only_calls_and_errors():
    lib = _capi.lib()
    assert lib.crnn_out_len(280) == 71 and lib.crnn_out_len(1024) == 257 and lib.crnn_out_len(8) == 3
    h = C.c_void_p()
    bad = _capi.Config(48, 1, 65, 256, 0, 1, 0, 0, 0, 0)
    assert lib.crnn_create(C.byref(bad), C.byref(h)) == _capi.CRNN_ERR_ARG
    cfg = _capi.Config(32, 1, 65, 256, 0, 1, 0, 0, 0, 0)
    assert lib.crnn_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.crnn_resized_width(h, 537, 91) == 188   # invoice_code.png: int(537 / (91 / 32.0))
    assert lib.crnn_resized_width(h, 10, 0) == _capi.CRNN_ERR_ZERO_H
    with pytest.raises(ZeroDivisionError):
        _capi.check(h, lib.crnn_resized_width(h, 10, 0))
    # weights: a missing tensor is reported by name
    import numpy as np
    t = (_capi.Tensor * 1)()
    a = np.zeros((64, 1, 3, 3), np.float32)
    t[0].name = b"module.cnn.conv0.weight"
    t[0].data = a.ctypes.data_as(C.POINTER(C.c_float))
    t[0].ndim = 4
    for i, s in enumerate(a.shape):
        t[0].shape[i] = s
    assert lib.crnn_load_weights(h, t, 1) == _capi.CRNN_ERR_ARG
    assert b"cnn.conv0.bias" in lib.crnn_last_error(h)
    lib.crnn_destroy(h)


def test_python_class_mirrors_reference_signature():
    import inspect
    from ocr_amd.crnn.network_hip import CRNN
    sig = inspect.signature(CRNN.__init__)
    assert list(sig.parameters)[:9] == ["self", "imgH", "nc", "nclass", "nh", "leakyRelu", "lstmFlag", "GPU", "alphabet"]
    for name in ("load_weights", "ocr", "predict", "predict_job", "predict_batch"):
        assert callable(getattr(CRNN, name))
    assert list(inspect.signature(CRNN.predict_batch).parameters) == ["self", "boxes", "batch_size"]


def test_load_weights_pth_host_side(tmp_path, synth_en):
    """The .pth reader and the weight repack are host code: they run without a GPU (the device is
    only touched on the first compute call)."""
    import numpy as np
    import torch
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    path = str(tmp_path / "w.pth")
    torch.save({"module." + k: torch.from_numpy(np.asarray(v)) for k, v in synth_en.items()}, path)
    m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
    m.load_weights(path)
    bad = dict(synth_en)
    bad.pop("cnn.conv3.weight")
    with pytest.raises(Exception):
        CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish).load_state_dict(bad)
