"""GPU: the chained transform kernel (wino4_out_in_h3_kernel: a layer's Winograd output transform and the next
layer's input transform in one launch: conv3 -> pool2 -> conv4, conv4 -> conv5, conv5 -> pool3 -> conv6) against
the kernels it replaces.

The chained kernel performs the same operations on the same values, so everything downstream must be
BIT-identical with it on (default) and off (CRNN_WINO4_CHAIN=0, read per call).  It only runs from 128 lines
up, so these batches are larger than the ones the stage-by-stage oracle tests use; the 256-line tests of
tests/test_hip_scale.py pin the same path to the reference's strings."""
import os

import numpy as np
import pytest

from ocr_amd.crnn import keys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zh(synth_zh):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, lanes=1)
    m.load_state_dict(synth_zh)
    return m


def _with_chain(value, fn):
    old = os.environ.pop("CRNN_WINO4_CHAIN", None)
    if value is not None:
        os.environ["CRNN_WINO4_CHAIN"] = value
    try:
        return fn()
    finally:
        os.environ.pop("CRNN_WINO4_CHAIN", None)
        if old is not None:
            os.environ["CRNN_WINO4_CHAIN"] = old


# (lines, width, ragged): 280 -> 18 tiles per row in conv3 and conv4 (two tile rows / one); 288 -> conv3's width
# is a multiple of 4, so pool2's last column lies past conv3's last tile (19 tiles in conv4); 200 with ragged
# valid widths -> the width masks of both layers; 132 -> an odd tile count and a short last segment
# 512 lines: twice the headline batch, tensors near the 32-bit offset limit of the kernel's buffer addressing
# 16 / 24 / 40: one or two tiles per row -- segments shorter than the segment count, the first tile also the last
CASES = [(256, 280, False), (128, 288, False), (130, 200, True), (160, 132, True), (512, 280, False), (128, 16, False),
         (136, 24, True), (128, 40, True)]


@pytest.mark.parametrize("n,w,ragged", CASES)
def test_chained_transforms_are_bit_identical_to_the_separate_kernels(zh, n, w, ragged):
    rs = np.random.RandomState(n + w)
    x = (rs.randint(0, 256, size=(n, 32, w)).astype(np.float32) / 255.0 - 0.5) / 0.5
    widths = None
    if ragged:
        widths = rs.randint(max(8, w // 3), w + 1, size=n).astype(np.int32)
        widths[0] = w
        for i in range(n):
            x[i, :, widths[i]:] = 0.0
    for stage in (5, 6, 10):
        off = _with_chain("0", lambda: zh.debug_stage(x, stage, widths))
        on = _with_chain(None, lambda: zh.debug_stage(x, stage, widths))
        assert np.isfinite(off).all()
        assert np.array_equal(off, on), "stage %d differs with all chained kernels" % stage
        for mask, what in (("0x8", "conv3 -> conv4"), ("0x10", "conv4 -> conv5"), ("0x20", "conv5 -> conv6")):
            one = _with_chain(mask, lambda: zh.debug_stage(x, stage, widths))
            assert np.array_equal(off, one), "stage %d differs with %s chained" % (stage, what)


def test_leaky_relu_and_dense_head_variants(synth_en):
    """LeakyReLU(0.2) is a compile-time variant of the chained kernels; the dense head (lstmFlag=False) shares the conv
    stack: both bit-identical with the kernels on and off, ragged widths."""
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict
    rs = np.random.RandomState(11)
    n, w = 144, 240
    x = (rs.randint(0, 256, size=(n, 32, w)).astype(np.float32) / 255.0 - 0.5) / 0.5
    widths = rs.randint(60, w + 1, size=n).astype(np.int32)
    for i in range(n):
        x[i, :, widths[i]:] = 0.0
    leaky = CRNN(32, 1, 65, 256, leakyRelu=True, lstmFlag=True, GPU=True, alphabet=keys.alphabetEnglish, lanes=1)
    leaky.load_state_dict(synth_en)
    dense = CRNN(32, 1, 65, 256, leakyRelu=False, lstmFlag=False, GPU=True, alphabet=keys.alphabetEnglish, lanes=1)
    dense.load_state_dict(synth_state_dict(65, seed=0, lstm=False))
    for m in (leaky, dense):
        off = _with_chain("0", lambda: m.debug_stage(x, 10, widths))
        on = _with_chain(None, lambda: m.debug_stage(x, 10, widths))
        assert np.isfinite(off).all() and np.abs(off).max() > 0
        assert np.array_equal(off, on)


def test_stage_requests_get_the_stage_written(zh):
    """crnn_debug_stage(3) / (4) / (5) ask for exactly the activations the chained kernels never write: the
    request turns the one kernel in question off and the stage comes back as with CRNN_WINO4_CHAIN=0."""
    rs = np.random.RandomState(3)
    x = (rs.randint(0, 256, size=(128, 32, 280)).astype(np.float32) / 255.0 - 0.5) / 0.5
    for stage in (3, 4, 5):
        off = _with_chain("0", lambda: zh.debug_stage(x, stage))
        on = _with_chain(None, lambda: zh.debug_stage(x, stage))
        assert np.abs(off).max() > 0
        assert np.array_equal(off, on)


def test_chained_kernels_run_in_the_default_path(zh):
    """The per-launch names of a 256-line batch show the chained kernels (and not the ones they replace)."""
    import torch
    dev = torch.device("cuda", 0)
    lines = torch.randint(0, 256, (256, 32, 280), dtype=torch.uint8, device=dev)
    T = 71
    lab = torch.zeros((256, T), dtype=torch.int32, device=dev)
    ln = torch.zeros((256,), dtype=torch.int32, device=dev)
    zh.set_profiling(True)
    try:
        zh.forward_device(lines.data_ptr(), 256, 280, lab.data_ptr(), ln.data_ptr())
        zh.synchronize()
        names = [k[0] for k in zh.kernel_times()]
    finally:
        zh.set_profiling(False)
    assert "conv3_wout4_pool2_win4" in names and "conv4_wout4_win4" in names and "conv5_wout4_pool3_win4" in names
    for gone in ("conv4_win4", "conv5_win4", "conv6_win4", "conv3_wout4_pool2", "conv4_wout4", "conv5_wout4_pool3"):
        assert gone not in names


def test_conv0_on_the_mfma_against_its_scalar_form(synth_zh):
    """conv0 inside conv0_win4_h3_kernel runs on the fp16 MFMA in the f16x3 form (CRNN_CONV0_MFMA, read at load);
    its scalar fp32 form is one environment variable away.  Stage 1 (conv1 + pool1, the first tensor downstream
    of the fused kernel) must agree to fp32 rounding and the logits to the 1e-3 budget (two fp32-accurate evaluations
    of a network that amplifies rounding ~1000x differ by a few 1e-4), arg-max sequences equal away from
    near-ties -- ragged widths included (the kernel masks pooled columns)."""
    import subprocess
    import sys
    import json
    child = r"""
import json, sys, os, numpy as np
sys.path.insert(0, %r)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
rs = np.random.RandomState(21)
n, w = 96, 280
x = (rs.randint(0, 256, size=(n, 32, w)).astype(np.float32) / 255.0 - 0.5) / 0.5
widths = rs.randint(40, w + 1, size=n).astype(np.int32); widths[0] = w
for i in range(n): x[i, :, widths[i]:] = 0.0
m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, lanes=1)
m.load_state_dict(synth_state_dict(5530, seed=0))
s1 = m.debug_stage(x, 1, widths); lg = m.debug_stage(x, 10, widths)
np.savez(sys.argv[1], s1=s1, lg=lg)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, val in (("mfma", None), ("scalar", "0")):
            e = dict(os.environ)
            e.pop("CRNN_CONV0_MFMA", None)
            if val is not None:
                e["CRNN_CONV0_MFMA"] = val
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", child, f], env=e, capture_output=True, text=True, timeout=240)
            assert r.returncode == 0, r.stderr[-2000:]
            out[tag] = dict(np.load(f))
    a, b = out["mfma"], out["scalar"]
    scale = np.abs(b["s1"]).max()
    err1 = np.abs(a["s1"] - b["s1"]).max() / scale
    errl = np.abs(a["lg"] - b["lg"]).max()
    print("conv0 mfma vs scalar: stage 1 rel %.2e, logits abs %.2e" % (err1, errl))
    assert err1 < 2e-6
    assert errl < 1e-3
    top2 = np.sort(b["lg"], axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2e-3          # rows whose winner is not a near-tie
    assert clear.mean() > 0.99
    assert np.array_equal(a["lg"].argmax(-1)[clear], b["lg"].argmax(-1)[clear])


def test_plain_gemms_on_the_ping_pong_kernel(zh):
    """From whole 256-row units (n T a multiple of 256: 256 lines of width 280) the LSTM input projections and the first
    embedding run on wino4_gemm_h3_pp_kernel with a scale-and-bias epilogue (CRNN_LIN_PP, read per call).  Same operands,
    same products, a different summation order inside the MFMA chains: stages 7 (embedding), 9 and the logits agree
    with lin_gemm_h3_kernel's to fp32 rounding / the usual amplification."""
    rs = np.random.RandomState(5)
    x = (rs.randint(0, 256, size=(256, 32, 280)).astype(np.float32) / 255.0 - 0.5) / 0.5
    def both(stage):
        os.environ["CRNN_LIN_PP"] = "0"
        try:
            a = zh.debug_stage(x, stage)
        finally:
            os.environ.pop("CRNN_LIN_PP", None)
        return a, zh.debug_stage(x, stage)
    a, b = both(7)
    assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
    a, b = both(10)
    assert np.abs(a - b).max() < 1e-3
    top2 = np.sort(a, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2e-3
    assert clear.mean() > 0.99 and np.array_equal(a.argmax(-1)[clear], b.argmax(-1)[clear])
    import torch
    dev = torch.device("cuda", 0)
    lines = torch.randint(0, 256, (256, 32, 280), dtype=torch.uint8, device=dev)
    lab = torch.zeros((256, 71), dtype=torch.int32, device=dev)
    ln = torch.zeros((256,), dtype=torch.int32, device=dev)
    zh.set_profiling(True)
    try:
        zh.forward_device(lines.data_ptr(), 256, 280, lab.data_ptr(), ln.data_ptr())
        zh.synchronize()
        names = [k[0] for k in zh.kernel_times()]
    finally:
        zh.set_profiling(False)
    assert "lstm0_ih_pp" in names and "emb0_pp" in names and "lstm1_ih_pp" in names


@pytest.mark.parametrize("n,w,ragged", [(256, 280, False), (100, 280, True), (7, 64, True), (130, 200, True)])
def test_conv6_output_transform_writing_the_lstm_operand(zh, n, w, ragged):
    """conv6's output transform writes the f16x3 rows and 32-row scales of the first LSTM projection itself
    (wino42_out_split_kernel; CRNN_CONV6_SPLIT=0: wino42_out_kernel + rows_to_f16x3_kernel).  Same expressions on the
    same values: the embedding (stage 7), the last recurrence (9) and the logits are bit-identical, and a request for
    the fp32 sequence tensor (stage 6) still gets it."""
    rs = np.random.RandomState(n * 3 + w)
    x = (rs.randint(0, 256, size=(n, 32, w)).astype(np.float32) / 255.0 - 0.5) / 0.5
    widths = None
    if ragged:
        widths = rs.randint(max(8, w // 3), w + 1, size=n).astype(np.int32)
        widths[0] = w
        for i in range(n):
            x[i, :, widths[i]:] = 0.0
    def off(stage):
        os.environ["CRNN_CONV6_SPLIT"] = "0"
        try:
            return zh.debug_stage(x, stage, widths)
        finally:
            os.environ.pop("CRNN_CONV6_SPLIT", None)
    for stage in (6, 7, 9, 10):
        a, b = off(stage), zh.debug_stage(x, stage, widths)
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert np.array_equal(a, b), "stage %d" % stage
