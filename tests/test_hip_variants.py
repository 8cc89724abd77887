"""GPU: every selectable kernel variant of the Winograd stage gives the reference's strings.

The variants are chosen by environment variables the library reads once per process
(CRNN_WINOGRAD_F4 / CRNN_WINOGRAD masks, CRNN_WINO_FUSED, CRNN_WINO_DMA / CRNN_WINO4_SHAPE tile shapes), so each runs in a child
process; the child prints the decoded strings of the synthetic 32x280 golden lines and of the
variable-width English-head lines, and the parent compares them with the goldens captured from the
reference (tests/golden/make_golden.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, numpy as np
sys.path.insert(0, %r)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
G = np.load(%r)
M = json.load(open(%r))
zh = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=int(sys.argv[1]))
zh.load_state_dict(synth_state_dict(5530, seed=0))
en = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=int(sys.argv[1]))
en.load_state_dict(synth_state_dict(65, seed=0))
out = {"text": zh.predict_lines(list(G["syn280/text/gray"])),
       "noise": zh.predict_lines(list(G["syn280/noise/gray"])),
       "var": en.predict_lines([G["var/%%d/gray" %% w] for w in M["var_widths"]])}
print("RESULT " + json.dumps(out))
"""

F2 = {"CRNN_WINOGRAD_F4": "0"}   # F(4x4) off: conv1-5 all go through the F(2x2) / direct variants below
VARIANTS = [
    ({**F2, "CRNN_WINOGRAD": "0"}, 1),                        # direct implicit GEMM everywhere
    ({**F2, "CRNN_WINO_FUSED": "0"}, 1),                      # three-kernel F(2x2) (batched GEMM + output transform)
    ({**F2, "CRNN_WINO_DMA": "0"}, 1),                        # fused F(2x2), register-staged operands
    ({**F2, "CRNN_WINO_DMA": "1"}, 1),                        # fused F(2x2), direct-to-LDS, 128x64 tiles
    ({**F2, "CRNN_WINO_DMA": "2"}, 1),                        # 128x128 tiles
    ({**F2, "CRNN_WINO_DMA": "3"}, 1),                        # 64x64 tiles
    (F2, 4),                                                  # F(2x2) per-launch tile choice, lanes sharing the device
    ({**F2, "CRNN_WINOGRAD": "0x38"}, 2),                     # F(2x2) on conv3-5 only
    ({}, 1),                                                  # default: F(4x4) on conv1-5 (pool1-3 in output transforms), F(4,2) on conv6
    ({}, 4),
    ({"CRNN_WINOGRAD_F4": "0x3C"}, 2),                        # conv1 on the fused F(2x2) kernel
    ({"CRNN_WINO4_SHAPE": "2"}, 1),                           # F(4x4) GEMM units of 128x64
    ({"CRNN_WINOGRAD_F4": "0x30", "CRNN_WINOGRAD": "0x08"}, 1),   # F(4x4) conv4-5, F(2x2) conv3, direct conv1-2
    ({"CRNN_WINO4_LW": "0"}, 1),                              # F(4x4) GEMM without the loader wave (every wave loads)
    ({"CRNN_WINO4_POOL21": "0"}, 1),                          # pool2 / pool3 as separate kernels
    ({"CRNN_WINO6": "0"}, 1),                                 # conv6 as implicit GEMM instead of 1-D F(4,2)
    ({"CRNN_CLS_FILTER": "0"}, 1),                            # classifier arg-max from the plain fp32 GEMM
    ({"CRNN_CLS_FILTER": "0", "CRNN_LIN_GEMM": "0"}, 1),      # ... on conv_gemm_kernel
    ({"CRNN_LIN_GEMM": "0"}, 1),                              # input projections / classifier on conv_gemm_kernel
    ({"CRNN_F16X3": "0"}, 1),                                 # Winograd GEMMs on the f32 MFMA instead of the f16x3 form
    ({"CRNN_F16X3": "0"}, 4),
    ({"CRNN_SPLIT_H3": "0"}, 1),                              # hidden-split recurrence on the f32 MFMA
    ({"CRNN_GEMM16_DMA": "0"}, 1),                            # classifier pre-filter GEMM on conv_gemm_bf16_kernel
    ({"CRNN_GEMM16_DMA": "0", "CRNN_CLS_F16": "0"}, 2),
    ({"CRNN_CONV0_FUSED": "0"}, 1),                           # conv0 + pool0 and conv1's input transform as two kernels
    ({"CRNN_WINO4_FUSE": "0"}, 1),                            # conv1 / conv2 as GEMM + output transform (product tensor written)
    ({"CRNN_WINO4_FUSE": "0x3e"}, 1),                         # every F(4x4) layer on the fused GEMM + output transform kernel
    ({"CRNN_WINO4_FUSE": "0x3e", "CRNN_WINO4_POOL21": "0"}, 2),   # ... conv3 / conv5 too (pool2 / pool3 as separate kernels)
    ({"CRNN_CLS_F16": "0"}, 1),                               # classifier pre-filter in bf16 instead of fp16
    ({"CRNN_H3_LW": "0"}, 1),                                 # f16x3 Winograd GEMMs without the loader waves
    ({"CRNN_H3_LW": "64"}, 2),                                # ... with loader waves on every layer
    ({"CRNN_REC_H3": "0"}, 1),                                # recurrence on the f32 MFMA
    ({"CRNN_LSTM_SPLIT": "0"}, 1),                            # host batches on the one-workgroup (f16x3) recurrence
]


@pytest.mark.gpu
@pytest.mark.parametrize("env,lanes", VARIANTS, ids=lambda v: ",".join("%s=%s" % kv for kv in v.items()) if isinstance(v, dict) else "lanes%d" % v)
def test_variant_strings_match_reference_goldens(env, lanes, golden):
    _, M = golden
    e = dict(os.environ)
    for k in ("CRNN_WINOGRAD", "CRNN_WINO_FUSED", "CRNN_WINO_DMA", "CRNN_WINOGRAD_F4", "CRNN_WINO4_SHAPE", "CRNN_WINO4_MIN_N", "CRNN_LIN_GEMM", "CRNN_WINO4_LW", "CRNN_WINO4_POOL21", "CRNN_WINO6", "CRNN_CLS_FILTER", "CRNN_F16X3", "CRNN_REC_H3", "CRNN_LSTM_SPLIT", "CRNN_H3_LW", "CRNN_CLS_F16", "CRNN_WINO4_FUSE", "CRNN_CONV0_FUSED", "CRNN_GEMM16_DMA", "CRNN_SPLIT_H3"):
        e.pop(k, None)
    e.update(env)
    code = CHILD % (ROOT, os.path.join(ROOT, "tests", "golden", "crnn_golden.npz"),
                    os.path.join(ROOT, "tests", "golden", "crnn_golden.json"))
    r = subprocess.run([sys.executable, "-c", code, str(lanes)], env=e, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["text"] == M["syn280_text"]
    assert out["noise"] == M["syn280_noise"]
    assert out["var"] == M["var_en"]
