"""GPU: every selectable kernel variant of the Winograd stage gives the reference's strings.

The variants are chosen by environment variables the library reads once per process
(CRNN_WINOGRAD_F4 mask, CRNN_WINO4_FUSE, CRNN_F16X3, CRNN_GEMM_PP, ...), so each runs in a child
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

VARIANTS = [
    ({"CRNN_WINOGRAD_F4": "0"}, 1),                           # F(4x4) off: conv1-5 as direct implicit GEMM (conv_gemm_kernel)
    ({}, 1),                                                  # default: F(4x4) on conv1-5 (pool1-3 in output transforms), F(4,2) on conv6
    ({}, 4),
    ({"CRNN_WINOGRAD_F4": "0x30"}, 2),                        # F(4x4) conv4-5 only, direct conv1-3
    ({"CRNN_WINO4_POOL21": "0"}, 1),                          # pool2 / pool3 as separate kernels
    ({"CRNN_WINO6": "0"}, 1),                                 # conv6 as implicit GEMM instead of 1-D F(4,2)
    ({"CRNN_CLS_FILTER": "0"}, 1),                            # classifier arg-max from the plain fp32 GEMM
    ({"CRNN_CLS_FILTER": "0", "CRNN_LIN_GEMM": "0"}, 1),      # ... on conv_gemm_kernel
    ({"CRNN_LIN_GEMM": "0"}, 1),                              # input projections / classifier on conv_gemm_kernel
    ({"CRNN_F16X3": "0"}, 1),                                 # every GEMM and the recurrence on the f32 MFMA instead of the f16x3 form
    ({"CRNN_F16X3": "0"}, 4),
    ({"CRNN_GEMM16_DMA": "0"}, 1),                            # classifier pre-filter GEMM on conv_gemm_bf16_kernel
    ({"CRNN_GEMM16_DMA": "0", "CRNN_CLS_F16": "0"}, 2),
    ({"CRNN_CONV0_FUSED": "0"}, 1),                           # conv0 + pool0 and conv1's input transform as two kernels
    ({"CRNN_WINO4_FUSE": "0"}, 1),                            # conv1 / conv2 as GEMM + output transform (product tensor written)
    ({"CRNN_WINO4_FUSE": "0x3e"}, 1),                         # every F(4x4) layer on the fused GEMM + output transform kernel
    ({"CRNN_WINO4_FUSE": "0x3e", "CRNN_WINO4_POOL21": "0"}, 2),   # ... conv3 / conv5 too (pool2 / pool3 as separate kernels)
    ({"CRNN_CLS_F16": "0"}, 1),                               # classifier pre-filter in bf16 instead of fp16
    ({"CRNN_REC_H3": "0"}, 1),                                # recurrence on the f32 MFMA (one-workgroup kernel; no latency form)
    ({"CRNN_LSTM_SPLIT": "0"}, 1),                            # host batches on the one-workgroup (f16x3) recurrence
    ({"CRNN_LSTM_SPLIT": "0", "CRNN_REC_RES": "0"}, 1),       # ... streaming all of W_hh (no resident quarter)
    ({"CRNN_GEMM_PP": "0"}, 1),                               # conv4 / conv5 GEMMs on the loader-wave form instead of the ping-pong form
    ({"CRNN_GEMM_PP": "2", "CRNN_WINO4_FUSE": "0"}, 2),       # the ping-pong form on every layer it can take
    ({"CRNN_EQUALIZE": "0"}, 1),                              # no load-time channel equalisation
    ({"CRNN_LIN_PP": "0"}, 1),                                # LSTM input projections / first embedding on lin_gemm_h3_kernel at every size
    ({"CRNN_CONV0_MFMA": "0"}, 1),                            # conv0 (inside the fused conv0 / conv1-input kernel) as scalar fp32 FMAs
]


@pytest.mark.gpu
@pytest.mark.parametrize("env,lanes", VARIANTS, ids=lambda v: ",".join("%s=%s" % kv for kv in v.items()) if isinstance(v, dict) else "lanes%d" % v)
def test_variant_strings_match_reference_goldens(env, lanes, golden):
    _, M = golden
    e = dict(os.environ)
    for k in ("CRNN_WINOGRAD_F4", "CRNN_WINO4_MIN_N", "CRNN_LIN_GEMM", "CRNN_WINO4_POOL21", "CRNN_WINO6", "CRNN_CLS_FILTER", "CRNN_F16X3", "CRNN_REC_H3", "CRNN_LSTM_SPLIT", "CRNN_CLS_F16", "CRNN_WINO4_FUSE", "CRNN_CONV0_FUSED", "CRNN_GEMM16_DMA", "CRNN_REC_RES", "CRNN_GEMM_PP", "CRNN_EQUALIZE", "CRNN_CONV0_MFMA", "CRNN_WINO4_CHAIN", "CRNN_LIN_PP"):
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
