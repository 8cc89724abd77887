"""GPU: every selectable kernel variant of the Winograd stage gives the reference's strings.

The variants are chosen by environment variables the library reads once per process
(CRNN_WINOGRAD mask, CRNN_WINO_FUSED, CRNN_WINO_DMA tile shape / staging), so each runs in a child
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
    ({"CRNN_WINOGRAD": "0"}, 1),                              # direct implicit GEMM everywhere
    ({"CRNN_WINO_FUSED": "0"}, 1),                            # three-kernel Winograd (batched GEMM + output transform)
    ({"CRNN_WINO_DMA": "0"}, 1),                              # fused, register-staged operands
    ({"CRNN_WINO_DMA": "1"}, 1),                              # fused, direct-to-LDS, 128x64 tiles
    ({"CRNN_WINO_DMA": "2"}, 1),                              # 128x128 tiles
    ({"CRNN_WINO_DMA": "3"}, 1),                              # 64x64 tiles
    ({}, 1),                                                  # per-launch choice, one lane
    ({}, 4),                                                  # per-launch choice, lanes sharing the device
    ({"CRNN_WINOGRAD": "0x38"}, 2),                           # Winograd on conv3-5 only
]


@pytest.mark.gpu
@pytest.mark.parametrize("env,lanes", VARIANTS, ids=lambda v: ",".join("%s=%s" % kv for kv in v.items()) if isinstance(v, dict) else "lanes%d" % v)
def test_variant_strings_match_reference_goldens(env, lanes, golden):
    _, M = golden
    e = dict(os.environ)
    for k in ("CRNN_WINOGRAD", "CRNN_WINO_FUSED", "CRNN_WINO_DMA"):
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
