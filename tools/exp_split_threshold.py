#!/usr/bin/env python3
"""Host-path latency of one batch of n 32x280 lines with and without the hidden-split recurrence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
sd = synth_state_dict(5530, seed=0)
models = {}
for tag, v in (("one-wg", "0"), ("split", "4096")):
    os.environ["CRNN_LSTM_SPLIT"] = v
    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=1)
    m.load_state_dict(sd)
    models[tag] = m
rs = np.random.RandomState(0)
for n in [int(a) for a in sys.argv[1:]]:
    lines = [rs.randint(0, 256, size=(32, 280)).astype(np.uint8) for _ in range(n)]
    row = []
    for tag, m in models.items():
        m.predict_lines(lines, max_batch=4096)
        t0 = time.perf_counter()
        for _ in range(5):
            m.predict_lines(lines, max_batch=4096)
        row.append("%s %.2f ms" % (tag, (time.perf_counter() - t0) / 5 * 1e3))
    print("n=%d  " % n + "  ".join(row))
