"""Experiment / stress check (not collected by pytest): every line width from 8 to 330 and a few wide ones, one line per
call and in ragged batches, GPU against the oracle (65-class and 5530-class heads): exercises the partial-tile, segment
and padding edges of the Winograd transforms, the pooled output transforms and the classifier pre-filter.

    python tests/exp_width_sweep.py       (needs the GPU)
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
from oracle import oracle as O

rs = np.random.RandomState(7)
worst = 0.0
for nclass, alpha in ((65, keys.alphabetEnglish), (5530, keys.alphabetChinese)):
    sd = synth_state_dict(nclass, seed=0)
    m = CRNN(32, 1, nclass, 256, alphabet=alpha, device=0)
    m.load_state_dict(sd)
    ora = O.Oracle(sd, nclass)
    widths = list(range(8, 331)) + [511, 512, 513, 767, 1024, 1031] if nclass == 65 else list(range(8, 331, 7)) + [512, 1031]
    lines = [rs.randint(0, 256, size=(32, w)).astype(np.uint8) for w in widths]
    bad = 0
    # one line per call: logits and strings
    for w, ln in zip(widths, lines):
        raw, tt, logits, _, _ = m.forward_logits([ln])
        x = O.resize_normalize(ln)[None]
        ref = ora.forward(x)                       # [T, 1, nclass]
        err = float(np.abs(logits[0, :tt[0]] - ref[:, 0]).max())
        worst = max(worst, err)
        s_gpu = m.predict_lines([ln])[0]           # product path (pre-filter for the big head)
        s_ref = O.predict_gray(ora, ln, alpha)
        if s_gpu != s_ref or err > 1e-3:
            bad += 1
            print("MISMATCH nclass %d width %d err %.2e %r %r" % (nclass, w, err, s_gpu, s_ref))
    # ragged batches through the bucketing host path
    got = m.predict_lines(lines, max_batch=64, width_spread=1.3)
    want = [O.predict_gray(ora, ln, alpha) for ln in lines]
    nb = sum(a != b for a, b in zip(got, want))
    print("nclass %d: %d widths, single-line mismatches %d, batched mismatches %d, worst logit error so far %.2e" %
          (nclass, len(widths), bad, nb, worst))
