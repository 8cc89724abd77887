#!/usr/bin/env python3
"""Per-kernel times at several batch sizes (tile-count quantisation experiments).
usage: python tools/exp_batch.py 227 256 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict

m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=1)
m.load_state_dict(synth_state_dict(5530, seed=0))
W = int(os.environ.get("W", "280"))
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.randint(0, 256, (n, 32, W), dtype=torch.uint8, device="cuda:0")
    lab = torch.zeros((n, W // 4 + 1), dtype=torch.int32, device="cuda:0")
    ln = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
    m.set_profiling(True)
    acc = {}
    for k in range(6):
        m.forward_device(g.data_ptr(), n, W, lab.data_ptr(), ln.data_ptr(), sync=True)
        if k:
            for name, ms, fl, by in m.kernel_times():
                a = acc.setdefault(name, [0.0, fl]); a[0] += ms / 5
    tot = sum(v[0] for v in acc.values())
    c15 = sum(v[0] for k, v in acc.items() if k[:5] in ("conv1", "conv2", "conv3", "conv4", "conv5"))
    print("n=%d total %.3f ms (%.1f us/line) conv1-5 incl. transforms %.3f ms  " % (n, tot, tot * 1e3 / n, c15) +
          "  ".join("%s %.3f ms %.0f TF" % (k, v[0], v[1] / v[0] / 1e9) for k, v in acc.items() if v[1] and "rec" not in k))
