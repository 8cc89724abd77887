"""Experiment (not collected by pytest): how many classes survive the certified bf16 pre-filter of the classifier's
arg-max (ocr_amd/csrc/crnn_kernels.hip, cls_select_kernel) -- emulated on the host from the GPU's own classifier input
(stage 9, the second BiLSTM's output) for noise and text lines with the synthetic 5530-class weights.

    python tests/exp_cls_filter.py          (needs the GPU)
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
from oracle import oracle as O


def to_bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16          # round to nearest even
    return u.astype(np.uint32).view(np.float32)


G = np.load(os.path.join(ROOT, "tests/golden/crnn_golden.npz"))
sd = synth_state_dict(5530, seed=0)
m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0)
m.load_state_dict(sd)
W = sd["rnn.1.embedding.weight"].astype(np.float32)
b = sd["rnn.1.embedding.bias"].astype(np.float32)
wn = np.sqrt((W.astype(np.float64) ** 2).sum(1))
W16 = to_bf16(W)
for tag in ("noise", "text"):
    x = np.stack([O.resize_normalize(g) for g in G["syn280/%s/gray" % tag][:16]])
    a = m.debug_stage(x, 9).reshape(-1, 512)
    an = np.sqrt((a.astype(np.float64) ** 2).sum(1))
    exact = a.astype(np.float64) @ W.astype(np.float64).T + b
    approx = to_bf16(a).astype(np.float64) @ W16.astype(np.float64).T + b
    approx = to_bf16(approx.astype(np.float32)).astype(np.float64)      # the kernel stores them as bf16
    eps = 0.00785 * an[:, None] * wn[None, :] + 0.004 * np.abs(approx)
    assert (np.abs(approx - exact) <= eps).all(), "bound violated"
    lb = (approx - eps).max(1)
    cand = (approx + eps >= lb[:, None]).sum(1)
    hit = (approx + eps >= lb[:, None])[np.arange(len(a)), exact.argmax(1)].all()
    print("%-5s rows %d  |a| %.1f  |w| %.2f  eps %.3f  logit std %.2f  max|approx-exact| %.4f  candidates: mean %.1f median %d "
          "p90 %d max %d  single %.0f%%  arg-max always among them: %s" %
          (tag, len(a), an.mean(), wn.mean(), eps.mean(), exact.std(), np.abs(approx - exact).max(), cand.mean(),
           np.median(cand), np.percentile(cand, 90), cand.max(), 100.0 * (cand == 1).mean(), hit))
