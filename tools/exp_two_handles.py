"""Experiment: alternate batches between two handles (two streams) to overlap the LSTM
recurrence of one batch with the conv stack of the next."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict

B, W, NC = 256, 280, 5530
dev = torch.device("cuda", 0)
sd = synth_state_dict(NC, seed=0)
for nh in (1, 2, 3):
    models = []
    for i in range(nh):
        m = CRNN(32, 1, NC, 256, alphabet=keys.alphabetChinese, device=0)
        m.load_state_dict(sd)
        models.append(m)
    lines = torch.randint(0, 256, (4, B, 32, W), dtype=torch.uint8, device=dev)
    T = W // 4 + 1
    lab = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(nh)]
    ln = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(nh)]
    def run(n):
        for k in range(n):
            j = k % nh
            models[j].forward_device(lines[k % 4].data_ptr(), B, W, lab[j].data_ptr(), ln[j].data_ptr())
        for m in models:
            m.synchronize()
    run(4)
    t0 = time.perf_counter(); run(40); el = time.perf_counter() - t0
    print("handles=%d: %.3f ms/step, %.0f lines/s" % (nh, el / 40 * 1e3, 40 * B / el), flush=True)
    del models
