"""Experiment: raw device throughput vs lanes, no copy-back / string building."""
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
for nl in [int(a) for a in sys.argv[1:]] or [3]:
    m = CRNN(32, 1, NC, 256, alphabet=keys.alphabetChinese, device=0, lanes=nl)
    m.load_state_dict(sd)
    lines = torch.randint(0, 256, (4, B, 32, W), dtype=torch.uint8, device=dev)
    T = W // 4 + 1
    lab = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(nl)]
    ln = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(nl)]
    def run(n):
        t_call = 0
        for k in range(n):
            j = k % nl
            t = time.perf_counter()
            m.forward_device(lines[k % 4].data_ptr(), B, W, lab[j].data_ptr(), ln[j].data_ptr())
            t_call += time.perf_counter() - t
        m.synchronize()
        return t_call
    run(6)
    t0 = time.perf_counter(); tc = run(100); el = time.perf_counter() - t0
    print("lanes=%d: %.3f ms/step, %.0f lines/s, host time in forward_device %.3f ms/step" % (nl, el / 100 * 1e3, 100 * B / el, tc / 100 * 1e3), flush=True)
    del m
