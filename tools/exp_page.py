import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
from ocr_amd.crnn.batching import width_buckets
zh = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=1)
zh.load_state_dict(synth_state_dict(5530, seed=0))
G = np.load("tests/golden/crnn_golden.npz")
shapes = [G[k].shape for k in G.files if k.startswith("inv/") and k.endswith("/bgr")]
print("crop shapes", shapes)
rs = np.random.RandomState(3)
ph, pw = 2339, 1654
page = rs.randint(0, 256, size=(ph, pw, 3)).astype(np.uint8)
rr = []
for _ in range(48):
    h, w, _c = shapes[rs.randint(len(shapes))]
    x1, y1 = rs.randint(0, pw - w), rs.randint(0, ph - h)
    rr.append((x1, y1, x1 + w, y1 + h))
ws = [int(w / (h / 32.0)) for (x1, y1, x2, y2) in rr for w, h in [(x2 - x1, y2 - y1)]]
print("resized widths", sorted(ws))
print("buckets", [len(b) for b in width_buckets(ws, 256, 1.25, 24576)])
zh.ocr_page(page, rr)
for k in range(3):
    t0 = time.perf_counter(); zh.set_page(page); t1 = time.perf_counter()
    out = zh.predict_rects(rr); t2 = time.perf_counter()
    print("set_page %.2f ms  predict_rects %.2f ms  device stages (last batch) %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, zh.timing()))
