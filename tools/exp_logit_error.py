import os, sys, json, numpy as np
sys.path.insert(0, os.getcwd())
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
G = np.load("tests/golden/crnn_golden.npz"); M = json.load(open("tests/golden/crnn_golden.json"))
en = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0); en.load_state_dict(synth_state_dict(65))
lines = [G["var/%d/gray" % w] for w in M["var_widths"]]
raw, tt, logits, _, _ = en.forward_logits(lines)
errs = [float(np.abs(logits[i, :tt[i]] - G["var/%d/en_logits" % w]).max()) for i, w in enumerate(M["var_widths"])]
print("max |logit - reference| per width", dict(zip(M["var_widths"], ["%.2e" % e for e in errs])))
inv = sorted(k[:-4] for k in G.files if k.startswith("inv/") and k.endswith("/bgr"))
crops = [G[k + "/bgr"] for k in inv]
raw, tt, logits, _, _ = en.forward_logits(crops)
print("invoice crops (65-class): max err %.2e" % max(float(np.abs(logits[i, :tt[i]] - G[k + "/en_logits"]).max()) for i, k in enumerate(inv)))
