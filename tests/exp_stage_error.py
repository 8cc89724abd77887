"""Max |logit - oracle| on the two lines of tests/test_hip_parity.py::test_stages_against_oracle (5530-class head),
per stage, for the kernel selection given by the environment."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
from oracle import oracle as O
G = np.load(os.path.join(ROOT, "tests/golden/crnn_golden.npz"))
sd = synth_state_dict(5530, seed=0)
m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0); m.load_state_dict(sd)
ora = O.Oracle(sd, 5530)
x = np.stack([O.resize_normalize(G["syn280/noise/gray"][0]), O.resize_normalize(G["syn280/text/gray"][0])])
logits, taps = ora.forward(x, stages=True)
errs = []
for st in range(6):
    got = np.transpose(m.debug_stage(x, st), (0, 3, 1, 2))
    errs.append("%.1e" % np.abs(got - taps[st]).max())
got = np.transpose(m.debug_stage(x, 6), (0, 2, 1))[:, :, None, :]
errs.append("%.1e" % np.abs(got - taps[6]).max())
lg = np.transpose(m.debug_stage(x, 10), (1, 0, 2))
print(os.environ.get("TAG", ""), "stage errors p0,p1,c2,p2,c4,p3,seq:", " ".join(errs), " logits: noise %.2e text %.2e" %
      (np.abs(lg - logits)[:, 0].max(), np.abs(lg - logits)[:, 1].max()))
