"""oracle/torch_port.py (the CPU baseline bench.py times) against the golden fixtures."""
import numpy as np
import torch

from ocr_amd.crnn import keys
from oracle.torch_port import TorchPort


def test_port_matches_reference_goldens(golden, synth_zh):
    G, M = golden
    m = TorchPort(5530, alphabet=keys.alphabetChinese).load_numpy(synth_zh)
    for tag in ("noise", "text"):
        assert m.predict_gray_batch(G["syn280/%s/gray" % tag]) == M["syn280_%s" % tag]
    it = M["invoice"][6]
    x = ((G["inv/%s/resized" % it["name"]] / 255.0 - 0.5) / 0.5).astype(np.float32)
    with torch.no_grad():
        lg = m(torch.from_numpy(x)[None, None])[:, 0].numpy()
    assert np.abs(lg - G["inv/invoice_number/zh_logits"]).max() < 1e-4
    assert m.decode(lg.argmax(1)) == it["zh"]
