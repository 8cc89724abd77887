"""Experiment (not collected by pytest): logit error of Winograd F(4x4,3x3) / F(2x4,3x3) in fp32 on
conv1-5, measured on the CPU with the torch restatement, against the reference goldens.

    python tests/exp_winograd_f4_error.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ocr_amd.crnn import keys  # noqa: E402
from ocr_amd.crnn.synth import synth_state_dict  # noqa: E402
from oracle.torch_port import TorchPort  # noqa: E402

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                    [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                   [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                   dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
# F(4,3) with the points 0, 1, -1, 2, -1/2, inf (what the kernels use: about 3x less rounding error than +-1, +-2)
BT4B = torch.tensor([[1, 1.5, -2, -1.5, 1, 0], [0, -1, -2.5, -.5, 1, 0], [0, 1, .5, -2.5, 1, 0],
                     [0, -.5, -1, .5, 1, 0], [0, 2, -1, -2, 1, 0], [0, 1, 1.5, -2, -1.5, 1]], dtype=torch.float64)
G4B = torch.tensor([[1, 0, 0], [-1 / 3, -1 / 3, -1 / 3], [1 / 3, -1 / 3, 1 / 3], [1 / 15, 2 / 15, 4 / 15],
                    [-16 / 15, 8 / 15, -4 / 15], [0, 0, 1]], dtype=torch.float64)
AT4B = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -.5, 0], [0, 1, 1, 4, .25, 0], [0, 1, -1, 8, -.125, 1]],
                    dtype=torch.float64)
SETS = {2: (BT2, G2, AT2), 4: (BT4, G4, AT4), 5: (BT4B, G4B, AT4B)}   # 5 = the 4-output form with the better points


class WinoConv(nn.Module):
    """3x3 pad-1 conv as F(mh x mw, 3x3), every step in fp32 (weights transformed in f64 then rounded)."""

    def __init__(self, conv, mh, mw, f64_transforms=False):
        super().__init__()
        self.mh, self.mw = mh, mw
        self.f64 = f64_transforms
        bth, gh, ath = SETS[mh]
        btw, gw, atw = SETS[mw]
        w = conv.weight.detach().double()                       # [O, I, 3, 3]
        self.U = torch.einsum("ak,oikl,bl->aboi", gh, w, gw).float().contiguous()
        self.bias = conv.bias.detach()
        self.bth, self.btw, self.ath, self.atw = bth.float(), btw.float(), ath.float(), atw.float()

    def forward(self, x):
        n, c, h, w = x.shape
        mh, mw = min(self.mh, 4), min(self.mw, 4)
        th, tw = -(-h // mh), -(-w // mw)
        xp = F.pad(x, (1, tw * mw - w + 1, 1, th * mh - h + 1))
        # patches [n, c, th, tw, mh+2, mw+2]
        p = xp.unfold(2, mh + 2, mh).unfold(3, mw + 2, mw)
        if self.f64:    # transforms in float64, V rounded to fp32 once
            v = torch.einsum("ak,ncyxkl->ncyxal", self.bth.double(), p.double())
            v = torch.einsum("bl,ncyxal->abnyxc", self.btw.double(), v).float()
        else:
            v = torch.einsum("ak,ncyxkl->ncyxal", self.bth, p)
            v = torch.einsum("bl,ncyxal->abnyxc", self.btw, v)      # [a, b, n, th, tw, c]
        a, b = v.shape[:2]
        m = torch.matmul(v.reshape(a, b, -1, c), self.U.transpose(2, 3))   # [a,b,ntile,O]
        o = m.shape[-1]
        m = m.reshape(a, b, n, th, tw, o)
        if self.f64:
            y = torch.einsum("ia,abnyxo->ibnyxo", self.ath.double(), m.double())
            y = torch.einsum("jb,ibnyxo->noyixj", self.atw.double(), y).float()
        else:
            y = torch.einsum("ia,abnyxo->ibnyxo", self.ath, m)
            y = torch.einsum("jb,ibnyxo->noyixj", self.atw, y)      # [n,o,th,mh,tw,mw]
        y = y.reshape(n, o, th * mh, tw * mw)[:, :, :h, :w]
        return y + self.bias[None, :, None, None]


class WinoConv6(nn.Module):
    """conv6 (2x2, no padding, input height 2 -> output height 1) as 1-D F(4,2) along the width, fp32."""
    BT = torch.tensor([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]],
                      dtype=torch.float64)
    G = torch.tensor([[.5, 0], [-.5, -.5], [-1 / 6, 1 / 6], [1 / 6, 1 / 3], [0, 1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 0], [0, 1, -1, 8, 1]], dtype=torch.float64)

    def __init__(self, conv):
        super().__init__()
        w = conv.weight.detach().double()                       # [O, I, 2, 2]
        self.U = torch.einsum("bl,oikl->bkoi", self.G, w).float().contiguous()   # [5, ky, O, I]
        self.bias = conv.bias.detach()

    def forward(self, x):
        n, c, h, w = x.shape
        wo = w - 1
        tw = -(-wo // 4)
        xp = F.pad(x, (0, tw * 4 + 1 - w))
        p = xp.unfold(3, 5, 4)                                  # [n, c, 2, tw, 5]
        v = torch.einsum("bl,nckxl->bknxc", self.BT.float(), p)   # [5, ky, n, tw, c]
        m = torch.einsum("bknxc,bkoc->bnxo", v, self.U)          # sum over ky and c
        y = torch.einsum("jb,bnxo->noxj", self.AT.float(), m).reshape(n, -1, 1, tw * 4)[..., :wo]
        return y + self.bias[None, :, None, None]


def build(nclass, sd, layers, mh, mw, f64=()):
    m = TorchPort(nclass).load_numpy(sd)
    for i in layers:
        if i == 6:
            m.cnn.conv6 = WinoConv6(m.cnn.conv6)
        else:
            setattr(m.cnn, "conv%d" % i, WinoConv(getattr(m.cnn, "conv%d" % i), mh, mw, i in f64))
    return m


def prep(gray):
    from PIL import Image
    return torch.from_numpy(TorchPort.resize_normalize(Image.fromarray(gray, mode="L")))[None, None]


def main():
    G = np.load(os.path.join(ROOT, "tests/golden/crnn_golden.npz"))
    M = json.load(open(os.path.join(ROOT, "tests/golden/crnn_golden.json"), encoding="utf-8"))
    sd = synth_state_dict(65, seed=0)
    cases = [("var/%d" % w, G["var/%d/gray" % w], G["var/%d/en_logits" % w]) for w in M["var_widths"]]
    variants = [("direct", [], 2, 2), ("F2x2 c1-5", [1, 2, 3, 4, 5], 2, 2), ("F2x4 c1-5", [1, 2, 3, 4, 5], 2, 4),
                ("F4x4 c1-5", [1, 2, 3, 4, 5], 4, 4), ("F4x4 c2-5", [2, 3, 4, 5], 4, 4),
                ("F4x4 c4-5", [4, 5], 4, 4), ("F4x4 c1", [1], 4, 4), ("F(4,2) c6", [6], 4, 4),
                ("F4x4 c1-5 + F(4,2) c6", [1, 2, 3, 4, 5, 6], 4, 4),
                ("F4x4' c1-5 (pts 2,-1/2)", [1, 2, 3, 4, 5], 5, 5), ("F4x4' c2-5 (pts 2,-1/2)", [2, 3, 4, 5], 5, 5),
                ("F4x4' c1 (pts 2,-1/2)", [1], 5, 5),
                ("F4x4 c1 f64 transforms", [1], 4, 4, (1,)), ("F4x4 c1-5 f64 transforms", [1, 2, 3, 4, 5], 4, 4, (1, 2, 3, 4, 5))]
    torch.set_num_threads(8)
    # the hardest line of the GPU suite: uniform noise (stage test), here with the 65-class head
    noise = G["syn280/noise/gray"][0]
    ref_net = build(65, sd, [], 2, 2)
    with torch.no_grad():
        noise_ref = ref_net(prep(noise))[:, 0].numpy()
    for var in variants:
        name, layers, mh, mw = var[:4]
        net = build(65, sd, layers, mh, mw, var[4] if len(var) > 4 else ())
        worst, flips = 0.0, 0
        with torch.no_grad():
            for tag, gray, ref in cases:
                lg = net(prep(gray))[:, 0].numpy()
                worst = max(worst, float(np.abs(lg - ref).max()))
                flips += int((lg.argmax(1) != ref.argmax(1)).sum())
            nz = float(np.abs(net(prep(noise))[:, 0].numpy() - noise_ref).max())
        print("%-26s max |logit - reference| %.2e   argmax flips %d   noise line vs direct %.2e" % (name, worst, flips, nz), flush=True)


if __name__ == "__main__":
    main()
