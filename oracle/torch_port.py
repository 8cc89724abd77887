"""A plain torch.nn restatement of the reference CRNN, written for this repo.

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/crnn_oracle.c header): it is the
closest stand-in for "the reference torch CPU path" that can travel to the GPU
box (the reference's own Python cannot).  Used by bench.py's cpu_baseline leg
and checked against the golden fixtures in tests/test_torch_port.py.

Follows crnn/network_torch.py:19-32 (BidirectionalLSTM), :44-86 (layer list,
module names so the reference state-dict keys load unchanged), :88-108
(forward), :130-147 (predict) and crnn/util.py:6-22.
"""
import numpy as np
import torch
import torch.nn as nn
from PIL import Image


class _BiLSTM(nn.Module):
    def __init__(self, n_in, n_hidden, n_out):
        super().__init__()
        self.rnn = nn.LSTM(n_in, n_hidden, bidirectional=True)
        self.embedding = nn.Linear(2 * n_hidden, n_out)

    def forward(self, x):
        rec, _ = self.rnn(x)
        t, b, h = rec.shape
        return self.embedding(rec.reshape(t * b, h)).reshape(t, b, -1)


class TorchPort(nn.Module):
    def __init__(self, nclass, nh=256, leaky=False, lstm=True, alphabet=None):
        super().__init__()
        chans = [1, 64, 128, 256, 256, 512, 512, 512]
        ks = [3, 3, 3, 3, 3, 3, 2]
        bn = {2, 4, 6}
        pools = {0: nn.MaxPool2d(2, 2), 1: nn.MaxPool2d(2, 2), 3: nn.MaxPool2d((2, 2), (2, 1), (0, 1)),
                 5: nn.MaxPool2d((2, 2), (2, 1), (0, 1))}
        cnn = nn.Sequential()
        pool_id = 0
        for i in range(7):
            cnn.add_module("conv%d" % i, nn.Conv2d(chans[i], chans[i + 1], ks[i], 1, 1 if ks[i] == 3 else 0))
            if i in bn:
                cnn.add_module("batchnorm%d" % i, nn.BatchNorm2d(chans[i + 1]))
            cnn.add_module("relu%d" % i, nn.LeakyReLU(0.2) if leaky else nn.ReLU())
            if i in pools:
                cnn.add_module("pooling%d" % pool_id, pools[i])
                pool_id += 1
        self.cnn = cnn
        self.lstm = lstm
        if lstm:
            self.rnn = nn.Sequential(_BiLSTM(512, nh, nh), _BiLSTM(nh, nh, nclass))
        else:
            self.linear = nn.Linear(2 * nh, nclass)
        self.alphabet = alphabet

    def load_numpy(self, sd):
        missing = self.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
        assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys)
        self.eval()
        return self

    def forward(self, x):
        f = self.cnn(x)
        assert f.shape[2] == 1
        seq = f.squeeze(2).permute(2, 0, 1)
        if self.lstm:
            return self.rnn(seq)
        t, b, c = seq.shape
        return self.linear(seq.reshape(t * b, c)).reshape(t, b, -1)

    @staticmethod
    def resize_normalize(img, img_h=32):
        scale = img.size[1] * 1.0 / img_h
        w = int(img.size[0] / scale)
        img = img.resize((w, img_h), Image.BILINEAR)
        return ((np.array(img) / 255.0 - 0.5) / 0.5).astype(np.float32)

    def decode(self, idx):
        out = []
        for i in range(len(idx)):
            if idx[i] != 0 and not (i > 0 and idx[i - 1] == idx[i]):
                out.append(self.alphabet[idx[i] - 1])
        return "".join(out)

    @torch.no_grad()
    def predict_gray_batch(self, lines):
        """uint8 [n,32,w] equal-width lines -> list of str (preprocess + forward + argmax + decode)."""
        x = np.stack([self.resize_normalize(Image.fromarray(ln, mode="L")) for ln in lines])
        out = self.forward(torch.from_numpy(x)[:, None])
        idx = out.argmax(2).T.contiguous().numpy()
        return [self.decode(r) for r in idx]
