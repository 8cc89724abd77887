"""CPU experiment: how close does an fp32 GEMM emulated with split-bf16 operands (bf16x3 / x6) stay to
the plain fp32 network?  Products of bf16 values are exact in fp32 and the MFMA accumulates in fp32, so
conv(x_hi, w_hi) + conv(x_hi, w_lo) + conv(x_lo, w_hi) in torch fp32 models the hardware result up to
summation order.  Test infrastructure only."""
import sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_port import TorchPort
from ocr_amd.crnn.synth import synth_state_dict

torch.set_num_threads(8)
NCLASS = 5530
MODE = sys.argv[1] if len(sys.argv) > 1 else "x3"      # x3 | x6 | x3t (truncating split)
ROUND = not MODE.endswith("t")


def bf16(x):
    if MODE.startswith("h"):
        return x.to(torch.float16).to(torch.float32)
    if ROUND:
        return x.to(torch.bfloat16).to(torch.float32)
    return (x.view(torch.int32) & -65536).view(torch.float32)


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = bf16(r)
        parts.append(p)
        r = r - p
    return parts


def terms(n):
    if n == 2:
        return [(0, 0), (0, 1), (1, 0)]
    return [(i, j) for i in range(3) for j in range(3) if i + j <= 2]


def pow2_scale(t, target=2.0 ** 13):
    """power of two s with max|t * s| in (target/2, target]: exact to apply and to undo"""
    mx = float(t.abs().max())
    return 2.0 ** np.floor(np.log2(target / mx)) if mx > 0 else 1.0


def split_op(op, x, w):
    n = 2 if MODE[-1] == "3" else 3
    if MODE.startswith("hs"):                # fp16 with per-tensor power-of-two scaling (keeps the low parts normal)
        sx, sw = pow2_scale(x), pow2_scale(w)
        xs, ws = split(x * sx, 2), split(w * sw, 2)
        out = None
        for i, j in reversed(terms(2)):
            t = op(xs[i], ws[j])
            out = t if out is None else out + t
        return out / (sx * sw)
    xs, ws = split(x, n), split(w, n)
    out = None
    for i, j in reversed(terms(n)):          # small terms first
        t = op(xs[i], ws[j])
        out = t if out is None else out + t
    return out


sd = synth_state_dict(NCLASS)
m = TorchPort(NCLASS).load_numpy(sd)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "crnn_golden.npz"))
gray = np.concatenate([g["syn280/text/gray"], g["syn280/noise/gray"]])
rng = np.random.default_rng(1)
x = torch.from_numpy(((gray.astype(np.float32) / 255.0) - 0.5) / 0.5)[:, None]
print("lines", x.shape)

with torch.no_grad():
    ref = m(x)                                # [T, n, nclass]

    # split-emulated forward: BN folded into conv weights first (as the GPU path does)
    h = x
    mods = dict(m.cnn.named_children())
    for i in range(7):
        c = mods["conv%d" % i]
        w, b = c.weight.double(), c.bias.double()
        if "batchnorm%d" % i in mods:
            bn = mods["batchnorm%d" % i]
            s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            w = w * s[:, None, None, None]
            b = (b - bn.running_mean.double()) * s + bn.bias.double()
        w, b = w.float(), b.float()
        pad = c.padding
        if i == 0:
            h = F.conv2d(h, w, b, padding=pad)            # conv0 stays a direct fp32 kernel
        else:
            h = split_op(lambda a, ww: F.conv2d(a, ww, None, padding=pad), h, w) + b[None, :, None, None]
        h = torch.relu(h)
        if "pooling%d" % {0: 0, 1: 1, 3: 2, 5: 3}.get(i, -1) in mods and i in (0, 1, 3, 5):
            h = mods["pooling%d" % {0: 0, 1: 1, 3: 2, 5: 3}[i]](h)
    seq = h.squeeze(2).permute(2, 0, 1)                    # [T, n, 512]
    for l in range(2):
        bl = m.rnn[l]
        T, n, _ = seq.shape
        outs = []
        for d, sfx in enumerate(["", "_reverse"]):
            wih = getattr(bl.rnn, "weight_ih_l0" + sfx); whh = getattr(bl.rnn, "weight_hh_l0" + sfx)
            bias = getattr(bl.rnn, "bias_ih_l0" + sfx) + getattr(bl.rnn, "bias_hh_l0" + sfx)
            xp = split_op(lambda a, ww: a @ ww.T, seq.reshape(T * n, -1), wih).reshape(T, n, -1) + bias
            hh = torch.zeros(n, 256); cc = torch.zeros(n, 256); o = [None] * T
            for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                gts = xp[t] + hh @ whh.T                   # recurrence stays fp32
                i_, f_, g_, o_ = gts.chunk(4, 1)
                cc = torch.sigmoid(f_) * cc + torch.sigmoid(i_) * torch.tanh(g_)
                hh = torch.sigmoid(o_) * torch.tanh(cc)
                o[t] = hh
            outs.append(torch.stack(o))
        rec = torch.cat(outs, 2)
        seq = (split_op(lambda a, ww: a @ ww.T, rec.reshape(T * n, -1), bl.embedding.weight) + bl.embedding.bias).reshape(T, n, -1)
    out = seq

err = (out - ref).abs()
print("mode", MODE, "max |dlogit| %.3e  mean %.3e  logit scale %.2f" % (err.max(), err.mean(), ref.abs().max()))
ia, ib = ref.argmax(2), out.argmax(2)
print("argmax mismatches", int((ia != ib).sum()), "of", ia.numel())
top2 = ref.topk(2, 2).values
print("min top-2 margin %.4f" % float((top2[..., 0] - top2[..., 1]).min()))
