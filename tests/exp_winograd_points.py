"""Experiment (not collected by pytest): rounding error of Winograd F(4,3) on conv1 for several sets of interpolation
points (Cook-Toom matrices built here and checked against the direct correlation), input = conv1's real input for the
uniform-noise golden line; then the exact rational matrices of the set the kernels use (0, 1, -1, 2, -1/2, inf).

    python tests/exp_winograd_points.py
"""
import numpy as np, torch, sys, os, json
from numpy.polynomial import polynomial as P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ocr_amd.crnn.synth import synth_state_dict
from oracle.torch_port import TorchPort
from PIL import Image
def cook_toom(points, m, r):
    n = m + r - 1
    a = np.array(points, dtype=np.float64)
    AT = np.zeros((m, n)); 
    for i in range(m): AT[i, :n-1] = a ** i
    AT[m-1, n-1] = 1
    G = np.zeros((n, r))
    for j in range(n-1):
        Nj = np.prod([a[j] - a[l] for l in range(n-1) if l != j])
        G[j] = a[j] ** np.arange(r) / Nj
    G[n-1, r-1] = 1
    BT = np.zeros((n, n))
    for j in range(n-1):
        c = np.array([1.0])
        for l in range(n-1):
            if l != j: c = P.polymul(c, [-a[l], 1.0])
        BT[j, :len(c)] = c
    c = np.array([1.0])
    for l in range(n-1): c = P.polymul(c, [-a[l], 1.0])
    BT[n-1, :len(c)] = c
    return AT, G, BT
G_ = np.load(os.path.join(ROOT, 'tests/golden/crnn_golden.npz'))
sd = synth_state_dict(65, seed=0)
net = TorchPort(65).load_numpy(sd)
x = torch.from_numpy(TorchPort.resize_normalize(Image.fromarray(G_["syn280/noise/gray"][0], mode="L")))[None, None]
with torch.no_grad():
    p0 = net.cnn[:3](x)          # conv0, relu0, pooling0
    conv1 = net.cnn.conv1
    ref = torch.nn.functional.conv2d(p0.double(), conv1.weight.double(), conv1.bias.double(), padding=1)
    d32 = torch.nn.functional.conv2d(p0, conv1.weight, conv1.bias, padding=1)
print('p0 max %.2f  out std %.3f  direct fp32 err %.2e' % (p0.abs().max(), ref.std(), (d32.double()-ref).abs().max()))
def wino(points, m):
    AT, G, BT = cook_toom(points, m, 3)
    ATt, Gt, BTt = [torch.tensor(t) for t in (AT, G, BT)]
    w = conv1.weight.detach().double()
    U = torch.einsum("ak,oikl,bl->aboi", Gt, w, Gt).float()
    n, c, h, wd = p0.shape
    th, tw = -(-h // m), -(-wd // m)
    xp = torch.nn.functional.pad(p0, (1, tw*m - wd + 1, 1, th*m - h + 1))
    p = xp.unfold(2, m+2, m).unfold(3, m+2, m)
    v = torch.einsum("ak,ncyxkl->ncyxal", BTt.float(), p)
    v = torch.einsum("bl,ncyxal->abnyxc", BTt.float(), v)
    a, b = v.shape[:2]
    mm = torch.matmul(v.reshape(a, b, -1, c), U.transpose(2, 3))
    o = mm.shape[-1]
    mm = mm.reshape(a, b, n, th, tw, o)
    y = torch.einsum("ia,abnyxo->ibnyxo", ATt.float(), mm)
    y = torch.einsum("jb,ibnyxo->noyixj", ATt.float(), y).reshape(n, o, th*m, tw*m)[:, :, :h, :wd] + conv1.bias[None, :, None, None]
    return (y.double() - ref).abs().max().item()
with torch.no_grad():
    print('F(2,3) {0,1,-1}         %.2e' % wino([0, 1, -1], 2))
    for pts in ([0,1,-1,2,-2], [0,1,-1,0.5,-0.5], [0,1,-1,0.5,-2], [0,1,-1,2,-0.5], [0, 0.5, -0.5, 1.5, -1.5], [0,1,-1,0.75,-0.75],[0, 0.5,-0.5,1,-1]):
        print('F(4,3)', pts, '%.2e' % wino(pts, 4))
from fractions import Fraction
def cook_toom_frac(points, m, r):
    n = m + r - 1
    a = [Fraction(p) for p in points]
    AT = [[a[j] ** i for j in range(n-1)] + [Fraction(1 if i == m-1 else 0)] for i in range(m)]
    G = []
    for j in range(n-1):
        Nj = Fraction(1)
        for l in range(n-1):
            if l != j: Nj *= (a[j] - a[l])
        G.append([a[j] ** k / Nj for k in range(r)])
    G.append([Fraction(0)] * (r-1) + [Fraction(1)])
    def polymul(c, d):
        out = [Fraction(0)] * (len(c) + len(d) - 1)
        for i, x in enumerate(c):
            for j, y in enumerate(d): out[i+j] += x*y
        return out
    BT = []
    for j in range(n-1):
        c = [Fraction(1)]
        for l in range(n-1):
            if l != j: c = polymul(c, [-a[l], Fraction(1)])
        BT.append(c + [Fraction(0)] * (n - len(c)))
    c = [Fraction(1)]
    for l in range(n-1): c = polymul(c, [-a[l], Fraction(1)])
    BT.append(c + [Fraction(0)] * (n - len(c)))
    return AT, G, BT
for pts in ([0, 1, -1, 2, Fraction(-1, 2)],):
    AT, G, BT = cook_toom_frac(pts, 4, 3)
    print("AT"); [print([str(x) for x in r]) for r in AT]
    print("G"); [print([str(x) for x in r]) for r in G]
    print("BT"); [print([str(x) for x in r]) for r in BT]
