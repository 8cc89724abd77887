"""Procedural (seeded, hash-based) weights for the CRNN.

The trained checkpoints the reference loads (``models/ocr-lstm.pth`` /
``ocr-english.pth``, ``config.py:58-60``) are not part of the reference
snapshot, so every parity test, golden fixture and benchmark uses weights
generated here.  The generator is counter based (splitmix64 of
``hash(name, seed) + index``) and uses only integer ops plus exact
int->float conversions, adds and multiplies, so the same bits come out on
every machine; 45 MB of weights never need committing.

The key set and shapes are those of the reference module's ``state_dict()``
(``crnn/network_torch.py:44-86``, enumerated in SURVEY.md section 8 a17).
"""
import numpy as np

_NM = [64, 128, 256, 256, 512, 512, 512]
_KS = [3, 3, 3, 3, 3, 3, 2]
_BN = (2, 4, 6)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name, seed):
    h = 0xCBF29CE484222325 ^ (seed & 0xFFFFFFFFFFFFFFFF)
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _bits(name, seed, n):
    base = np.uint64(_fnv1a(name, seed))
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64)
    return _splitmix64(ctr)


def uniform01(name, seed, shape):
    """U[0,1) with 24 random bits (exact in float32)."""
    n = int(np.prod(shape))
    z = _bits(name, seed, n)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def normal(name, seed, shape):
    """~N(0,1): Irwin-Hall sum of four 16-bit uniforms (exact arithmetic)."""
    n = int(np.prod(shape))
    z = _bits(name, seed, n)
    m = np.uint64(0xFFFF)
    s = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + ((z >> np.uint64(48)) & m))
    # sum of 4 U{0..65535}: mean 131070, var 4*(65536^2-1)/12
    x = (s.astype(np.float64) - 131070.0) * (1.0 / (65536.0 * np.sqrt(1.0 / 3.0)))
    return x.astype(np.float32).reshape(shape)


def synth_state_dict(nclass, nh=256, nc=1, seed=0, lstm=True, blank_bias=None):
    """Named float32 numpy tensors with the reference's state-dict keys/shapes."""
    if blank_bias is None:  # large enough that the CTC blank wins on ~1/4 of the timesteps
        blank_bias = 12.0 if nclass > 1000 else 7.5
    sd = {}
    for i in range(7):
        cin = nc if i == 0 else _NM[i - 1]
        cout, k = _NM[i], _KS[i]
        fan_in = cin * k * k
        w = normal("cnn.conv%d.weight" % i, seed, (cout, cin, k, k)) * np.float32(np.sqrt(2.0 / fan_in))
        sd["cnn.conv%d.weight" % i] = w.astype(np.float32)
        sd["cnn.conv%d.bias" % i] = (normal("cnn.conv%d.bias" % i, seed, (cout,)) * np.float32(0.1)).astype(np.float32)
        if i in _BN:
            p = "cnn.batchnorm%d." % i
            sd[p + "weight"] = (uniform01(p + "weight", seed, (cout,)) + np.float32(0.5)).astype(np.float32)
            sd[p + "bias"] = (normal(p + "bias", seed, (cout,)) * np.float32(0.1)).astype(np.float32)
            sd[p + "running_mean"] = (normal(p + "running_mean", seed, (cout,)) * np.float32(0.1)).astype(np.float32)
            sd[p + "running_var"] = (uniform01(p + "running_var", seed, (cout,)) + np.float32(0.5)).astype(np.float32)
    if lstm:
        for layer, (nin, nout) in enumerate(((512, nh), (nh, nclass))):
            for sfx in ("", "_reverse"):
                p = "rnn.%d.rnn." % layer
                a_ih = np.float32(3.0 / np.sqrt(nin))
                a_hh = np.float32(3.0 / np.sqrt(nh))
                n = p + "weight_ih_l0" + sfx
                sd[n] = ((uniform01(n, seed, (4 * nh, nin)) * np.float32(2.0) - np.float32(1.0)) * a_ih).astype(np.float32)
                n = p + "weight_hh_l0" + sfx
                sd[n] = ((uniform01(n, seed, (4 * nh, nh)) * np.float32(2.0) - np.float32(1.0)) * a_hh).astype(np.float32)
                for b in ("bias_ih_l0", "bias_hh_l0"):
                    n = p + b + sfx
                    sd[n] = (normal(n, seed, (4 * nh,)) * np.float32(0.1)).astype(np.float32)
            p = "rnn.%d.embedding." % layer
            sd[p + "weight"] = (normal(p + "weight", seed, (nout, 2 * nh)) * np.float32(8.0 / np.sqrt(2.0 * nh))).astype(np.float32)
            b = (normal(p + "bias", seed, (nout,)) * np.float32(0.1)).astype(np.float32)
            if layer == 1:
                b[0] += np.float32(blank_bias)  # so the CTC blank wins sometimes
            sd[p + "bias"] = b
    else:
        sd["linear.weight"] = (normal("linear.weight", seed, (nclass, 2 * nh)) * np.float32(8.0 / np.sqrt(2.0 * nh))).astype(np.float32)
        b = (normal("linear.bias", seed, (nclass,)) * np.float32(0.1)).astype(np.float32)
        b[0] += np.float32(blank_bias)
        sd["linear.bias"] = b
    return sd
