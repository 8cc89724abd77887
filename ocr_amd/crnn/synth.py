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


# ---------------------------------------------------------------------------------------------------------------------
# Stress weight sets: the same architecture and key set, value distributions that trained checkpoints can have and the
# base family above does not -- wide per-channel dynamic range inside one tensor, outlier weights, a hot channel,
# duplicated / flat classifier rows.  They exist to pin the reduced-precision machinery of the GPU path (split-fp16 GEMM
# operands with block scales, the classifier's certified pre-filter) against the reference's fp32 arithmetic on other
# distributions than the one it was developed on.  tests/golden/make_golden_stress.py captures the reference's outputs.
STRESS_KINDS = ("chan_spread", "outliers", "cls_dup")


def _consumers(sd, i, lstm):
    """(key, axis) of every tensor that reads output channel c of conv layer i (axis = the input-channel axis)."""
    if i < 6:
        return [("cnn.conv%d.weight" % (i + 1), 1)]
    if lstm:
        return [("rnn.0.rnn.weight_ih_l0" + s, 1) for s in ("", "_reverse")]
    return [("linear.weight", 1)]


def _scale_axis(t, s, axis):
    shp = [1] * t.ndim
    shp[axis] = -1
    return (t.astype(np.float64) * s.reshape(shp)).astype(np.float32)


def chan_spread_factors(layer, seed=0, nh=256):
    """The factor s_c that `chan_spread` puts on output channel c of conv `layer` (0..6) / of the first embedding (7)."""
    tag = "stress.chan_spread."
    if layer == 7:
        return np.power(10.0, (uniform01(tag + "emb0", seed, (nh,)).astype(np.float64) * 2.0 - 1.0) * 2.0)
    span = 3.0 if layer in _BN else 1.5
    return np.power(10.0, (uniform01(tag + "conv%d" % layer, seed, (_NM[layer],)).astype(np.float64) * 2.0 - 1.0) * span)


def stress_state_dict(nclass, kind, nh=256, nc=1, seed=0, lstm=True):
    """`synth_state_dict` transformed into one of STRESS_KINDS.

    chan_spread  every hidden channel c between two layers is rescaled by s_c on the producing side and 1/s_c on the
                 consuming side (ReLU and max-pooling commute with a positive factor, so the network computes the same
                 function up to rounding, as a trained network whose BatchNorm gains differ by orders of magnitude does):
                 BatchNorm layers 2/4/6 through gamma and beta with s_c = 10^U(-3,3) -- gamma/sqrt(var) spans 1e-3..1e3,
                 the folded conv weights 2^20 inside one tensor, and so do the activations of one pixel and the columns of
                 the next layer's weights; conv0/1/3/5 through their rows and bias with 10^U(-1.5,1.5); the first
                 embedding (rows / the second LSTM's input columns) with 10^U(-2,2).
    outliers     1 % of the elements of every conv / LSTM weight matrix are 1000x the rest (each tensor renormalised to
                 the RMS it had, so the activations stay in the working range), and two hot channels: conv1's channel 7
                 and BatchNorm4's channel 3 are 64x larger with only 1/8 in the consumer's column (8x hotter in effect).
    cls_dup      the classifier head: rows 100..199 are copies of row 100 with one bias (100 exact ties whenever that
                 class wins: more than the 64 candidates of cls_select_kernel's fast path; arg-max = lowest index, 100),
                 rows 1000..1299 are zero with one bias (a flat stretch of 300 equal logits that wins where nothing else
                 reaches it); both biases are set so that each block wins a fair share of the timesteps.
    """
    assert kind in STRESS_KINDS, kind
    sd = synth_state_dict(nclass, nh=nh, nc=nc, seed=seed, lstm=lstm)
    tag = "stress.%s." % kind
    if kind == "chan_spread":
        for i in range(7):
            s = chan_spread_factors(i, seed)
            if i in _BN:
                p = "cnn.batchnorm%d." % i
                sd[p + "weight"] = (sd[p + "weight"].astype(np.float64) * s).astype(np.float32)
                sd[p + "bias"] = (sd[p + "bias"].astype(np.float64) * s).astype(np.float32)
            else:
                sd["cnn.conv%d.weight" % i] = _scale_axis(sd["cnn.conv%d.weight" % i], s, 0)
                sd["cnn.conv%d.bias" % i] = (sd["cnn.conv%d.bias" % i].astype(np.float64) * s).astype(np.float32)
            for key, ax in _consumers(sd, i, lstm):
                sd[key] = _scale_axis(sd[key], 1.0 / s, ax)
        if lstm:
            s = chan_spread_factors(7, seed, nh)
            sd["rnn.0.embedding.weight"] = _scale_axis(sd["rnn.0.embedding.weight"], s, 0)
            sd["rnn.0.embedding.bias"] = (sd["rnn.0.embedding.bias"].astype(np.float64) * s).astype(np.float32)
            for sfx in ("", "_reverse"):
                k = "rnn.1.rnn.weight_ih_l0" + sfx
                sd[k] = _scale_axis(sd[k], 1.0 / s, 1)
    elif kind == "outliers":
        keys_ = ["cnn.conv%d.weight" % i for i in range(7)]
        if lstm:
            keys_ += ["rnn.%d.rnn.weight_%s_l0%s" % (l, w, s) for l in (0, 1) for w in ("ih", "hh") for s in ("", "_reverse")]
        for k in keys_:
            w = sd[k].astype(np.float64)
            hot = uniform01(tag + k, seed, w.shape) < np.float32(0.01)
            rms0 = np.sqrt((w ** 2).mean())
            w = np.where(hot, w * 1000.0, w)
            sd[k] = (w * (rms0 / np.sqrt((w ** 2).mean()))).astype(np.float32)
        # hot channels: 64x on the producing side, 1/8 on the consuming side
        sd["cnn.conv1.weight"][7] *= np.float32(64.0)
        sd["cnn.conv1.bias"][7] *= np.float32(64.0)
        sd["cnn.conv2.weight"][:, 7] *= np.float32(0.125)
        sd["cnn.batchnorm4.weight"][3] *= np.float32(64.0)
        sd["cnn.batchnorm4.bias"][3] *= np.float32(64.0)
        sd["cnn.conv5.weight"][:, 3] *= np.float32(0.125)
        if lstm:    # these weights push the blank's logit up by ~4.5; take it out of its bias so that characters still win
            sd["rnn.1.embedding.bias"][0] -= np.float32(4.5)
    else:  # cls_dup
        assert nclass >= 1300, "cls_dup needs the large head"
        wk, bk = ("rnn.1.embedding.weight", "rnn.1.embedding.bias") if lstm else ("linear.weight", "linear.bias")
        W, b = sd[wk].copy(), sd[bk].copy()
        W[100:200] = W[100]
        b[100:200] = np.float32(CLS_DUP_BIAS[0])
        W[1000:1300] = 0.0
        b[1000:1300] = np.float32(CLS_DUP_BIAS[1])
        sd[wk], sd[bk] = W, b
    return sd


# biases of the duplicated block and of the flat block of `cls_dup`, tuned (tests/golden/make_golden_stress.py prints the
# shares) so that on the golden lines each block wins a fair share of the timesteps beside the blank and ordinary classes
CLS_DUP_BIAS = (9.0, 13.0)
