"""Parity on weight distributions that stress the reduced-precision machinery of the GPU path (synth.py
stress_state_dict: channel scales spread over 2^20 inside one tensor, outlier weights and hot channels, duplicated and
flat classifier rows), against outputs of the REFERENCE itself on the same weights and lines
(tests/golden/stress.npz / stress.json, written by make_golden_stress.py from the imported reference).

  * CPU: the oracle against those fixtures (how much of the 1e-3 budget the checker itself uses on them);
  * GPU, through the C ABI, default arithmetic (f16x3 GEMMs, certified classifier pre-filter) and CRNN_F16X3=0 (f32
    MFMA): decoded strings and raw arg-max sequences identical, logits within 1e-3 of the reference's.
"""
import json
import os

import numpy as np
import pytest

from ocr_amd.crnn import keys
from ocr_amd.crnn.synth import STRESS_KINDS, stress_state_dict

LOGIT_TOL = 1e-3
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def stress():
    with open(os.path.join(HERE, "golden", "stress.json"), encoding="utf-8") as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "golden", "stress.npz")), meta


def test_stress_sets_are_what_they_claim():
    """The generator's own properties (CPU, no model run): spreads, outliers, duplicated rows."""
    base = stress_state_dict.__globals__["synth_state_dict"](5530, seed=0)
    a = stress_state_dict(5530, "chan_spread")
    g = np.abs(a["cnn.batchnorm2.weight"] / np.sqrt(a["cnn.batchnorm2.running_var"] + 1e-5))
    assert g.max() / g.min() > 1e5                                  # gamma / sqrt(var) spans ~1e-3 .. 1e3
    w3 = np.abs(a["cnn.conv3.weight"]).max(axis=(0, 2, 3))          # per input channel: the inverse spread
    assert w3.max() / w3.min() > 1e5
    b = stress_state_dict(5530, "outliers")
    w = np.abs(b["rnn.0.rnn.weight_hh_l0"]).ravel()
    big = w > 20 * np.median(w)
    assert 0.005 < big.mean() < 0.015
    assert abs(np.sqrt((b["cnn.conv5.weight"].astype(np.float64) ** 2).mean()) /
               np.sqrt((base["cnn.conv5.weight"].astype(np.float64) ** 2).mean()) - 1) < 0.2   # (the hot column moves it a little)
    c = stress_state_dict(5530, "cls_dup")
    W = c["rnn.1.embedding.weight"]
    assert (W[100:200] == W[100]).all() and not W[1000:1300].any()
    assert set(a) == set(base) == set(b) == set(c)


def test_load_time_equalisation_undoes_the_channel_spread():
    """crnn_load_weights (host-only: runs without a GPU) spends the network's per-channel rescaling freedom itself: the
    factors it reports for `chan_spread` must be the base set's factors with the injected 10^+-3 spread taken out again,
    to within the power-of-two rounding, i.e. both weight sets end up as the same equalised network."""
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import chan_spread_factors, synth_state_dict
    base = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
    base.load_state_dict(synth_state_dict(65))
    spread = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
    spread.load_state_dict(stress_state_dict(65, "chan_spread"))
    for layer in range(8):
        cb, cs = base.channel_scales(layer).astype(np.float64), spread.channel_scales(layer).astype(np.float64)
        assert (np.log2(cb) % 1 == 0).all() and (np.log2(cs) % 1 == 0).all()          # exact powers of two
        resid = np.log2(cs * chan_spread_factors(layer)) - np.log2(cb)
        assert np.abs(resid).max() <= 1.5, (layer, resid.min(), resid.max())
        if layer in (2, 4, 6):
            assert np.log2(cs).max() - np.log2(cs).min() >= 16                          # ~2^20 of spread removed
    off = None
    os.environ["CRNN_EQUALIZE"] = "0"
    try:
        off = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
        off.load_state_dict(stress_state_dict(65, "chan_spread"))
    finally:
        del os.environ["CRNN_EQUALIZE"]
    assert all((off.channel_scales(layer) == 1).all() for layer in range(8))


def test_oracle_vs_reference_on_stress_weights(stress):
    """CPU: the oracle (fp32 C restatement) on the stress fixtures, 65-class head on every line, big head on two short ones."""
    from oracle import oracle as O
    S, M = stress
    worst = 0.0
    for kind in ("chan_spread", "outliers"):
        ora = O.Oracle(stress_state_dict(65, kind), 65)
        for i, name in enumerate(M["names"]):
            ln = S["lines/" + name]
            lg = ora.forward(O.resize_normalize(ln)[None])[:, 0]
            e = float(np.abs(lg - S["%s/%s/en_logits" % (kind, name)]).max())
            worst = max(worst, e)
            assert e < LOGIT_TOL, (kind, name, e)
            assert O.decode(O.argmax(lg[:, None])[0], keys.alphabetEnglish) == M[kind]["en"][i]
    for kind in STRESS_KINDS:
        ora = O.Oracle(stress_state_dict(5530, kind), 5530)
        for name in ("inv_number", "noise100"):
            i = M["names"].index(name)
            ln = S["lines/" + name]
            lg = ora.forward(O.resize_normalize(ln)[None])[:, 0]
            e = float(np.abs(np.take_along_axis(lg, S["%s/%s/zh_cols" % (kind, name)].astype(np.int64), 1)
                             - S["%s/%s/zh_logits" % (kind, name)]).max())
            worst = max(worst, e)
            assert e < LOGIT_TOL, (kind, name, e)
            assert np.array_equal(O.argmax(lg[:, None])[0], S["%s/%s/zh_idx" % (kind, name)])
            assert O.decode(O.argmax(lg[:, None])[0], keys.alphabetChinese) == M[kind]["zh"][i]
    print("oracle vs reference, stress weights: worst |dlogit| %.2e" % worst)


def _model(nclass, kind, env):
    from ocr_amd.crnn.network_hip import CRNN
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)            # read by crnn_create
    try:
        m = CRNN(32, 1, nclass, 256, alphabet=keys.alphabetChinese if nclass > 100 else keys.alphabetEnglish, device=0)
        m.load_state_dict(stress_state_dict(nclass, kind))
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"CRNN_F16X3": "0"}], ids=["default", "f32mfma"])
@pytest.mark.parametrize("kind", STRESS_KINDS)
def test_hip_vs_reference_on_stress_weights(stress, kind, env):
    from ocr_amd.crnn import _capi
    S, M = stress
    names = M["names"]
    lines = [S["lines/" + n] for n in names]
    worst = {"en": 0.0, "zh": 0.0}
    zh = _model(5530, kind, env)
    for i, name in enumerate(names):
        raw, tt, lg, _, _ = zh.forward_logits([lines[i]])
        T = int(tt[0])
        got = np.take_along_axis(lg[0, :T], S["%s/%s/zh_cols" % (kind, name)].astype(np.int64), 1)
        e = float(np.abs(got - S["%s/%s/zh_logits" % (kind, name)]).max())
        worst["zh"] = max(worst["zh"], e)
        assert e < LOGIT_TOL, (kind, name, e)
        assert np.array_equal(raw[0, :T], S["%s/%s/zh_idx" % (kind, name)]), (kind, name)
    # the product path: logits not requested (classifier pre-filter on the default path), one call per line and one ragged batch
    want_idx = [S["%s/%s/zh_idx" % (kind, n)] for n in names]
    for i in range(len(names)):
        r = zh._run([lines[i]], _capi.PAD_PER_LINE, want_raw=True)[3]
        assert np.array_equal(r[0, :len(want_idx[i])], want_idx[i]), (kind, names[i])
    assert zh.predict_lines(lines) == M[kind]["zh"]
    r = zh._run(lines, _capi.PAD_PER_LINE, want_raw=True)[3]
    for i in range(len(names)):
        assert np.array_equal(r[i, :len(want_idx[i])], want_idx[i]), (kind, names[i], "batch")
    if kind != "cls_dup":
        en = _model(65, kind, env)
        for i, name in enumerate(names):
            raw, tt, lg, _, _ = en.forward_logits([lines[i]])
            e = float(np.abs(lg[0, :tt[0]] - S["%s/%s/en_logits" % (kind, name)]).max())
            worst["en"] = max(worst["en"], e)
            assert e < LOGIT_TOL, (kind, name, e)
        assert en.predict_lines(lines) == M[kind]["en"]
    print("HIP vs reference, %s, %s: worst |dlogit| en %.2e zh %.2e" % (kind, "/".join(env) or "default", worst["en"], worst["zh"]))


@pytest.mark.gpu
def test_cls_select_general_path_runs_on_ties(stress):
    """cls_dup at batch scale: 256 lines in one call, so cls_select_kernel's general loop (> 64 candidates: the 100
    duplicated rows, the 300 flat rows) decides thousands of rows; the result must equal the fp32 GEMM's arg-max
    (logits requested: no filter) everywhere, and the tied blocks must both have won somewhere with their LOWEST index."""
    S, M = stress
    zh = _model(5530, "cls_dup", {})
    rs = np.random.RandomState(11)
    lines = [rs.randint(0, 256, size=(32, 280)).astype(np.uint8) for _ in range(64)]
    from ocr_amd.crnn import _capi
    fast = zh._run(lines, _capi.PAD_PER_LINE, want_raw=True)[3]
    raw, tt, lg, _, _ = zh.forward_logits(lines)
    assert np.array_equal(fast, raw)
    assert (raw == 100).mean() > 0.05 and (raw == 1000).mean() > 0.05
    assert not np.isin(raw, np.r_[101:200, 1001:1300]).any()
    assert np.array_equal(lg.argmax(2), raw)          # numpy's arg-max: first maximum, like torch's on the CPU
