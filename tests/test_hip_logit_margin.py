"""The 1e-3 logit bound of the north star as a gate, on the inputs where the error is largest: fresh uniform-noise
lines over a sweep of widths (the synthetic model's random recurrent weights amplify rounding ~1000x on them).

  * HIP vs the CPU oracle, both heads, one line per call, widths that hit every tiling edge;
  * HIP vs the REFERENCE's own logits (tests/golden/noise_logits.npz, written by make_golden_noise.py from the
    imported reference) on ten noise lines of different widths: all 65 classes, and for the 5530-class head the 16
    largest classes of every timestep plus a fixed stride through all classes;
  * the oracle against the same reference logits (CPU, not marked gpu): how much of the budget the checker itself uses.

The worst figures are printed (pytest -s) and recorded in DESIGN.md with the commit that produced them."""
import os

import numpy as np
import pytest

from ocr_amd.crnn import keys

LOGIT_TOL = 1e-3
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def noise():
    return np.load(os.path.join(HERE, "golden", "noise_logits.npz"))


def _widths(noise):
    return sorted(int(k.split("/")[0][1:]) for k in noise.files if k.endswith("/gray"))


def test_oracle_vs_reference_noise_logits(noise, oracle_en, oracle_zh):
    """CPU: the oracle on the reference's hardest inputs (a subset for the big head: the oracle takes ~1 s a line)."""
    from oracle import oracle as O
    worst = 0.0
    for w in _widths(noise):
        x = O.resize_normalize(noise["w%d/gray" % w])[None]
        e = float(np.abs(oracle_en.forward(x)[:, 0] - noise["w%d/en_logits" % w]).max())
        worst = max(worst, e)
        assert e < LOGIT_TOL, ("en", w, e)
    for w in _widths(noise)[:4]:
        x = O.resize_normalize(noise["w%d/gray" % w])[None]
        lg = oracle_zh.forward(x)[:, 0]
        e = float(np.abs(np.take_along_axis(lg, noise["w%d/zh_cols" % w].astype(np.int64), 1) - noise["w%d/zh_logits" % w]).max())
        worst = max(worst, e)
        assert e < LOGIT_TOL, ("zh", w, e)
        assert np.array_equal(lg.argmax(1), noise["w%d/zh_idx" % w])
    print("oracle vs reference, noise lines: worst |dlogit| %.2e" % worst)


@pytest.mark.gpu
def test_hip_vs_reference_noise_logits(noise, synth_en, synth_zh):
    from ocr_amd.crnn.network_hip import CRNN
    en = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0)
    en.load_state_dict(synth_en)
    zh = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0)
    zh.load_state_dict(synth_zh)
    worst = {"en": 0.0, "zh": 0.0}
    for w in _widths(noise):
        ln = noise["w%d/gray" % w]
        raw, tt, lg, _, _ = en.forward_logits([ln])
        e = float(np.abs(lg[0, :tt[0]] - noise["w%d/en_logits" % w]).max())
        worst["en"] = max(worst["en"], e)
        assert e < LOGIT_TOL, ("en", w, e)
        raw, tt, lg, _, _ = zh.forward_logits([ln])
        got = np.take_along_axis(lg[0, :tt[0]], noise["w%d/zh_cols" % w].astype(np.int64), 1)
        e = float(np.abs(got - noise["w%d/zh_logits" % w]).max())
        worst["zh"] = max(worst["zh"], e)
        assert e < LOGIT_TOL, ("zh", w, e)
        assert np.array_equal(raw[0, :tt[0]], noise["w%d/zh_idx" % w]), w
        # the product path (classifier pre-filter, logits not requested) decides the same classes
        from ocr_amd.crnn import _capi
        assert np.array_equal(zh._run([ln], _capi.PAD_PER_LINE, want_raw=True)[3][0, :tt[0]], noise["w%d/zh_idx" % w]), w
    print("HIP vs reference, noise lines: worst |dlogit| en %.2e zh %.2e" % (worst["en"], worst["zh"]))


@pytest.mark.gpu
def test_hip_vs_oracle_width_sweep(synth_en, synth_zh, oracle_en, oracle_zh):
    """Fresh noise lines (not in any fixture): worst |logit - oracle| < 1e-3 over the sweep, for both heads."""
    from oracle import oracle as O
    from ocr_amd.crnn.network_hip import CRNN
    rs = np.random.RandomState(7)
    worst = {}
    for tag, nclass, alpha, sd, ora, widths in (
            ("en", 65, keys.alphabetEnglish, synth_en, oracle_en, list(range(8, 331, 13)) + [511, 513, 767, 1031]),
            ("zh", 5530, keys.alphabetChinese, synth_zh, oracle_zh, list(range(29, 331, 21)) + [512])):
        m = CRNN(32, 1, nclass, 256, alphabet=alpha, device=0)
        m.load_state_dict(sd)
        worst[tag] = 0.0
        for w in widths:
            ln = rs.randint(0, 256, size=(32, w)).astype(np.uint8)
            raw, tt, lg, _, _ = m.forward_logits([ln])
            ref = ora.forward(O.resize_normalize(ln)[None])[:, 0]
            e = float(np.abs(lg[0, :tt[0]] - ref).max())
            worst[tag] = max(worst[tag], e)
            assert e < LOGIT_TOL, (tag, w, e)
            assert m.predict_lines([ln])[0] == O.decode(O.argmax(ref[:, None])[0], alpha), (tag, w)
    print("HIP vs oracle, width sweep: worst |dlogit| en %.2e zh %.2e" % (worst["en"], worst["zh"]))
