"""bf16 path (bf16 MFMA inputs, fp32 accumulate): closeness to the fp32 oracle stage by stage and
character-level agreement with the reference's strings.  Not held to the 1e-3 logit bound: the
north star asks this path to REPORT character agreement.  Needs an MI355X."""
import numpy as np
import pytest

from ocr_amd.crnn import keys

pytestmark = pytest.mark.gpu


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def agreement(got, want):
    err = sum(edit_distance(g, w) for g, w in zip(got, want))
    tot = sum(len(w) for w in want)
    return 1.0 - err / max(1, tot)


@pytest.fixture(scope="module")
def bf16_en(synth_en):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, precision="bf16")
    m.load_state_dict(synth_en)
    return m


@pytest.fixture(scope="module")
def bf16_zh(synth_zh):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, precision="bf16")
    m.load_state_dict(synth_zh)
    return m


def test_bf16_conv_stages_close_to_fp32_oracle(golden, bf16_zh, oracle_zh):
    from oracle import oracle as O
    G, _ = golden
    x = np.stack([O.resize_normalize(G["syn280/noise/gray"][0]), O.resize_normalize(G["syn280/text/gray"][0])])
    _, taps = oracle_zh.forward(x, stages=True)
    for st in range(7):
        got = bf16_zh.debug_stage(x, st)
        got = np.transpose(got, (0, 3, 1, 2)) if st < 6 else np.transpose(got, (0, 2, 1))[:, :, None, :]
        ref = taps[st]
        rel = np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())
        assert rel < 3e-2, "stage %d rel err %.3g" % (st, rel)   # bf16 has 8 significant bits
    # first recurrent layer output: bounded in (-1, 1); mean abs deviation small
    got = np.transpose(bf16_zh.debug_stage(x, 8), (1, 0, 2))
    assert np.abs(got - taps[8]).mean() < 3e-2


def test_bf16_character_agreement_report(golden, bf16_zh, bf16_en):
    """Character agreement of the bf16 path with the reference strings (synthetic weights whose
    recurrent matrices amplify rounding; trained weights are far better conditioned)."""
    G, M = golden
    crops = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]]
    a_zh = agreement(bf16_zh.predict_lines(crops), [it["zh"] for it in M["invoice"]])
    a_en = agreement(bf16_en.predict_lines(crops), [it["en"] for it in M["invoice"]])
    lines = list(G["syn280/noise/gray"]) + list(G["syn280/text/gray"])
    a_syn = agreement(bf16_zh.predict_lines(lines), M["syn280_noise"] + M["syn280_text"])
    print("\nbf16 character agreement: invoice zh %.3f, invoice en %.3f, synthetic-280 zh %.3f" % (a_zh, a_en, a_syn))
    # measured on the round-3 build: 0.800 / 0.834 / 0.864 (deterministic for a given build); the floor leaves a small margin
    assert a_zh >= 0.75 and a_en >= 0.78 and a_syn >= 0.80
    # deterministic
    assert bf16_zh.predict_lines(crops) == bf16_zh.predict_lines(crops)


def test_bf16_batched_equals_single(golden, bf16_en):
    """Per-line masking is exact on the bf16 path too: a padded batch decodes as each line alone."""
    G, M = golden
    crops = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]][:8]
    one = [bf16_en.ocr(c) for c in crops]
    assert bf16_en.predict_lines(crops, width_spread=1e9) == one
