"""The CPU oracle (oracle/crnn_oracle.c) against fixtures captured from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
from PIL import Image

from ocr_amd.crnn import keys
from oracle import oracle as O

LOGIT_TOL = 1e-3  # north star: logits within 1e-3 of the reference torch CPU path


def test_alphabets():
    assert len(keys.alphabetChinese) == 5529 and len(keys.alphabetEnglish) == 64
    assert keys.alphabetEnglish.startswith("0123456789") and keys.alphabetEnglish[10] == "0"


def test_gray_and_resize_bit_exact(golden):
    G, M = golden
    for it in M["invoice"]:
        n = it["name"]
        gray = O.bgr_to_gray(G["inv/%s/bgr" % n])
        assert np.array_equal(gray, G["inv/%s/gray" % n]), n
        w = O.resized_width(gray.shape[1], gray.shape[0])
        assert w == it["w"]
        assert np.array_equal(O.resize_u8(gray, w, 32), G["inv/%s/resized" % n]), n
    for tag in ("upscale", "downscale"):
        gray = G["edge/%s/gray" % tag]
        w = O.resized_width(gray.shape[1], gray.shape[0])
        assert np.array_equal(O.resize_u8(gray, w, 32), G["edge/%s/resized" % tag]), tag


def test_bgr_noncontiguous_view(golden):
    G, _ = golden
    page = G["inv/title/bgr"]
    crop = page[5:60, 12:300]  # detectors pass slices of the page (vat_detect.py:291)
    assert np.array_equal(O.bgr_to_gray(crop), O.bgr_to_gray(np.ascontiguousarray(crop)))


@pytest.mark.parametrize("shape,out", [((7, 50), None), ((32, 280), None), ((100, 33), None), ((61, 999), None),
                                       ((33, 64), None), ((200, 1000), (17, 32)), ((5, 5), (40, 32))])
def test_resize_matches_installed_pillow(shape, out):
    """The Pillow restatement against the Pillow in this image (third-party, not the reference)."""
    rs = np.random.RandomState(shape[0] * 1000 + shape[1])
    a = rs.randint(0, 256, size=shape).astype(np.uint8)
    if out is None:
        out = (O.resized_width(shape[1], shape[0]), 32)
    want = np.array(Image.fromarray(a, mode="L").resize(out, Image.BILINEAR))
    assert np.array_equal(O.resize_u8(a, out[0], out[1]), want)


def test_normalize_lut():
    a = np.arange(256, dtype=np.uint8).reshape(1, 256)
    got = O.resize_normalize(np.repeat(a, 32, 0))
    want = ((np.repeat(a, 32, 0) / 255.0 - 0.5) / 0.5).astype(np.float32)
    assert np.array_equal(got, want)


def test_invoice_strings_and_logits(golden, oracle_zh, oracle_en):
    G, M = golden
    for it in M["invoice"]:
        n = it["name"]
        x = O.resize_normalize(G["inv/%s/gray" % n])
        lg = oracle_zh.forward(x[None])
        idx = O.argmax(lg)[0]
        assert np.array_equal(idx, G["inv/%s/zh_idx" % n]), n
        assert O.decode(idx, keys.alphabetChinese) == it["zh"], n
        assert np.abs(lg[:, 0].max(1) - G["inv/%s/zh_toplogit" % n]).max() < LOGIT_TOL
        le = oracle_en.forward(x[None])
        assert np.abs(le[:, 0] - G["inv/%s/en_logits" % n]).max() < LOGIT_TOL, n
        assert O.decode(O.argmax(le)[0], keys.alphabetEnglish) == it["en"], n
    lg = oracle_zh.forward(O.resize_normalize(G["inv/invoice_number/gray"])[None])
    assert np.abs(lg[:, 0] - G["inv/invoice_number/zh_logits"]).max() < LOGIT_TOL


def test_stage_taps(golden, oracle_zh):
    G, _ = golden
    x = O.resize_normalize(G["inv/tax/gray"])
    _, taps = oracle_zh.forward(x[None], stages=True)
    names = ["pooling0", "pooling1", "relu2", "pooling2", "relu4", "pooling3", "relu6"]
    for i, nm in enumerate(names):
        want = G["taps/tax/%s" % nm]
        got = taps[i][0][..., :24]
        assert got.shape == want.shape, nm
        assert np.abs(got - want).max() < 1e-4, nm
    assert np.abs(taps[8][:, 0] - G["taps/tax/lstm0_rec"]).max() < 1e-4
    assert np.abs(taps[7][:, 0] - G["taps/tax/lstm0_out"]).max() < 1e-3
    assert np.abs(taps[9][:, 0] - G["taps/tax/lstm1_rec"]).max() < 5e-4


def test_synthetic_280_batch(golden, oracle_zh):
    G, M = golden
    for tag in ("noise", "text"):
        lines = G["syn280/%s/gray" % tag]
        x = np.stack([O.resize_normalize(ln) for ln in lines])
        idx = O.argmax(oracle_zh.forward(x))
        assert np.array_equal(idx, G["syn280/%s/zh_idx" % tag]), tag
        assert [O.decode(i, keys.alphabetChinese) for i in idx] == M["syn280_%s" % tag]


def test_variable_width_english(golden, oracle_en):
    G, M = golden
    for w, s in zip(M["var_widths"], M["var_en"]):
        x = O.resize_normalize(G["var/%d/gray" % w])
        assert x.shape == (32, w)
        lg = oracle_en.forward(x[None])
        assert lg.shape[0] == w // 4 + 1
        assert np.abs(lg[:, 0] - G["var/%d/en_logits" % w]).max() < LOGIT_TOL
        assert O.decode(O.argmax(lg)[0], keys.alphabetEnglish) == s


def test_edges(golden, oracle_zh):
    G, M = golden
    assert O.predict_gray(oracle_zh, G["edge/narrow/gray"], keys.alphabetChinese) == "" == M["edge_narrow"]
    for tag in ("upscale", "downscale", "w8"):
        assert O.predict_gray(oracle_zh, G["edge/%s/gray" % tag], keys.alphabetChinese) == M["edge_%s" % tag], tag
    with pytest.raises(ZeroDivisionError):
        O.resize_normalize(np.zeros((0, 10), np.uint8))
    with pytest.raises(ValueError):
        O.resize_normalize(np.zeros((100, 2), np.uint8))  # w = int(2/3.125) = 0 -> Pillow raises


def test_ctc_collapse_rules():
    idx = np.array([0, 3, 3, 0, 3, 5, 5, 5, 0, 0, 1], np.int32)
    assert O.ctc_collapse(idx).tolist() == [2, 2, 4, 0]
    assert O.ctc_collapse(np.zeros(4, np.int32)).tolist() == []
    # ties -> lowest index (torch max on CPU)
    lg = np.zeros((1, 1, 7), np.float32)
    lg[0, 0, [2, 5]] = 1.0
    assert O.argmax(lg)[0, 0] == 2


def test_constructor_variants(golden):
    """lstmFlag=False (dense head) and leakyRelu=True against the reference's outputs."""
    from ocr_amd.crnn.synth import synth_state_dict
    G, M = golden
    V = M["variants"]
    dense = O.Oracle(synth_state_dict(65, seed=0, lstm=False), 65, lstm=False)
    leaky = O.Oracle(synth_state_dict(65, seed=0), 65, leaky=True)
    for i, n in enumerate(V["names"]):
        x = O.resize_normalize(G["inv/%s/gray" % n])
        for tag, orc in (("dense", dense), ("leaky", leaky)):
            lg = orc.forward(x[None])
            assert np.abs(lg[:, 0] - G["variants/%s/%s/logits" % (tag, n)]).max() < LOGIT_TOL, (tag, n)
            assert O.decode(O.argmax(lg)[0], keys.alphabetEnglish) == V[tag][i], (tag, n)
