"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden fixtures captured from the reference.  Needs an MI355X."""
import numpy as np
import pytest

from ocr_amd.crnn import keys

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3   # north star: fp32 logits within 1e-3 of the reference torch CPU path


@pytest.fixture(scope="module")
def hip_zh(synth_zh):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 5530, 256, leakyRelu=False, lstmFlag=True, GPU=True, alphabet=keys.alphabetChinese)
    m.load_state_dict(synth_zh)
    return m


@pytest.fixture(scope="module")
def hip_en(synth_en):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 65, 256, leakyRelu=False, lstmFlag=True, GPU=True, alphabet=keys.alphabetEnglish)
    m.load_state_dict(synth_en)
    return m


def nchw(a):
    return np.transpose(a, (0, 3, 1, 2))


def test_preprocess_bit_exact(golden, hip_zh):
    """gray + Pillow BILINEAR + normalise: integer work, must be bit-exact."""
    from oracle import oracle as O
    G, M = golden
    imgs, want = [], []
    for it in M["invoice"]:
        n = it["name"]
        imgs.append(G["inv/%s/bgr" % n])
        want.append(((G["inv/%s/resized" % n] / 255.0 - 0.5) / 0.5).astype(np.float32))
    for tag in ("upscale", "downscale", "w8", "narrow"):
        imgs.append(G["edge/%s/gray" % tag])
        want.append(O.resize_normalize(G["edge/%s/gray" % tag]))
    got = hip_zh.preprocess(imgs)
    for g, w, im in zip(got, want, imgs):
        assert g.shape == w.shape, (g.shape, w.shape, im.shape)
        assert np.array_equal(g, w)
    # a strided view into a "page" (detectors pass slices: vat_detect.py:318)
    page = G["inv/title/bgr"]
    crop = page[5:60, 12:300]
    g = hip_zh.preprocess([crop])[0]
    assert np.array_equal(g, O.resize_normalize(O.bgr_to_gray(crop)))


@pytest.mark.parametrize("shape", [(7, 50), (32, 280), (100, 33), (61, 999), (33, 64), (300, 900), (31, 31), (64, 4096),
                                   (70, 4100), (33, 5000)])   # the last two: wider than the row-staged kernel takes
def test_preprocess_random_sizes(shape, hip_zh):
    from oracle import oracle as O
    rs = np.random.RandomState(shape[0] * 1000 + shape[1])
    a = rs.randint(0, 256, size=shape).astype(np.uint8)
    assert np.array_equal(hip_zh.preprocess([a])[0], O.resize_normalize(a))
    c = rs.randint(0, 256, size=shape + (3,)).astype(np.uint8)
    assert np.array_equal(hip_zh.preprocess([c])[0], O.resize_normalize(O.bgr_to_gray(c)))


def test_stages_against_oracle(golden, hip_zh, oracle_zh):
    """Every stage of the network, layer by layer, on two lines of different content."""
    from oracle import oracle as O
    G, _ = golden
    x = np.stack([O.resize_normalize(G["syn280/noise/gray"][0]), O.resize_normalize(G["syn280/text/gray"][0])])
    logits, taps = oracle_zh.forward(x, stages=True)
    tol = [1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 2e-4, 1e-3, 2e-4, 1e-3]  # LSTM stages amplify rounding (|W| ~ 3/sqrt(in))
    for st in range(7):
        got = hip_zh.debug_stage(x, st)
        got = nchw(got) if st < 6 else np.transpose(got, (0, 2, 1))[:, :, None, :]
        assert got.shape == taps[st].shape, st
        assert np.abs(got - taps[st]).max() < tol[st], "stage %d" % st
    for st in (8, 7, 9):
        got = np.transpose(hip_zh.debug_stage(x, st), (1, 0, 2))
        assert np.abs(got - taps[st]).max() < tol[st], "stage %d" % st
    got = np.transpose(hip_zh.debug_stage(x, 10), (1, 0, 2))
    assert np.abs(got - logits).max() < LOGIT_TOL


def test_invoice_crops_match_reference(golden, hip_zh, hip_en):
    """Config 1: the 17 invoice crops, one at a time through CRNN.ocr (BGR in, text out)."""
    G, M = golden
    for it in M["invoice"]:
        n = it["name"]
        bgr = G["inv/%s/bgr" % n]
        assert hip_zh.ocr(bgr) == it["zh"], n
        assert hip_en.ocr(bgr) == it["en"], n
        raw, tt, logits, _, _ = hip_zh.forward_logits([bgr])
        assert tt[0] == it["T"]
        assert np.array_equal(raw[0, :tt[0]], G["inv/%s/zh_idx" % n]), n
        assert np.abs(logits[0, :tt[0]].max(1) - G["inv/%s/zh_toplogit" % n]).max() < LOGIT_TOL
        raw, tt, logits, _, _ = hip_en.forward_logits([bgr])
        assert np.abs(logits[0, :tt[0]] - G["inv/%s/en_logits" % n]).max() < LOGIT_TOL, n
    raw, tt, logits, _, _ = hip_zh.forward_logits([G["inv/invoice_number/bgr"]])
    assert np.abs(logits[0, :tt[0]] - G["inv/invoice_number/zh_logits"]).max() < LOGIT_TOL


def test_predict_pil(golden, hip_zh):
    from PIL import Image
    G, M = golden
    it = M["invoice"][0]
    img = Image.fromarray(G["inv/%s/gray" % it["name"]], mode="L")
    assert hip_zh.predict(img) == it["zh"]
    assert hip_zh.predict(Image.fromarray(G["edge/narrow/gray"], mode="L")) == "" == M["edge_narrow"]
    for tag in ("upscale", "downscale", "w8"):
        assert hip_zh.predict(Image.fromarray(G["edge/%s/gray" % tag], mode="L")) == M["edge_%s" % tag], tag
    with pytest.raises(ValueError):
        hip_zh.predict(Image.fromarray(np.zeros((100, 2), np.uint8), mode="L"))  # resized width 0
    with pytest.raises(ValueError):
        hip_zh.predict(Image.fromarray(np.zeros((10, 10, 3), np.uint8)))  # not mode 'L'
    with pytest.raises(ValueError):
        hip_zh.ocr(np.zeros((0, 0, 3), np.uint8))


def test_batched_lines_equal_single_predict(golden, hip_zh):
    """Mixed widths in ONE padded device batch decode exactly as each line alone (per-line masks)."""
    G, M = golden
    imgs = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]]
    want = [it["zh"] for it in M["invoice"]]
    imgs += [G["edge/narrow/gray"], G["edge/w8/gray"], G["edge/upscale/gray"]]
    want += [M["edge_narrow"], M["edge_w8"], M["edge_upscale"]]
    # one bucket with everything (worst padding), then the bucketed default
    got = hip_zh.predict_lines(imgs, max_batch=64, width_spread=1e9)
    assert got == want
    assert hip_zh.predict_lines(imgs) == want
    raw, tt, _, _, _ = hip_zh.forward_logits(imgs[:17])
    for i, it in enumerate(M["invoice"]):
        assert np.array_equal(raw[i, :tt[i]], G["inv/%s/zh_idx" % it["name"]]), it["name"]


def test_predict_batch_reference_semantics(golden, hip_zh, hip_en):
    """CRNN.predict_batch (network_torch.py:157-200): zero right padding, all padded timesteps decoded."""
    from PIL import Image
    G, M = golden
    pb = M["predict_batch"]
    boxes = [{"img": Image.fromarray(G["inv/%s/gray" % n], mode="L")} for n in pb["names"]]
    out = hip_zh.predict_batch(boxes, batch_size=pb["batch_size"])
    assert out is boxes and [b["text"] for b in out] == pb["zh"]
    pb = M["predict_batch_en"]
    boxes = [{"img": Image.fromarray(G["inv/%s/gray" % n], mode="L")} for n in pb["names"]]
    assert [b["text"] for b in hip_en.predict_batch(boxes, batch_size=pb["batch_size"])] == pb["en"]
    boxes = [{"img": Image.fromarray(G["inv/%s/gray" % n], mode="L")} for n in pb["names"]]
    assert [b["text"] for b in hip_en.predict_job(boxes)] == [
        it["en"] for n in pb["names"] for it in M["invoice"] if it["name"] == n]


def test_synthetic_280_and_variable_width(golden, hip_zh, hip_en):
    G, M = golden
    for tag in ("noise", "text"):
        lines = list(G["syn280/%s/gray" % tag])
        raw, tt, _, lab, ln = hip_zh.forward_logits(lines)
        assert (tt == 71).all()
        assert np.array_equal(raw[:, :71], G["syn280/%s/zh_idx" % tag]), tag
        assert hip_zh.predict_lines(lines) == M["syn280_%s" % tag]
    lines = [G["var/%d/gray" % w] for w in M["var_widths"]]
    assert hip_en.predict_lines(lines) == M["var_en"]
    raw, tt, logits, _, _ = hip_en.forward_logits(lines)
    for i, w in enumerate(M["var_widths"]):
        assert tt[i] == w // 4 + 1
        assert np.abs(logits[i, :tt[i]] - G["var/%d/en_logits" % w]).max() < LOGIT_TOL, w


def test_constructor_variants(golden):
    """lstmFlag=False and leakyRelu=True (network_torch.py:60-64, :85-86, :102-106)."""
    from PIL import Image
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict
    G, M = golden
    V = M["variants"]
    dense = CRNN(32, 1, 65, 256, leakyRelu=False, lstmFlag=False, GPU=True, alphabet=keys.alphabetEnglish)
    dense.load_state_dict(synth_state_dict(65, seed=0, lstm=False))
    leaky = CRNN(32, 1, 65, 256, leakyRelu=True, lstmFlag=True, GPU=True, alphabet=keys.alphabetEnglish)
    leaky.load_state_dict(synth_state_dict(65, seed=0))
    imgs = [G["inv/%s/gray" % n] for n in V["names"]]
    for tag, m in (("dense", dense), ("leaky", leaky)):
        for i, n in enumerate(V["names"]):
            raw, tt, logits, _, _ = m.forward_logits([imgs[i]], pad_mode=1)
            assert np.abs(logits[0, :tt[0]] - G["variants/%s/%s/logits" % (tag, n)]).max() < LOGIT_TOL, (tag, n)
            assert m.predict(Image.fromarray(imgs[i], mode="L")) == V[tag][i], (tag, n)
        assert m.predict_lines(imgs) == V[tag], tag     # mixed widths in one call stay exact
    boxes = [{"img": Image.fromarray(im, mode="L")} for im in imgs]
    assert [b["text"] for b in leaky.predict_batch(boxes, batch_size=3)] == V["leaky_predict_batch"]


def test_ocr_context_adapter(golden, hip_zh):
    """Batched CRNN branch for TextOcrModel.batch_ocr (ocr_context.py:302-310): same texts as one
    crop at a time, "" for None / unusable crops."""
    from ocr_amd.ocr_context import CrnnOcrContext
    G, M = golden
    ctx = CrnnOcrContext(hip_zh)
    crops = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]]
    images = crops[:5] + [None, np.zeros((0, 0, 3), np.uint8), np.zeros((90, 2, 3), np.uint8)] + crops[5:]
    want = [it["zh"] for it in M["invoice"]]
    want = want[:5] + ["", "", ""] + want[5:]
    assert ctx.batch_ocr(images) == want
    assert ctx.batch_ocr([]) == []
    assert ctx.ocr(crops[0]) == want[0] and ctx.ocr(None) == "" and ctx._chinese_ocr_instance(crops[1]) == want[1]


@pytest.mark.gpu
def test_load_weights_pth_with_module_prefix(tmp_path, golden, synth_en):
    """network_torch.py:110-120: a flat .pth whose keys carry the DataParallel 'module.' prefix."""
    import torch
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    path = str(tmp_path / "ocr-english.pth")
    torch.save({"module." + k: torch.from_numpy(np.asarray(v)) for k, v in synth_en.items()}, path)
    a = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0)
    a.load_weights(path)
    b = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0)
    b.load_state_dict(synth_en)
    G, M = golden
    lines = [G["var/%d/gray" % w] for w in M["var_widths"]]
    assert a.predict_lines(lines) == b.predict_lines(lines) == M["var_en"]


@pytest.mark.gpu
def test_page_rectangles_against_oracle(golden, hip_zh, oracle_zh):
    """crops given as rectangles of one resident page decode exactly as the reference decodes the same
    crops cut on the host (CRNN.ocr on page[y1:y2, x1:x2], ocr_context.py:119 fed by vat_detect.py:291,318).  The
    expected strings come from the CPU oracle on the host-cut crops and, for the pasted invoice crops, from the
    goldens captured from the reference itself -- not from the HIP path."""
    from oracle import oracle as orc
    from ocr_amd.ocr_context import CrnnOcrContext
    G, M = golden
    rs = np.random.RandomState(7)
    page = rs.randint(0, 256, size=(600, 900, 3)).astype(np.uint8)
    # paste the invoice crops so that real text-like content is on the page
    rects, gold, y = [], [], 5
    for it in sorted(M["invoice"], key=lambda it: it["name"])[:8]:
        c = G["inv/%s/bgr" % it["name"]]
        h, w = c.shape[:2]
        if y + h > page.shape[0] or w + 7 > page.shape[1]:
            continue
        page[y:y + h, 7:7 + w] = c
        rects.append((7, y, 7 + w, y + h))
        gold.append(it["zh"])
        y += h + 3
    assert len(gold) >= 4
    rects += [(100, 300, 420, 333), (0, 0, 900, 40), (850, 550, 2000, 2000), (10, 500, 14, 540), (300, 100, 308, 140)]

    def oracle_text(crop):
        g = orc.bgr_to_gray(crop) if crop.ndim == 3 else crop
        if orc.resize_normalize(g).shape[-1] < 8:
            return ""                                      # network_torch.py:141-142
        return orc.predict_gray(oracle_zh, g, keys.alphabetChinese)
    want = [oracle_text(page[y1:y2, x1:x2]) for x1, y1, x2, y2 in rects]
    assert want[:len(gold)] == gold                        # the oracle agrees with the reference on the pasted crops
    assert hip_zh.ocr_page(page, rects) == want
    assert [hip_zh.ocr(page[y1:y2, x1:x2]) for x1, y1, x2, y2 in rects] == want
    hip_zh.set_page(page)
    assert hip_zh.predict_rects(rects) == want
    # a strided view of a larger array as the page, gray pages, and the adapter's conventions
    big = np.zeros((700, 1000, 3), np.uint8)
    big[50:650, 60:960] = page
    assert hip_zh.ocr_page(big[50:650, 60:960], rects) == want
    gray = page[:, :, 1].copy()
    hip_zh.set_page(gray)
    assert hip_zh.predict_rects(rects[:4]) == [oracle_text(gray[y1:y2, x1:x2]) for x1, y1, x2, y2 in rects[:4]]
    ctx = CrnnOcrContext(hip_zh)
    got = ctx.page_ocr(page, rects + [(5, 5, 5, 30), (-3, 0, 10, 10)])
    assert got[:len(rects)] == want and got[len(rects):] == ["", ""]
    with pytest.raises(ZeroDivisionError):
        hip_zh.predict_rects([(5, 5, 50, 5)])


@pytest.mark.gpu
def test_forward_is_the_reference_forward(golden, hip_en, hip_zh):
    """CRNN.forward / __call__ (network_torch.py:88-108): normalised [b,1,32,w] in, logits [T,b,nclass] out,
    against the logits the reference's own forward gave (goldens)."""
    import torch
    G, M = golden
    for it in M["invoice"][:6]:
        n = it["name"]
        x = ((G["inv/%s/resized" % n] / 255.0 - 0.5) / 0.5).astype(np.float32)[None, None]
        out = hip_en(x)
        ref = G["inv/%s/en_logits" % n]
        assert out.shape == (ref.shape[0], 1, 65)
        assert np.abs(out[:, 0] - ref).max() < LOGIT_TOL, n
    n = "invoice_number"
    x = torch.from_numpy(((G["inv/%s/resized" % n] / 255.0 - 0.5) / 0.5).astype(np.float32)[None, None])
    out = hip_zh.forward(x)
    assert isinstance(out, torch.Tensor) and tuple(out.shape) == (G["inv/%s/zh_logits" % n].shape[0], 1, 5530)
    assert np.abs(out[:, 0].numpy() - G["inv/%s/zh_logits" % n]).max() < LOGIT_TOL
    # a batch of two equal-width inputs: column b of the output is sample b
    two = np.concatenate([x.numpy(), x.numpy()[:, :, :, ::-1]], 0)
    o2 = hip_zh(two)
    assert np.array_equal(o2[:, 0], out[:, 0].numpy()) and not np.array_equal(o2[:, 1], o2[:, 0])
    with pytest.raises(ValueError):
        hip_zh(np.zeros((1, 3, 32, 100), np.float32))


@pytest.mark.gpu
def test_random_crops_against_oracle(hip_en, oracle_en):
    """Seeded random crops (gray and BGR, heights 9..80, widths giving resized widths 8..~500) through the
    whole path against the CPU oracle line by line: strings and raw arg-max identical, logits within
    LOGIT_TOL.  Exercises odd widths at every layer (Winograd edge tiles, odd pooled widths, masks)."""
    from oracle import oracle as orc
    rs = np.random.RandomState(11)
    crops = []
    for k in range(24):
        h = int(rs.randint(9, 81))
        w = int(rs.randint(max(3, h // 4), h * 16))
        shape = (h, w, 3) if k % 3 == 0 else (h, w)
        a = rs.randint(0, 256, size=shape).astype(np.uint8)
        # some structure: dark strokes on a light background
        a[..., :] = np.where(rs.rand(*shape) < 0.8, 200 + a // 5, a // 3).astype(np.uint8)
        crops.append(a)
    got = hip_en.predict_lines(crops)
    raw, tt, logits, _, _ = hip_en.forward_logits(crops)
    from ocr_amd.crnn import keys
    for i, a in enumerate(crops):
        g = orc.bgr_to_gray(a) if a.ndim == 3 else a
        x = orc.resize_normalize(g)
        if x.shape[-1] < 8:
            assert got[i] == ""
            continue
        full = oracle_en.forward(x[None])            # [T, 1, nclass]
        idx, lg = orc.argmax(full)[0], full[:, 0, :]
        assert got[i] == orc.decode(idx, keys.alphabetEnglish), i
        assert tt[i] == len(idx) and np.array_equal(raw[i, :tt[i]], idx), i
        assert np.abs(logits[i, :tt[i]] - lg).max() < LOGIT_TOL, i


@pytest.mark.gpu
def test_width_edges_strings_against_oracle(hip_zh, hip_en, oracle_zh, oracle_en):
    """Widths around every tiling edge of the conv stack (4x4 / 2x2 Winograd tiles, tile-row segments of the pooled
    output transform, the 4-wide tiles of conv6's F(4,2), 128-row GEMM units) for both heads, one line per call and
    as one ragged batch: decoded strings identical to the oracle's (the 5530-class head goes through the classifier's
    bf16 pre-filter)."""
    from oracle import oracle as orc
    from ocr_amd.crnn import keys
    widths = [8, 9, 11, 12, 13, 15, 16, 17, 19, 20, 21, 31, 32, 33, 35, 63, 64, 65, 127, 128, 129, 255, 256, 257,
              259, 261, 263, 280, 511, 513, 1031]
    rs = np.random.RandomState(5)
    lines = [np.where(rs.rand(32, w) < 0.8, 200 + rs.randint(0, 50, (32, w)), rs.randint(0, 90, (32, w))).astype(np.uint8)
             for w in widths]
    for model, ora, alpha in ((hip_en, oracle_en, keys.alphabetEnglish), (hip_zh, oracle_zh, keys.alphabetChinese)):
        want = [orc.predict_gray(ora, ln, alpha) for ln in lines]
        assert [model.predict_lines([ln])[0] for ln in lines] == want
        assert model.predict_lines(lines, max_batch=16, width_spread=1.5) == want
