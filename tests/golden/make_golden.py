"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference and torch CPU).  It
imports the reference's crnn/network_torch.py, crnn/util.py and crnn/keys.py
as they lie in /root/reference -- nothing from the reference is copied -- and
records inputs and the reference's outputs as plain numpy data.

`loguru` and `cv2` are absent from this image; network_torch.py imports them
at module level (:7, :15) but uses them only inside CRNN.ocr (:124, :127),
which this script never calls.  Two empty placeholder modules are put in
sys.modules so the import statement succeeds; they provide no behaviour.
CRNN.ocr's one arithmetic line (BGR -> RGB -> PIL 'L') is reproduced with
Pillow directly.

The trained checkpoints are not in the reference snapshot
(.MISSING_LARGE_BLOBS:7-8), so the model is filled from the build's
procedural weights (ocr_amd/crnn/synth.py), seed recorded in the fixture.

Usage:  python tests/golden/make_golden.py
"""
import glob
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

for _m in ("loguru", "cv2"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["loguru"].logger = None
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from PIL import Image  # noqa: E402

from crnn import keys as ref_keys  # noqa: E402  (reference)
from crnn.network_torch import CRNN as RefCRNN  # noqa: E402  (reference)
from crnn.util import resizeNormalize as ref_resizeNormalize  # noqa: E402  (reference)
from ocr_amd.crnn.synth import synth_state_dict  # noqa: E402

torch.manual_seed(0)
SEED = 0


def build_ref(nclass, alphabet, lstm=True, leaky=False):
    m = RefCRNN(32, 1, nclass, 256, leakyRelu=leaky, lstmFlag=lstm, GPU=False, alphabet=alphabet)
    sd = synth_state_dict(nclass, seed=SEED, lstm=lstm)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    assert not missing.unexpected_keys
    m.eval()  # network_torch.py:120 does this inside load_weights
    return m


def ref_logits(m, gray_u8):
    """What CRNN.predict (:130-147) computes, with the logits kept."""
    img = Image.fromarray(gray_u8, mode="L")
    x = ref_resizeNormalize(img, 32).astype(np.float32)
    xt = torch.from_numpy(x).view(1, 1, *x.shape)
    with torch.no_grad():
        out = m(xt)  # [T,1,nclass]
    return x, out[:, 0].numpy()


def top2(logits):
    part = np.sort(logits, axis=1)
    return part[:, -1] - part[:, -2]


def synth_lines(seed, n, w, textlike=False):
    rs = np.random.RandomState(seed)
    if not textlike:
        return rs.randint(0, 256, size=(n, 32, w)).astype(np.uint8)
    out = np.full((n, 32, w), 255, np.uint8)
    for i in range(n):
        for _ in range(w // 6):
            x0 = rs.randint(0, w - 3)
            y0 = rs.randint(2, 24)
            ww, hh = rs.randint(1, 5), rs.randint(2, 8)
            out[i, y0:y0 + hh, x0:x0 + ww] = rs.randint(0, 90)
    return out


def main():
    zh = build_ref(5530, ref_keys.alphabetChinese)
    en = build_ref(65, ref_keys.alphabetEnglish)
    meta = {"seed": SEED, "torch": torch.__version__, "pillow": Image.__version__ if hasattr(Image, "__version__") else "",
            "invoice": [], "notes": "weights = ocr_amd.crnn.synth.synth_state_dict(nclass, seed)"}
    arrays = {}

    # ---- config 1: the 17 images/invoice crops, one at a time -----------------
    files = sorted(glob.glob(os.path.join(REF, "images/invoice/*.png")))
    assert len(files) == 17
    for f in files:
        name = os.path.splitext(os.path.basename(f))[0]
        rgb = np.array(Image.open(f).convert("RGB"))
        bgr = np.ascontiguousarray(rgb[:, :, ::-1])  # what the detectors hand to CRNN.ocr
        pil_l = Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).convert("L")  # :124
        gray = np.array(pil_l)
        x, lg_zh = ref_logits(zh, gray)
        _, lg_en = ref_logits(en, gray)
        resized = np.array(pil_l.resize((x.shape[1], 32), Image.BILINEAR))
        s_zh = zh.predict(pil_l)
        s_en = en.predict(pil_l)
        arrays["inv/%s/bgr" % name] = bgr
        arrays["inv/%s/gray" % name] = gray
        arrays["inv/%s/resized" % name] = resized
        arrays["inv/%s/zh_idx" % name] = lg_zh.argmax(1).astype(np.int32)
        arrays["inv/%s/zh_margin" % name] = top2(lg_zh).astype(np.float32)
        arrays["inv/%s/zh_toplogit" % name] = lg_zh.max(1).astype(np.float32)
        arrays["inv/%s/en_idx" % name] = lg_en.argmax(1).astype(np.int32)
        arrays["inv/%s/en_logits" % name] = lg_en.astype(np.float32)
        meta["invoice"].append({"name": name, "w": int(x.shape[1]), "T": int(lg_zh.shape[0]), "zh": s_zh, "en": s_en})
        assert np.array_equal(((resized / 255.0 - 0.5) / 0.5).astype(np.float32), x)

    # full logits of the 5530-class head for one short crop
    g = arrays["inv/invoice_number/gray"]
    _, lg = ref_logits(zh, g)
    arrays["inv/invoice_number/zh_logits"] = lg.astype(np.float32)

    # ---- stage taps for one crop (layer-by-layer debugging) -----------------
    x, _ = ref_logits(zh, arrays["inv/tax/gray"])
    with torch.no_grad():
        h = torch.from_numpy(x)[None, None]
        taps = {}
        for nm, mod in zh.cnn.named_children():
            h = mod(h)
            taps[nm] = h
        for nm in ("pooling0", "pooling1", "relu2", "pooling2", "relu4", "pooling3", "relu6"):
            arrays["taps/tax/%s" % nm] = taps[nm][0].numpy().astype(np.float32)[..., :24].copy()
        seq = taps["relu6"].squeeze(2).permute(2, 0, 1)
        r0, _ = zh.rnn[0].rnn(seq)
        arrays["taps/tax/lstm0_rec"] = r0[:, 0].numpy().astype(np.float32)
        m0 = zh.rnn[0](seq)
        arrays["taps/tax/lstm0_out"] = m0[:, 0].numpy().astype(np.float32)
        r1, _ = zh.rnn[1].rnn(m0)
        arrays["taps/tax/lstm1_rec"] = r1[:, 0].numpy().astype(np.float32)

    # ---- config 2: synthetic 32x280 lines (uniform noise + text-like) ----------
    for tag, lines in (("noise", synth_lines(1234, 6, 280)), ("text", synth_lines(4321, 6, 280, textlike=True))):
        arrays["syn280/%s/gray" % tag] = lines
        idx, mar, strs = [], [], []
        for ln in lines:
            _, lgz = ref_logits(zh, ln)
            idx.append(lgz.argmax(1).astype(np.int32))
            mar.append(top2(lgz).astype(np.float32))
            strs.append(zh.predict(Image.fromarray(ln, mode="L")))
        arrays["syn280/%s/zh_idx" % tag] = np.stack(idx)
        arrays["syn280/%s/zh_margin" % tag] = np.stack(mar)
        meta["syn280_%s" % tag] = strs
    # the whole 6-line batch through forward at once (batch == per-line when widths are equal)
    xb = torch.from_numpy(((arrays["syn280/noise/gray"] / 255.0 - 0.5) / 0.5).astype(np.float32))[:, None]
    with torch.no_grad():
        ob = zh(xb)
    assert np.array_equal(ob.argmax(2).numpy().T, arrays["syn280/noise/zh_idx"])

    # ---- config 3: English head, variable widths 64..1024 -------------------
    rs = np.random.RandomState(99)
    widths = [64, 96, 160, 352, 608, 1024]
    meta["var_widths"] = widths
    strs = []
    for i, w in enumerate(widths):
        ln = synth_lines(500 + i, 1, w, textlike=bool(i % 2))[0]
        arrays["var/%d/gray" % w] = ln
        _, lge = ref_logits(en, ln)
        arrays["var/%d/en_logits" % w] = lge.astype(np.float32)
        strs.append(en.predict(Image.fromarray(ln, mode="L")))
    meta["var_en"] = strs

    # ---- predict_batch (:157-200): zero-right-padded chunks -----------------
    names = ["tax", "invoice_number", "total", "seal_1", "bill_date"]
    boxes = [{"img": Image.fromarray(arrays["inv/%s/gray" % n], mode="L")} for n in names]
    with torch.no_grad():
        out = zh.predict_batch(boxes, batch_size=2)
    meta["predict_batch"] = {"names": names, "batch_size": 2, "zh": [b["text"] for b in out]}
    boxes = [{"img": Image.fromarray(arrays["inv/%s/gray" % n], mode="L")} for n in names]
    with torch.no_grad():
        out = en.predict_batch(boxes, batch_size=5)
    meta["predict_batch_en"] = {"names": names, "batch_size": 5, "en": [b["text"] for b in out]}

    # ---- edge cases -------------------------------------------------------
    # resized width < 8 -> '' (:141-142); tall narrow crop
    tall = synth_lines(7, 1, 40)[0][:, :6]  # 32x6 -> w=6
    arrays["edge/narrow/gray"] = tall
    meta["edge_narrow"] = zh.predict(Image.fromarray(tall, mode="L"))
    # up-scaling crop (H<32) and a strongly down-scaled one
    small = synth_lines(8, 1, 90)[0][:13, :]  # 13x90
    arrays["edge/upscale/gray"] = small
    meta["edge_upscale"] = zh.predict(Image.fromarray(small, mode="L"))
    x, lgz = ref_logits(zh, small)
    arrays["edge/upscale/zh_idx"] = lgz.argmax(1).astype(np.int32)
    arrays["edge/upscale/resized"] = np.array(Image.fromarray(small, mode="L").resize((x.shape[1], 32), Image.BILINEAR))
    big = np.kron(synth_lines(9, 1, 50)[0], np.ones((5, 3), np.uint8))[:151]  # 151x150
    arrays["edge/downscale/gray"] = big
    meta["edge_downscale"] = zh.predict(Image.fromarray(big, mode="L"))
    x, lgz = ref_logits(zh, big)
    arrays["edge/downscale/resized"] = np.array(Image.fromarray(big, mode="L").resize((x.shape[1], 32), Image.BILINEAR))
    arrays["edge/downscale/zh_idx"] = lgz.argmax(1).astype(np.int32)
    # width exactly 8 after resize (T = 3)
    w8 = synth_lines(10, 1, 8)[0]
    arrays["edge/w8/gray"] = w8
    meta["edge_w8"] = zh.predict(Image.fromarray(w8, mode="L"))

    # ---- constructor variants (network_torch.py:60-64 leakyRelu, :85-86/:102-106 lstmFlag=False)
    dense = build_ref(65, ref_keys.alphabetEnglish, lstm=False)
    leaky = build_ref(65, ref_keys.alphabetEnglish, lstm=True, leaky=True)
    vnames = ["tax", "invoice_number", "seal_1"]
    meta["variants"] = {"names": vnames, "dense": [], "leaky": []}
    for n in vnames:
        g = arrays["inv/%s/gray" % n]
        for tag, mdl in (("dense", dense), ("leaky", leaky)):
            _, lgv = ref_logits(mdl, g)
            arrays["variants/%s/%s/logits" % (tag, n)] = lgv.astype(np.float32)
            meta["variants"][tag].append(mdl.predict(Image.fromarray(g, mode="L")))
    boxes = [{"img": Image.fromarray(arrays["inv/%s/gray" % n], mode="L")} for n in vnames]
    with torch.no_grad():
        meta["variants"]["leaky_predict_batch"] = [b["text"] for b in leaky.predict_batch(boxes, batch_size=3)]

    np.savez_compressed(os.path.join(HERE, "crnn_golden.npz"), **arrays)
    with open(os.path.join(HERE, "crnn_golden.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, ensure_ascii=False, indent=1)
    print("wrote", len(arrays), "arrays;", os.path.getsize(os.path.join(HERE, "crnn_golden.npz")) / 1e6, "MB")
    mm = min(float(v.min()) for k, v in arrays.items() if k.endswith("margin"))
    print("min top-2 margin over all golden timesteps: %.3g" % mm)


if __name__ == "__main__":
    main()
