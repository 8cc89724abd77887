"""Reference outputs on the STRESS weight sets (ocr_amd/crnn/synth.py stress_state_dict).

Same rules as make_golden.py (whose helpers it uses): runs only in the build container, imports the reference's
crnn/network_torch.py as it lies in /root/reference (nothing copied), fills it with the build's own seeded weights and
records inputs and the reference's outputs as plain numpy data in tests/golden/stress.npz + stress.json:

  lines/<name>                 uint8 [32, W]   the input lines (two invoice crops, noise and text-like synthetic lines)
  <kind>/<name>/en_logits      f32 [T, 65]     reference forward(), 65-class head (chan_spread, outliers)
  <kind>/<name>/zh_cols        int32 [T, 257]  the 16 classes with the largest reference logits, then classes 0, 23, 46, ...
  <kind>/<name>/zh_logits      f32 [T, 257]    reference forward(), 5530-class head, at those classes
  <kind>/<name>/zh_idx         int32 [T]       reference arg-max
  stress.json                  the reference's decoded strings per kind / head / line, and for cls_dup the share of
                               timesteps won by the duplicated block (class 100) and by the flat block (class 1000)

Usage:  python tests/golden/make_golden_stress.py
"""
import json
import os

import numpy as np
import torch
from PIL import Image

import make_golden as mg   # imports the reference (see its header)
from ocr_amd.crnn.synth import STRESS_KINDS, stress_state_dict


def build_ref(nclass, alphabet, kind):
    m = mg.RefCRNN(32, 1, nclass, 256, leakyRelu=False, lstmFlag=True, GPU=False, alphabet=alphabet)
    sd = stress_state_dict(nclass, kind, seed=mg.SEED)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    m.eval()
    return m


def main():
    G = np.load(os.path.join(mg.HERE, "crnn_golden.npz"))
    lines = {"inv_tax": G["inv/tax/gray"], "inv_number": G["inv/invoice_number/gray"],
             "noise280a": mg.synth_lines(8101, 1, 280)[0], "noise280b": mg.synth_lines(8102, 1, 280)[0],
             "text280": mg.synth_lines(8103, 1, 280, textlike=True)[0], "noise100": mg.synth_lines(8104, 1, 100)[0]}
    arrays = {"lines/" + k: v for k, v in lines.items()}
    meta = {"seed": mg.SEED, "torch": torch.__version__, "names": list(lines), "kinds": list(STRESS_KINDS)}
    stride = np.arange(0, 5530, 23, dtype=np.int32)
    for kind in STRESS_KINDS:
        zh = build_ref(5530, mg.ref_keys.alphabetChinese, kind)
        en = build_ref(65, mg.ref_keys.alphabetEnglish, kind) if kind != "cls_dup" else None
        meta[kind] = {"zh": [], "en": []}
        won = np.zeros(3)
        margins = []
        for name, ln in lines.items():
            _, lz = mg.ref_logits(zh, ln)
            top = np.argsort(-lz, axis=1, kind="stable")[:, :16].astype(np.int32)
            cols = np.concatenate([top, np.broadcast_to(stride, (lz.shape[0], len(stride)))], 1)
            arrays["%s/%s/zh_cols" % (kind, name)] = cols
            arrays["%s/%s/zh_logits" % (kind, name)] = np.take_along_axis(lz, cols, 1).astype(np.float32)
            idx = lz.argmax(1).astype(np.int32)
            arrays["%s/%s/zh_idx" % (kind, name)] = idx
            meta[kind]["zh"].append(zh.predict(Image.fromarray(ln, mode="L")))
            srt = np.sort(lz, 1)
            margins.append(srt[:, -1] - srt[:, -2])
            if kind == "cls_dup":
                # the reference's logits of identical rows must be identical bits, or "ties -> lowest index" is not what it tests
                assert (lz[:, 100:200] == lz[:, 100:101]).all() and (lz[:, 1000:1300] == lz[:, 1000:1001]).all()
                assert not np.isin(idx, np.r_[101:200, 1001:1300]).any()
                won += [(idx == 100).sum(), (idx == 1000).sum(), len(idx)]
            if en is not None:
                _, le = mg.ref_logits(en, ln)
                arrays["%s/%s/en_logits" % (kind, name)] = le.astype(np.float32)
                meta[kind]["en"].append(en.predict(Image.fromarray(ln, mode="L")))
        mar = np.concatenate(margins)
        if kind == "cls_dup":
            nz = mar[mar > 0]
            meta[kind]["share_dup_block"] = round(float(won[0] / won[2]), 3)
            meta[kind]["share_flat_block"] = round(float(won[1] / won[2]), 3)
            meta[kind]["min_nonzero_margin"] = float(nz.min())
            print(kind, "dup block wins %.3f, flat block wins %.3f of %d timesteps" % (won[0] / won[2], won[1] / won[2], won[2]))
            assert won[0] / won[2] > 0.08 and won[1] / won[2] > 0.08
        else:
            meta[kind]["min_margin"] = float(mar.min())
        print(kind, "zh strings:", [len(s) for s in meta[kind]["zh"]], "distinct chars",
              len(set("".join(meta[kind]["zh"]))), "min top-2 margin %.3g" % mar[mar > 0].min())
    np.savez_compressed(os.path.join(mg.HERE, "stress.npz"), **arrays)
    with open(os.path.join(mg.HERE, "stress.json"), "w", encoding="utf-8") as f:
        json.dump(meta, f, ensure_ascii=False, indent=1)
    print("wrote", len(arrays), "arrays;", os.path.getsize(os.path.join(mg.HERE, "stress.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
