#!/usr/bin/env python3
"""Secondary measurements (BASELINE.json configs 3 and 5; config 2/4 are bench.py):
  config 3: English head (65 classes), widths drawn from {64,96,...,1024}, width-bucketed, per-line
            exact (masked) batches, device-resident lines; lines/s and padded-waste fraction.
  config 5: page-level stand-in.  Detectors / their weights / images/vat are absent from the reference
            snapshot, so a synthetic stream of BGR crops with the size distribution of the 17
            images/invoice crops (tests/golden) is replayed through CRNN.predict_lines (host crops ->
            gray -> Pillow-exact resize -> network -> strings); crops/s, PCIe-inclusive.
Prints one JSON object; run on an MI355X:  python tools/bench_configs.py [--precision bf16]
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--lines", type=int, default=16384)
    args = ap.parse_args()
    import torch
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.batching import padded_waste, width_buckets
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict
    dev = torch.device("cuda", 0)
    out = {"precision": args.precision}

    # ---------------- config 3
    en = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=4, precision=args.precision)
    en.load_state_dict(synth_state_dict(65, seed=0))
    rs = np.random.RandomState(3)
    widths = [int(w) for w in rs.choice(np.arange(64, 1025, 32), size=args.lines)]
    buckets = width_buckets(widths, max_batch=256, width_spread=1.25)
    waste = padded_waste(widths, buckets)
    # device-resident padded batches (uint8) + per-line widths
    batches = []
    for b in buckets:
        wp = max(widths[i] for i in b)
        g = torch.randint(0, 256, (len(b), 32, wp), dtype=torch.uint8, device=dev)
        T = wp // 4 + 1
        batches.append((g, [widths[i] for i in b], torch.zeros((len(b), T), dtype=torch.int32, device=dev),
                        torch.zeros((len(b),), dtype=torch.int32, device=dev)))

    def run3():
        for g, wv, lab, ln in batches:
            en.forward_device(g.data_ptr(), g.shape[0], g.shape[2], lab.data_ptr(), ln.data_ptr(), widths=wv)
        en.synchronize()
    run3()
    t0 = time.perf_counter(); run3(); el = time.perf_counter() - t0
    out["config3"] = {"lines": args.lines, "buckets": len(buckets), "padded_waste_frac": round(waste, 4),
                      "lines_per_s": round(args.lines / el, 1), "seconds": round(el, 3),
                      "mean_width": float(np.mean(widths)),
                      "note": "65-class head, widths uniform over {64..1024 step 32}, per-line masked batches (== predict)"}

    # ---------------- config 5 (crop replay)
    zh = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=3, precision=args.precision)
    zh.load_state_dict(synth_state_dict(5530, seed=0))
    G = np.load(os.path.join(ROOT, "tests", "golden", "crnn_golden.npz"))
    shapes = [G[k].shape for k in G.files if k.startswith("inv/") and k.endswith("/bgr")]
    ncrops = 2048
    crops = []
    for i in range(ncrops):
        h, w, _ = shapes[rs.randint(len(shapes))]
        crops.append(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8))
    zh.predict_lines(crops[:64])
    t0 = time.perf_counter(); texts = zh.predict_lines(crops); el = time.perf_counter() - t0
    out["config5_crop_replay"] = {"crops": ncrops, "crops_per_s": round(ncrops / el, 1), "seconds": round(el, 3),
                                  "note": "synthetic BGR crops with the 17 invoice crop sizes, host -> strings via "
                                          "predict_lines (H2D + gray + resize included); detectors absent, no pages/s"}
    assert len(texts) == ncrops

    # ---------------- preprocess stage alone (a1 + a2: BGR -> gray -> Pillow-exact resize -> normalise), device time
    big = [rs.randint(0, 256, size=(64, 560, 3)).astype(np.uint8) for _ in range(256)]
    zh.predict_lines(big)
    zh.set_profiling(True)
    zh.predict_lines(big)
    kt = {name: (ms, by) for name, ms, _fl, by in zh.kernel_times()}
    zh.set_profiling(False)
    pre_ms = kt["resize_h"][0] + kt["resize_v"][0]
    pre_by = kt["resize_h"][1] + kt["resize_v"][1]
    out["preprocess_stage"] = {"crops": 256, "crop": "64x560 BGR -> 32x280 f32",
                               "resize_h_ms": round(kt["resize_h"][0], 4), "resize_v_ms": round(kt["resize_v"][0], 4),
                               "algorithmic_GBps": round(pre_by / (pre_ms * 1e-3) / 1e9, 1),
                               "frac_of_hbm_peak": round(pre_by / (pre_ms * 1e-3) / 8e12, 4),
                               "note": "resize_h (gray conversion fused) + resize_v (normalise fused) of one 256-crop batch, HIP "
                                       "events around each launch; bytes = sources + 2 x uint8 intermediate + f32 output"}

    # ---------------- the reference's own call pattern: ocr_predict(crop), one crop at a time (ocr_context.py:119)
    one = [crops[i] for i in range(64)]
    for c in one[:8]:
        zh.ocr(c)
    t0 = time.perf_counter()
    for c in one:
        zh.ocr(c)
    el1 = time.perf_counter() - t0
    out["single_crop_calls"] = {"crops": len(one), "ms_per_call": round(el1 / len(one) * 1e3, 3),
                                "crops_per_s": round(len(one) / el1, 1),
                                "note": "CRNN.ocr(bgr) one crop per call, synchronous: bound by the T sequential recurrence steps"}

    # ---------------- config 5, page form: one upload per page, crops as rectangles
    npages, per_page = 24, 48
    ph, pw = 2339, 1654                       # A4 at 200 dpi
    pages, rects = [], []
    for _ in range(npages):
        pages.append(rs.randint(0, 256, size=(ph, pw, 3)).astype(np.uint8))
        rr = []
        for _ in range(per_page):
            h, w, _c = shapes[rs.randint(len(shapes))]
            x1, y1 = rs.randint(0, pw - w), rs.randint(0, ph - h)
            rr.append((x1, y1, x1 + w, y1 + h))
        rects.append(rr)
    zh.ocr_page(pages[0], rects[0])
    t0 = time.perf_counter()
    nt = sum(len(zh.ocr_page(pg, rr)) for pg, rr in zip(pages, rects))
    el = time.perf_counter() - t0
    # the same pages from two threads (the two request threads of a gunicorn worker): each call takes a
    # lane of its own, so two pages are on the device at once
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        list(ex.map(lambda a: zh.ocr_page(*a), list(zip(pages, rects))[:4]))
        t0 = time.perf_counter()
        n2 = sum(len(x) for x in ex.map(lambda a: zh.ocr_page(*a), zip(pages, rects)))
        el2 = time.perf_counter() - t0
    out["config5_page_rects_2_threads"] = {"pages_per_s": round(npages / el2, 1), "crops_per_s": round(n2 / el2, 1),
                                           "seconds": round(el2, 3)}
    out["config5_page_rects"] = {"pages": npages, "crops_per_page": per_page, "pages_per_s": round(npages / el, 1),
                                 "crops_per_s": round(nt / el, 1), "seconds": round(el, 3),
                                 "note": "synthetic 1654x2339 BGR pages, 48 rectangles of the invoice crop sizes each; "
                                         "page uploaded once (11.6 MB), gray+resize read the resident page; one "
                                         "GPU, one lane, detectors absent"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
