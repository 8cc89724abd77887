"""Stress the shared-handle host path: K threads call CRNN.ocr on the invoice crops of the goldens for many rounds
and every string is compared with the reference's golden string.  Prints one JSON line; exit status 1 on any mismatch.
(This is what found the packed-fp32 / MFMA co-run problem described in ocr_amd/csrc/crnn_kernels_nopk.hip: 10-20
wrong strings in 25 000 calls from two threads before the fix, none after.)

    python tools/stress_threads.py [--threads 2] [--rounds 500] [--lanes 3]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=500)
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--own-handle", type=int, default=0, help="1: every thread has a model object of its own")
    ap.add_argument("--shuffle", type=int, default=0, help="each thread visits its crops in a random order (seed)")
    ap.add_argument("--torch-noise", type=int, default=0, help="1: one more thread keeps small torch kernels (element-wise "
                    "compare + reduction on a stream of its own) on the device, the neighbours that exposed the AccVGPR "
                    "co-run fault of round 3 (DESIGN.md section 4)")
    args = ap.parse_args()
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict
    G = np.load(os.path.join(ROOT, "tests", "golden", "crnn_golden.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "crnn_golden.json"), encoding="utf-8") as f:
        M = json.load(f)
    crops = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]]
    want = [it["zh"] for it in M["invoice"]]
    sd = synth_state_dict(5530, seed=0)
    models = []
    for _ in range(args.threads if args.own_handle else 1):
        mm = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, lanes=args.lanes)
        mm.load_state_dict(sd)
        models.append(mm)
    m = models[0]
    bad = []
    calls = [0] * args.threads

    def work(k):
        rng = np.random.default_rng(args.shuffle + k)
        m = models[k % len(models)]
        prev = -1
        for r in range(args.rounds):
            idx = list(range(k, len(crops), args.threads))
            if args.shuffle:
                idx = [int(j) for j in rng.permutation(len(crops))]
            for i in idx:
                got = m.ocr(crops[i])
                calls[k] += 1
                if got != want[i]:
                    bad.append({"thread": k, "round": r, "crop": i, "prev": prev, "w": int(crops[i].shape[1]),
                                "got": got, "want": want[i]})
                    if len(bad) > 20:
                        return
                prev = i

    stop = []
    noise_n = [0]

    def noise():
        import torch
        dev = torch.device("cuda", 0)
        a = torch.zeros((256, 71), dtype=torch.int32, device=dev)
        b = torch.zeros((256, 71), dtype=torch.int32, device=dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            while not stop:
                for _ in range(50):
                    (a != b).any()
                torch.cuda.current_stream().synchronize()
                noise_n[0] += 50

    t0 = time.time()
    th = [threading.Thread(target=work, args=(k,)) for k in range(args.threads)]
    nt = threading.Thread(target=noise) if args.torch_noise else None
    if nt:
        nt.start()
    for t in th:
        t.start()
    for t in th:
        t.join()
    stop.append(1)
    if nt:
        nt.join()
    out = {"threads": args.threads, "lanes": args.lanes, "calls": sum(calls), "mismatches": len(bad),
           "seconds": round(time.time() - t0, 2), "first": bad[:3], "torch_noise_kernels": 2 * noise_n[0]}
    if hasattr(m, "split_timeouts"):
        try:
            out["split_timeouts"] = int(m.split_timeouts())
        except Exception:  # noqa: BLE001
            pass
    print(json.dumps(out, ensure_ascii=False))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
