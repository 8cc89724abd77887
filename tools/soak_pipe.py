"""Soak of the PIPELINED device path (two lanes, batches queued back to back, no sync): every batch's raw arg-max rows are
compared on the device with those of a synchronous call on the same input; mismatching (line, timestep) positions are
listed.  Prints one JSON line; exit status 1 on any mismatch.

    python tools/soak_pipe.py [--batches 20000] [--lines 256] [--width 280] [--lanes 2] [--slots 4] [--precision fp32|bf16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=20000)
    ap.add_argument("--lines", type=int, default=256)
    ap.add_argument("--width", type=int, default=280)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--nclass", type=int, default=5530, help="5530: the Chinese head, 65: the English head")
    ap.add_argument("--host-compare", type=int, default=0, help="1: copy every batch's rows back and compare on the host (no "
                    "comparison kernels of torch beside the library's kernels on the device)")
    ap.add_argument("--ragged", type=int, default=0, help="1: per-line valid widths (random, multiples of 4) in every batch")
    args = ap.parse_args()
    import torch
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict
    dev = torch.device("cuda", 0)
    alpha = keys.alphabetChinese if args.nclass == 5530 else keys.alphabetEnglish
    m = CRNN(32, 1, args.nclass, 256, alphabet=alpha, device=0, lanes=args.lanes, precision=args.precision)
    m.load_state_dict(synth_state_dict(args.nclass, seed=0))
    n, w, T = args.lines, args.width, args.width // 4 + 1
    rs = np.random.RandomState(77)
    inputs = [torch.from_numpy(rs.randint(0, 256, size=(n, 32, w)).astype(np.uint8)).to(dev) for _ in range(args.slots)]
    # --ragged: every slot has its own per-line valid widths (multiples of 4, 8 .. w); rows past a line's length are compared
    # too (the library leaves them as the synchronous call does)
    wid = [(rs.randint(2, w // 4 + 1, size=n) * 4).astype(np.int32) if args.ragged else None for _ in range(args.slots)]
    want = []
    for d, wd in zip(inputs, wid):   # synchronous calls: the reference rows (one-tile recurrence, one batch on the device)
        raw = torch.zeros((n, T), dtype=torch.int32, device=dev)
        lab = torch.zeros((n, T), dtype=torch.int32, device=dev)
        ln = torch.zeros((n,), dtype=torch.int32, device=dev)
        m.forward_device(d.data_ptr(), n, w, lab.data_ptr(), ln.data_ptr(), d_raw_ptr=raw.data_ptr(), widths=wd, sync=True)
        m.synchronize()
        want.append((raw.clone(), lab.clone(), ln.clone()))
    want_host = [x[0].cpu().numpy() for x in want]
    NS = args.lanes + 2
    outs = [(torch.zeros((n, T), dtype=torch.int32, device=dev), torch.zeros((n, T), dtype=torch.int32, device=dev),
             torch.zeros((n,), dtype=torch.int32, device=dev)) for _ in range(NS)]
    ext = [torch.cuda.ExternalStream(m.stream_ptr(l), device=dev) for l in range(args.lanes)]
    evs = [torch.cuda.Event() for _ in range(NS)]
    bad = []
    nbad = 0
    t0 = time.time()

    def launch(k):
        s = k % NS
        lane = m.forward_device(inputs[k % args.slots].data_ptr(), n, w, outs[s][1].data_ptr(), outs[s][2].data_ptr(),
                                d_raw_ptr=outs[s][0].data_ptr(), widths=wid[k % args.slots])
        evs[s].record(ext[lane])

    def check(k):
        nonlocal nbad
        s = k % NS
        evs[s].synchronize()
        wr, wl, wn = want[k % args.slots]
        if args.host_compare and not args.ragged:
            got = outs[s][0].cpu().numpy()          # (a copy engine transfer: no kernel)
            wh = want_host[k % args.slots]
            ok = bool(np.array_equal(got, wh))
            diff = torch.from_numpy(got != wh) if not ok else None
        elif args.ragged:      # compare what is defined for a ragged batch: the collapsed labels up to each line's length
            ok = bool((outs[s][2] == wn).all())
            if ok:
                msk = torch.arange(T, device=dev)[None, :] < wn[:, None]
                ok = bool(((outs[s][1] == wl) | ~msk).all())
            diff = (outs[s][1] != wl) if not ok else None
        else:
            diff = outs[s][0] != wr
            ok = not bool(diff.any())
        if not ok:
            pos = diff.nonzero().cpu().numpy().tolist()
            nbad += 1
            if len(bad) < 40:
                bad.append({"batch": k, "slot": k % args.slots, "positions": pos[:12], "count": len(pos)})

    for k in range(args.batches):
        launch(k)
        if k >= args.lanes:
            check(k - args.lanes)
        if k and k % 20000 == 0:      # (a long run keeps talking: a silent command is taken to be hung)
            print("soak_pipe: %d batches, %d bad, %.0f s" % (k, nbad, time.time() - t0), file=sys.stderr, flush=True)
    for k in range(max(0, args.batches - args.lanes), args.batches):
        check(k)
    print(json.dumps({"batches": args.batches, "lines": n, "width": w, "lanes": args.lanes, "precision": args.precision, "bad_batches": nbad,
                      "seconds": round(time.time() - t0, 1), "split_timeouts": m.split_timeouts(),
                      "split_nt": os.environ.get("CRNN_SPLIT_NT", "auto"), "first": bad}))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
