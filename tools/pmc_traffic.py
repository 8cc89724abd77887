"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md section HBM prescribes: on gfx950 FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (double it); WRITE_SIZE is exact for 16-byte streaming stores.  Both
counters are in KiB.  Launches of one step are identified by their order inside the step."""
import csv, glob, sys, collections, json

ORDER_DIRECT = ["normalize", "conv0_pool0", "conv1", "pool1", "conv2", "conv3", "pool2", "conv4", "conv5", "pool3",
                "conv6", "lstm0_ih", "lstm0_rec", "emb0", "lstm1_ih", "lstm1_rec", "classifier", "argmax", "ctc"]
# fp32 default: conv3..5 as Winograd (input transform, batched GEMM, output transform)
ORDER_WINO = ["normalize", "conv0_pool0", "conv1", "pool1", "conv2", "conv3_win", "conv3", "conv3_wout", "pool2",
              "conv4_win", "conv4", "conv4_wout", "conv5_win", "conv5", "conv5_wout", "pool3",
              "conv6", "lstm0_ih", "lstm0_rec", "emb0", "lstm1_ih", "lstm1_rec", "classifier", "argmax", "ctc"]
# fused: the GEMM applies the output transform itself (no *_wout launches)
ORDER_FUSED = ["normalize", "conv0_pool0", "conv1_win", "conv1_pool1", "conv2_win", "conv2", "conv3_win", "conv3", "pool2",
               "conv4_win", "conv4", "conv5_win", "conv5", "pool3",
               "conv6", "lstm0_ih", "lstm0_rec", "emb0", "lstm1_ih", "lstm1_rec", "classifier", "argmax", "ctc"]
# hybrid: every Winograd layer is two launches at batch 256 (128x128 tiles over the full rounds, then a
# small-tile launch over the remaining tiles, "<layer>_tail")
ORDER_HYBRID = []
for _k in ORDER_FUSED:
    ORDER_HYBRID.append(_k)
    if _k in ("conv1_pool1", "conv2", "conv3", "conv4", "conv5"):
        ORDER_HYBRID.append(_k + "_tail")
# f4 (fp32 default): conv1-5 as F(4x4): transform, GEMM, transform (pool1 / pool2 / pool3 inside the output
# transforms of conv1 / conv3 / conv5)
ORDER_F4 = ["normalize", "conv0_pool0"]
for _l in (1, 2, 3, 4, 5):
    ORDER_F4 += ["conv%d_win4" % _l, "conv%d_f4" % _l,
                 "conv%d_wout4" % _l + {3: "_pool2", 5: "_pool3"}.get(_l, "")]
ORDER_F4 += ["conv6_win4", "conv6_f4", "conv6_wout4",   # conv6 as 1-D F(4,2)
              "lstm0_ih", "lstm0_rec", "emb0", "lstm1_ih", "lstm1_rec",
              "cls_to_bf16", "cls_filter_bf16", "cls_select", "ctc"]   # classifier: certified bf16 pre-filter
MODE = sys.argv[4] if len(sys.argv) > 4 else "direct"
ORDER = {"wino": ORDER_WINO, "fused": ORDER_FUSED, "hybrid": ORDER_HYBRID, "f4": ORDER_F4}.get(MODE, ORDER_DIRECT)
if MODE.endswith(".json"):     # the launch order of a step as a bench.py line of the same build recorded it ("kernels")
    ORDER = list(json.loads(open(MODE).read().strip().splitlines()[-1])["kernels"].keys())


def per_launch(d, counter):
    f = (glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv'))[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and 'crnn::' in r['Kernel_Name']
            and 'crnn::warm_' not in r['Kernel_Name']]      # (set-up launches, not part of a step)
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    vals = [float(r['Counter_Value']) for r in rows]
    names = [r['Kernel_Name'].split('(')[0].replace('crnn::', '') for r in rows]
    n = len(ORDER)
    # a step starts at its normalize launch; only steps with exactly the launches of ORDER count (the asynchronous form
    # of a step queues one more launch behind each hidden-split recurrence, the gated repeat: those steps are left out)
    starts = [i for i, nm in enumerate(names) if nm.startswith('normalize_kernel')] + [len(names)]
    steps = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a == n]
    assert len(steps) >= 2, "no two steps of %d launches in %s" % (n, f)
    out = collections.OrderedDict()
    for i, nm in enumerate(ORDER):
        xs = [vals[a + i] for a, _ in steps[1:]]       # skip the first (cold) step
        out[nm] = (sum(xs) / len(xs), names[steps[1][0] + i])
    return out


fetch = per_launch(sys.argv[1], 'FETCH_SIZE')
write = per_launch(sys.argv[2], 'WRITE_SIZE')
res = {}
for nm in ORDER:
    rd = fetch[nm][0] * 1024 * 2     # KiB -> bytes, x2 gfx950 correction for wide coalesced reads
    wr = write[nm][0] * 1024
    res[nm] = {"kernel": fetch[nm][1], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
    print("%-12s %-22s read %9.2f MB  write %9.2f MB" % (nm, fetch[nm][1], rd / 1e6, wr / 1e6))
json.dump(res, open(sys.argv[3], 'w'), indent=1)
