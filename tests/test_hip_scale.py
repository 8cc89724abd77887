"""Full-size behaviour of the HIP path through the C ABI (BASELINE.json configs 2-4 sizes), checked
through size-independent properties, plus the serving conventions (threads, fork).  Needs an MI355X."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from ocr_amd.crnn import keys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zh3(synth_zh):
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, lanes=3)
    m.load_state_dict(synth_zh)
    return m


def _device_batch(model, lines, widths=None):
    """forward_device on a uint8 [n,32,w] numpy batch -> (raw [n,T], lab [n,T], len [n]) on the host."""
    import torch
    dev = torch.device("cuda", 0)
    n, _, w = lines.shape
    T = w // 4 + 1
    d = torch.from_numpy(lines).to(dev)
    raw = torch.zeros((n, T), dtype=torch.int32, device=dev)
    lab = torch.zeros((n, T), dtype=torch.int32, device=dev)
    ln = torch.zeros((n,), dtype=torch.int32, device=dev)
    model.forward_device(d.data_ptr(), n, w, lab.data_ptr(), ln.data_ptr(), d_raw_ptr=raw.data_ptr(), widths=widths)
    model.synchronize()
    return raw.cpu().numpy(), lab.cpu().numpy(), ln.cpu().numpy()


def test_batch256_matches_golden_lines_anywhere_in_the_batch(golden, zh3):
    """Config 2 size: 256 lines of 32x280.  The 12 golden lines are scattered through the batch; every
    copy must decode to the reference's arg-max sequence wherever it sits (lines are independent), on
    every lane, and repeated calls are bit-identical."""
    G, M = golden
    gold = np.concatenate([G["syn280/noise/gray"], G["syn280/text/gray"]])
    want = np.concatenate([G["syn280/noise/zh_idx"], G["syn280/text/zh_idx"]])
    rs = np.random.RandomState(5)
    lines = rs.randint(0, 256, size=(256, 32, 280)).astype(np.uint8)
    pos = rs.permutation(256)[:48]
    for j, p in enumerate(pos):
        lines[p] = gold[j % 12]
    first = None
    for _ in range(4):                     # rotates over the three lanes and back to lane 0
        raw, lab, ln = _device_batch(zh3, lines)
        for j, p in enumerate(pos):
            assert np.array_equal(raw[p], want[j % 12])
        if first is None:
            first = (raw.copy(), lab.copy(), ln.copy())
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first, (raw, lab, ln)))
    # CTC collapse property on the device output: labels = collapse(raw) for every line
    from ocr_amd.crnn.util import strLabelConverter
    texts = zh3.texts_from_labels(lab, ln)
    for i in (0, 17, 100, 255):
        assert texts[i] == strLabelConverter(raw[i], keys.alphabetChinese)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pipelined_batches_equal_a_serial_run(synth_zh, precision):
    """bench.py's mode: batches queued back to back on four lanes, so kernels of different batches share CUs.  Every
    batch must come out bit-identical to a batch run alone (kernels must not disturb each other: the packed-fp32 / MFMA
    co-run problem of crnn_kernels_nopk.hip showed as a few wrong pixels in one batch of many)."""
    import torch
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, lanes=4, precision=precision)
    m.load_state_dict(synth_zh)
    dev = torch.device("cuda", 0)
    n, w, T, NB = 256, 280, 71, 48
    rs = np.random.RandomState(11)
    inputs = [torch.from_numpy(rs.randint(0, 256, size=(n, 32, w)).astype(np.uint8)).to(dev) for _ in range(3)]
    ref = []
    for d in inputs:     # alone, synchronous
        raw = torch.zeros((n, T), dtype=torch.int32, device=dev)
        lab = torch.zeros((n, T), dtype=torch.int32, device=dev)
        ln = torch.zeros((n,), dtype=torch.int32, device=dev)
        m.forward_device(d.data_ptr(), n, w, lab.data_ptr(), ln.data_ptr(), d_raw_ptr=raw.data_ptr(), sync=True)
        ref.append((raw.cpu().numpy(), lab.cpu().numpy(), ln.cpu().numpy()))
    outs = [(torch.zeros((n, T), dtype=torch.int32, device=dev), torch.zeros((n, T), dtype=torch.int32, device=dev),
             torch.zeros((n,), dtype=torch.int32, device=dev)) for _ in range(NB)]
    torch.cuda.synchronize()
    for k, (raw, lab, ln) in enumerate(outs):     # queued without waiting
        m.forward_device(inputs[k % 3].data_ptr(), n, w, lab.data_ptr(), ln.data_ptr(), d_raw_ptr=raw.data_ptr())
    m.synchronize()
    for k, (raw, lab, ln) in enumerate(outs):
        r = ref[k % 3]
        bad = int((raw.cpu().numpy() != r[0]).sum())
        assert bad == 0, "batch %d: %d arg-max indices differ from the serial run" % (k, bad)
        assert np.array_equal(lab.cpu().numpy(), r[1]) and np.array_equal(ln.cpu().numpy(), r[2])


def test_masked_batch_equals_exact_width_batches(zh3):
    """Config 3 shape: lines of different widths in ONE padded device batch (per-line masks) decode
    exactly like the same lines run in batches of their own width."""
    rs = np.random.RandomState(11)
    widths = [1024, 1000, 640, 611, 333, 280, 96, 64, 9, 8] * 3
    wp = max(widths)
    lines = np.zeros((len(widths), 32, wp), np.uint8)
    for i, w in enumerate(widths):
        lines[i, :, :w] = rs.randint(0, 256, size=(32, w))
        lines[i, :, w:] = rs.randint(0, 256, size=(32, wp - w))     # garbage right of the line must not matter
    raw, lab, ln = _device_batch(zh3, lines, widths=widths)
    for w in sorted(set(widths)):
        sel = [i for i, ww in enumerate(widths) if ww == w]
        r2, l2, n2 = _device_batch(zh3, np.ascontiguousarray(lines[sel][:, :, :w]))
        T = w // 4 + 1
        for k, i in enumerate(sel):
            assert np.array_equal(raw[i, :T], r2[k]), (w, i)
            assert ln[i] == n2[k] and np.array_equal(lab[i, :ln[i]], l2[k, :n2[k]])


def test_two_threads_share_one_handle(golden, zh3):
    """gunicorn runs 2 threads per worker on one model object (gunicorn.conf.py:24)."""
    G, M = golden
    crops = [G["inv/%s/bgr" % it["name"]] for it in M["invoice"]]
    want = [it["zh"] for it in M["invoice"]]
    errs = []

    def work(k):
        try:
            # 400 rounds (6800 calls): a wrong string in 1 of ~1500 calls is what the packed-fp32 / MFMA co-run problem looked like
            # from here (crnn_kernels_nopk.hip; tools/stress_threads.py is the long form of this test)
            for r in range(400):
                for c, w in list(zip(crops, want))[k::2]:
                    got = zh3.ocr(c)
                    assert got == w, "thread %d round %d: %r != %r" % (k, r, got, w)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_error_codes_through_the_abi(zh3):
    from ocr_amd.crnn import _capi
    lib = _capi.lib()
    h = zh3._h
    a = np.zeros((32, 40), np.uint8)
    il = _capi.ImageList([a])
    lab = np.zeros((1, 4), np.int32)
    ln = np.zeros(1, np.int32)
    # tmax too small for T = 11
    rc = lib.crnn_predict_batch(h, il.ptrs, il.h, il.w, il.stride, il.ch, 1, 0, 4, None, lab.ctypes.data,
                                ln.ctypes.data, None, None)
    assert rc == _capi.CRNN_ERR_ARG and b"tmax" in lib.crnn_last_error(h)
    # 2 channels
    il.ch[0] = 2
    rc = lib.crnn_predict_batch(h, il.ptrs, il.h, il.w, il.stride, il.ch, 1, 0, 16, None, lab.ctypes.data,
                                ln.ctypes.data, None, None)
    assert rc == _capi.CRNN_ERR_ARG
    # device entry point: width below 8, null pointers
    assert lib.crnn_forward_device(h, None, 1, 280, None, None, None, None, 1) == _capi.CRNN_ERR_ARG
    assert lib.crnn_forward_device(h, 1 << 20, 1, 4, None, None, 1 << 20, 1 << 20, 1) == _capi.CRNN_ERR_ARG
    # weights not loaded
    from ocr_amd.crnn.network_hip import CRNN
    empty = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
    with pytest.raises(_capi.CrnnError):
        empty.ocr(np.zeros((32, 100, 3), np.uint8))
    # preprocess works without weights (it needs none)
    assert empty.preprocess([np.zeros((32, 100), np.uint8)])[0].shape == (32, 100)


def _child(q, sd_small):
    try:
        import numpy as np
        from ocr_amd.crnn import keys
        from ocr_amd.crnn.network_hip import CRNN
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        q.put(repr(e))


def test_model_built_before_fork_is_usable_in_the_child(synth_en):
    """gunicorn preload_app (gunicorn.conf.py:57): the model object is built and its weights are
    loaded in the master BEFORE any HIP call; the forked worker initialises the device on first use.
    (The parent must not have touched the GPU: run in a fresh interpreter.)"""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from ocr_amd.crnn import keys
from ocr_amd.crnn.network_hip import CRNN
from ocr_amd.crnn.synth import synth_state_dict
m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish)
m.load_state_dict(synth_state_dict(65, seed=0))          # host only: no HIP call yet
rs = np.random.RandomState(0)
img = rs.randint(0, 256, size=(32, 200, 3)).astype(np.uint8)
r, w = os.pipe()
pid = os.fork()
if pid == 0:
    os.close(r)
    try:
        os.write(w, ("C:" + m.ocr(img)).encode())
    except Exception as e:
        os.write(w, ("E:" + repr(e)).encode())
    os._exit(0)
os.close(w)
child = os.read(r, 4096).decode()
os.waitpid(pid, 0)
parent = m.ocr(img)                                       # the parent initialises its own context afterwards
print(child)
print("P:" + parent)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and len(lines) == 2, (out.stdout, out.stderr[-2000:])
    assert lines[0].startswith("C:") and lines[1].startswith("P:"), lines
    assert lines[0][2:] == lines[1][2:] and len(lines[0]) > 2


@pytest.mark.gpu
def test_first_call_on_a_fresh_thread_and_handle(synth_en):
    """Entry points that do not run the network (stream query, synchronize, timing) must work as the very
    first call on a handle and from a thread that has never used it (the current lane is per thread)."""
    import threading
    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=3)
    m.load_state_dict(synth_en)
    assert m.stream_ptr(2)
    m.synchronize()
    res = {}

    def other():
        res["stream"] = m.stream_ptr(1)
        res["text"] = m.predict_lines([np.full((32, 100), 200, np.uint8)])
        m.synchronize()
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert res["stream"] and res["text"] == m.predict_lines([np.full((32, 100), 200, np.uint8)])


@pytest.mark.gpu
def test_pages_from_two_threads_use_their_own_lanes(golden, zh3):
    """crnn_predict_page: concurrent callers with DIFFERENT pages (two request threads of a worker) each get a
    lane with its own page buffer; results equal the host-cut crops."""
    rs = np.random.RandomState(21)
    pages = [rs.randint(0, 256, size=(400, 700, 3)).astype(np.uint8) for _ in range(2)]
    rects = [(10, 20, 310, 60), (5, 100, 650, 140), (300, 200, 460, 236), (0, 300, 700, 399), (600, 10, 640, 50)]
    want = [[zh3.ocr(pg[y1:y2, x1:x2]) for x1, y1, x2, y2 in rects] for pg in pages]
    got = [None, None]
    errs = []

    def work(k):
        try:
            for _ in range(4):
                got[k] = zh3.ocr_page(pages[k], rects)
                assert got[k] == want[k]
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("rec_res,split_nt", [("1", "1"), ("0", "1"), ("1", "2")])
def test_hidden_split_recurrence_is_bit_identical(golden, synth_zh, rec_res, split_nt):
    """Small batches use the four-workgroups-per-group recurrence (lstm_rec_split_h3_kernel) in the synchronous host paths;
    it must reproduce the one-workgroup kernel of the same arithmetic (lstm_rec_h3_kernel) bit for bit: same logits for
    n = 1, 17, 48 and 200 lines of different widths (lines ending at different steps exercise the keep-old-h path of the
    exchange).  rec_res: the one-workgroup kernel with a quarter of W_hh resident ("1", the default) and streaming all of
    it ("0"; CRNN_REC_RES is read at every launch).  split_nt: 16 or 32 lines per group of four workgroups (CRNN_SPLIT_NT, read
    at every launch; "2" is what pipelined device batches from 128 lines take -- here forced on every batch size, so that
    groups with one line, with 15 empty rows in the second tile and with ragged lengths all go through it)."""
    from ocr_amd.crnn.network_hip import CRNN
    G, M = golden
    old = os.environ.get("CRNN_LSTM_SPLIT")
    old_res = os.environ.get("CRNN_REC_RES")
    old_nt = os.environ.get("CRNN_SPLIT_NT")
    os.environ["CRNN_REC_RES"] = rec_res
    os.environ["CRNN_SPLIT_NT"] = split_nt
    try:
        os.environ["CRNN_LSTM_SPLIT"] = "0"
        ref = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=1)
        ref.load_state_dict(synth_zh)
        os.environ["CRNN_LSTM_SPLIT"] = "512"
        spl = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0, lanes=1)
        spl.load_state_dict(synth_zh)
    finally:
        if old is None:
            os.environ.pop("CRNN_LSTM_SPLIT", None)
        else:
            os.environ["CRNN_LSTM_SPLIT"] = old
    rs = np.random.RandomState(5)
    inv = [G[k] for k in sorted(G.files) if k.startswith("inv/") and k.endswith("/bgr")]
    pool = inv + [rs.randint(0, 256, size=(32, int(w))).astype(np.uint8) for w in rs.randint(8, 700, size=40)]
    for n in (1, 17, 48, 200):
        lines = [pool[i % len(pool)] for i in range(n)]
        a = ref.forward_logits(lines)
        b = spl.forward_logits(lines)
        assert np.array_equal(a[1], b[1])                 # T per line
        assert np.array_equal(a[0], b[0])                 # raw arg-max
        for i in range(n):
            assert np.array_equal(a[2][i, :a[1][i]], b[2][i, :b[1][i]]), (n, i)
        assert ref.predict_lines(lines) == spl.predict_lines(lines)
    # the largest batch that takes the split kernel: 512 narrow lines (256 workgroups, one per CU), ragged widths
    lines = [rs.randint(0, 256, size=(32, int(w))).astype(np.uint8) for w in rs.randint(40, 90, size=512)]
    a = ref._run(lines, 0, want_raw=True)     # 0 = PAD_PER_LINE: every line decoded as if alone
    b = spl._run(lines, 0, want_raw=True)
    if old_res is None:
        os.environ.pop("CRNN_REC_RES", None)
    else:
        os.environ["CRNN_REC_RES"] = old_res
    if old_nt is None:
        os.environ.pop("CRNN_SPLIT_NT", None)
    else:
        os.environ["CRNN_SPLIT_NT"] = old_nt
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
    assert spl.split_timeouts() == 0


@pytest.mark.gpu
def test_split_recurrences_of_concurrent_callers(golden, synth_en):
    """Three threads, each with batches that use the hidden-split recurrence, on one handle with three lanes:
    the split kernels are chained (one on the device at a time), results equal the single-threaded ones."""
    from ocr_amd.crnn.network_hip import CRNN
    m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=3)
    m.load_state_dict(synth_en)
    rs = np.random.RandomState(9)
    jobs = [[rs.randint(0, 256, size=(32, int(w))).astype(np.uint8) for w in rs.randint(40, 400, size=n)]
            for n in (5, 40, 90)]
    want = [m.predict_lines(j, min_cols=1 << 30) for j in jobs]
    got = [None] * 3
    errs = []

    def work(k):
        try:
            for _ in range(6):
                got[k] = m.predict_lines(jobs[k], min_cols=1 << 30)
                assert got[k] == want[k]
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert m.split_timeouts() == 0        # no batch had to be repeated with the one-workgroup kernel


@pytest.mark.gpu
def test_split_recurrences_of_two_handles_on_one_gpu(synth_en):
    """Two handles on the one GPU (two model objects in one process), a thread each, both using the hidden-split recurrence
    -- host batches and pipelined device batches: the split launches are chained per DEVICE, not per handle, so they
    cannot end up half resident and spinning for each other.  Results must equal the single-threaded ones; the number of
    repeats with the one-workgroup kernel (crnn_split_timeouts, device-side repeats included) is reported and must be 0."""
    import torch
    from ocr_amd.crnn.network_hip import CRNN
    ms = []
    for _ in range(2):
        m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=2)
        m.load_state_dict(synth_en)
        ms.append(m)
    rs = np.random.RandomState(13)
    jobs = [[rs.randint(0, 256, size=(32, int(w))).astype(np.uint8) for w in rs.randint(40, 400, size=n)] for n in (30, 70)]
    # (160 lines: pipelined device batches from 128 lines take the two-tile form, lstm_rec_split2_h3_kernel, the host batches
    # of 30 and 70 lines the one-tile form -- both forms of both handles share the device's chain)
    batches = [rs.randint(0, 256, size=(160, 32, 200)).astype(np.uint8) for _ in range(2)]
    want_host = [ms[k].predict_lines(jobs[k], min_cols=1 << 30) for k in range(2)]
    want_dev = [_device_batch(ms[k], batches[k]) for k in range(2)]
    errs = []

    def work(k):
        try:
            dev = torch.device("cuda", 0)
            d = torch.from_numpy(batches[k]).to(dev)
            n, _, w = batches[k].shape
            T = w // 4 + 1
            outs = [(torch.zeros((n, T), dtype=torch.int32, device=dev), torch.zeros((n, T), dtype=torch.int32, device=dev),
                     torch.zeros((n,), dtype=torch.int32, device=dev)) for _ in range(4)]
            for rep in range(5):
                assert ms[k].predict_lines(jobs[k], min_cols=1 << 30) == want_host[k]
                for o in outs:                       # four batches queued back to back on the two lanes (no sync)
                    ms[k].forward_device(d.data_ptr(), n, w, o[1].data_ptr(), o[2].data_ptr(), d_raw_ptr=o[0].data_ptr())
                ms[k].synchronize()
                for o in outs:
                    assert all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(want_dev[k], o)), rep
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    tmo = [m.split_timeouts() for m in ms]
    print("split time-outs of the two handles:", tmo)
    assert tmo == [0, 0]


@pytest.mark.gpu
def test_sharded_recognition_with_the_real_recogniser(synth_en, oracle_en):
    """ocr_amd/crnn/parallel.py with CRNN.predict_lines as the recogniser: rank 0 and rank 1 of a world of two, each
    with its own handle on the one GPU of this box, recognise their shards; the merged result is the oracle's strings
    in input order (the gather is replaced by a list of the two ranks' parts: no process group is needed for that)."""
    from oracle import oracle as orc
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.parallel import recognize_sharded
    rs = np.random.RandomState(31)
    widths = [int(w) for w in rs.choice(np.arange(24, 420, 4), size=60)] + [5, 8, 300, 300]
    lines = [np.where(rs.rand(32, w) < 0.8, 200 + rs.randint(0, 50, (32, w)), rs.randint(0, 90, (32, w))).astype(np.uint8)
             for w in widths]
    want = [orc.predict_gray(oracle_en, ln, keys.alphabetEnglish) if ln.shape[1] >= 8 else "" for ln in lines]
    models = []
    for _ in range(2):
        m = CRNN(32, 1, 65, 256, alphabet=keys.alphabetEnglish, device=0, lanes=2)
        m.load_state_dict(synth_en)
        models.append(m)
    parts = [None, None]

    def capture(rank):
        def gather(part):
            parts[rank] = part
            return [part]                      # this rank's own view; the merge below uses both
        return gather
    for rank in range(2):
        try:
            recognize_sharded(models[rank].predict_lines, lines, widths, rank, 2, gather=capture(rank), max_batch=16)
        except AssertionError:
            pass                               # a single rank's part does not cover every line: expected here
    assert parts[0] and parts[1] and not (set(parts[0]) & set(parts[1]))
    assert sorted(set(parts[0]) | set(parts[1])) == list(range(len(lines)))
    assert min(len(parts[0]), len(parts[1])) >= len(lines) // 4        # both ranks got real work
    merged = recognize_sharded(models[0].predict_lines, lines, widths, 0, 2, gather=lambda part: [parts[0], parts[1]],
                               max_batch=16)
    assert merged == want


@pytest.mark.gpu
def test_classifier_prefilter_certificate(golden, synth_zh):
    """The reduced-precision pre-filter of the classifier's arg-max (cls_select_kernel) rests on a rigorous bound of the
    approximate logits' error -- fp16 operands (the LSTM head's default): 0.00102 |a| |w_c| + 0.0005 |l~| + 7e-7 (|a| + |w_c|)
    + 1e-7; bf16 operands (CRNN_CLS_F16=0, dense head): 0.00785 |a| |w_c| + 0.004 |l~| -- checked here for both on the GPU's
    own classifier input (stage 9) in float64, together with its consequence -- the true arg-max is always among the
    candidates -- and the product path's indices (filter on) against the exact ones."""
    from oracle import oracle as O
    from ocr_amd.crnn.network_hip import CRNN
    G, _ = golden

    def to_bf16(x):
        u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16          # round to nearest even
        return u.astype(np.uint32).view(np.float32)

    m = CRNN(32, 1, 5530, 256, alphabet=keys.alphabetChinese, device=0)
    m.load_state_dict(synth_zh)
    W = np.asarray(synth_zh["rnn.1.embedding.weight"], np.float32)
    b = np.asarray(synth_zh["rnn.1.embedding.bias"], np.float32)
    wn = np.sqrt((W.astype(np.float64) ** 2).sum(1))
    for tag in ("noise", "text"):
        lines = G["syn280/%s/gray" % tag]
        x = np.stack([O.resize_normalize(g) for g in lines])
        a = m.debug_stage(x, 9).reshape(-1, 512)
        an = np.sqrt((a.astype(np.float64) ** 2).sum(1))
        exact = a.astype(np.float64) @ W.astype(np.float64).T + b
        ncand = {}
        for kind, rnd, k1, k2, kabs in (("f16", lambda x: np.asarray(x, np.float32).astype(np.float16).astype(np.float32), 0.00102, 0.0005, 7e-7),
                                        ("bf16", to_bf16, 0.00785, 0.004, 0.0)):
            approx = rnd(a).astype(np.float64) @ rnd(W).astype(np.float64).T + b
            approx = rnd(approx.astype(np.float32)).astype(np.float64)      # the kernel stores them in the same format
            eps = k1 * an[:, None] * wn[None, :] + k2 * np.abs(approx) + kabs * (an[:, None] + wn[None, :]) + (1e-7 if kabs else 0.0)
            assert (np.abs(approx - exact) <= eps).all(), kind
            lb = (approx - eps).max(1)
            cand = approx + eps >= lb[:, None]
            assert cand[np.arange(len(a)), exact.argmax(1)].all(), kind
            assert cand.sum(1).max() < 64          # the kernel's fast path covers these rows
            ncand[kind] = cand.sum(1).mean()
        assert ncand["f16"] < ncand["bf16"]
        # product path (filter on) == exact arg-max wherever the top two are not within fp32 rounding of each other
        from ocr_amd.crnn import _capi
        raw = m._run(list(lines), _capi.PAD_PER_LINE, want_raw=True)[3]     # logits not requested: filter path
        top2 = np.sort(exact, 1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4
        assert clear.mean() > 0.9
        assert (raw.reshape(-1)[clear] == exact.argmax(1)[clear]).all()
