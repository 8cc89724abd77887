#!/usr/bin/env python3
"""Throughput benchmark of the CRNN text-line path on MI355X.

Workload (BASELINE.json configs[1]): CRNN with the 5530-class (ocr-lstm) head, fp32,
one step = one batch of 256 synthetic 32x280 grayscale lines, already resident in HBM
as uint8, through normalise -> conv stack -> 2x BiLSTM -> classifier -> argmax -> greedy
CTC collapse, label indices copied back and turned into Python strings on the host.
Weights are the seeded synthetic ones (the trained checkpoints are not in the reference
snapshot); timing does not depend on weight values.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU, every rank runs its own replica on its own lines (weak scaling,
no data-path collective; gloo is used only for the barrier and the max over ranks).
Rank 0 prints ONE JSON line.

  python bench.py --lines 1000000 [--gpus N] [--precision fp32|bf16]
BASELINE.json configs[3]: a fixed job of N synthetic 32x280 lines, generated on each GPU from a seed
before the timed region (resident in HBM), cut into width buckets and dealt to the ranks by
ocr_amd/crnn/parallel.py (strong scaling: the job is fixed, the ranks share it); the timed region runs every
rank's buckets through the pipeline and gathers the decoded strings of all ranks on the host.

What the timed region of the default mode contains and what it does not: it starts from uint8 lines of height
32 already in HBM (no upload, no resize -- those are tools/bench_configs.py's crop-replay figures, the
like-for-like number beside the CPU baseline, which starts from host images) and ends with Python strings on
the host (labels copied back per batch, vectorised code-point lookup).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The lanes are HIP streams: the runtime needs enough hardware queues for them (GPU_MAX_HW_QUEUES, read when HIP
# initialises; its default of 4 makes four lanes slower than three).  A deployment exports the variable itself
# (INTEGRATION.md); importing the binding raises it to 8 when it is unset and HIP is not up yet (under a profiler that
# has initialised HIP already it records the 4 in effect instead), and the value in effect goes into the JSON line
# (config.gpu_max_hw_queues).  See ocr_amd/crnn/_capi.py _more_hw_queues.
from ocr_amd.crnn import _capi  # noqa: E402,F401  (before torch: must precede HIP initialisation)


BATCH, WIDTH, NCLASS = int(os.environ.get("CRNN_BENCH_BATCH", "256")), 280, 5530   # the env override is for experiments only (configs[1] is 256)
FLOP_PER_LINE = 4283.46e6          # 2*MACs per 32x280 line, nclass 5530
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak
PEAK_HBM_GBS = 8000.0
# direct-form FLOPs (2 * MACs) per 32x280 line of the conv layers
DIRECT_MFLOP = {"conv1": 330.30, "conv2": 330.30, "conv3": 660.60, "conv4": 670.04, "conv5": 1340.08, "conv6": 148.90}


def cpu_baseline(lines_u8, sd, alphabet, want_seconds=12.0):
    """The torch.nn restatement (oracle/torch_port.py) timed on the host cores: preprocess +
    forward + argmax + Python CTC decode on 256-line batches of the same workload."""
    import torch
    from oracle.torch_port import TorchPort
    m = TorchPort(NCLASS, alphabet=alphabet).load_numpy(sd)
    cores = torch.get_num_threads()
    m.predict_gray_batch(lines_u8[:8])  # warm-up
    done, t0 = 0, time.perf_counter()
    texts = None
    while True:
        out = m.predict_gray_batch(lines_u8[:BATCH])
        texts = texts or out
        done += BATCH
        el = time.perf_counter() - t0
        if el >= want_seconds or done >= 8 * BATCH:
            break
    return {"value": round(done / el, 2), "unit": "text-lines/s", "cores": cores, "kind": "port",
            "sample": "%d lines (batches of %d synthetic 32x280 lines, the GPU workload) through oracle/torch_port.py "
                      "(torch.nn CPU fp32, %d threads): PIL resize+normalise, forward, argmax, Python CTC decode; %.1f s"
                      % (done, BATCH, cores, el)}, texts


def char_agreement(got, want):
    """1 - (sum of edit distances) / (sum of reference lengths)."""
    err = tot = 0
    for a, b in zip(got, want):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        err += prev[-1]
        tot += len(b)
    return 1.0 - err / max(1, tot)


def lines_job(args, model, dev, rank, world, dist):
    """BASELINE configs[3]: a fixed job of args.lines synthetic 32x280 lines.  Every rank computes the same partition
    (ocr_amd/crnn/parallel.py: width buckets dealt to ranks by cost), generates ITS lines on its GPU from the seed
    (bucket j of the job always holds the same lines, whatever the number of ranks), and -- timed -- runs its buckets
    through the lane pipeline, turns the labels into strings and gathers the strings of all ranks on the host."""
    import torch
    from ocr_amd.crnn.parallel import my_buckets
    N = args.lines
    NL = model.lanes
    T = WIDTH // 4 + 1
    widths = [WIDTH] * N
    mine = my_buckets(widths, rank, world, max_batch=BATCH)          # list of index lists (<= 256 lines each)
    nmine = sum(len(b) for b in mine)
    # the rank's lines, resident in HBM before the timed region; bucket j's lines come from seed + first index of j
    d_lines = torch.empty((max(nmine, 1), 32, WIDTH), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    offs, o = [], 0
    for b in mine:
        gen.manual_seed(args.seed * 1000003 + b[0])
        d_lines[o:o + len(b)] = torch.randint(0, 256, (len(b), 32, WIDTH), dtype=torch.uint8, device=dev, generator=gen)
        offs.append(o)
        o += len(b)
    NS = NL + 1
    d_lab = [torch.zeros((BATCH, T), dtype=torch.int32, device=dev) for _ in range(NS)]
    d_len = [torch.zeros((BATCH,), dtype=torch.int32, device=dev) for _ in range(NS)]
    h_lab = [torch.zeros((BATCH, T), dtype=torch.int32).pin_memory() for _ in range(NS)]
    h_len = [torch.zeros((BATCH,), dtype=torch.int32).pin_memory() for _ in range(NS)]
    ext = [torch.cuda.ExternalStream(model.stream_ptr(l), device=dev) for l in range(NL)]
    evs = [torch.cuda.Event() for _ in range(NS)]
    torch.cuda.synchronize()

    def launch(k):
        s, n = k % NS, len(mine[k])
        lane = model.forward_device(d_lines[offs[k]].data_ptr(), n, WIDTH, d_lab[s].data_ptr(), d_len[s].data_ptr())
        with torch.cuda.stream(ext[lane]):
            h_lab[s].copy_(d_lab[s], non_blocking=True)
            h_len[s].copy_(d_len[s], non_blocking=True)
            evs[s].record(ext[lane])

    def collect(k, part):
        s, n = k % NS, len(mine[k])
        evs[s].synchronize()
        for i, t in zip(mine[k], model.texts_from_labels(h_lab[s].numpy()[:n], h_len[s].numpy()[:n])):
            part[i] = t

    def job(buckets):
        part = {}
        for k in buckets:
            launch(k)
            if k - buckets[0] >= NL:
                collect(k - NL, part)
        for k in buckets[max(0, len(buckets) - NL):]:
            collect(k, part)
        return part

    nb = len(mine)
    job(list(range(min(nb, max(args.warmup, NL)))))       # warm-up: workspace allocation, clocks
    model.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    part = job(list(range(nb)))
    model.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, part)                # the host gather of decoded strings (no device collective)
        dist.barrier()
    else:
        parts = [part]
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    total = sum(len(p) for p in parts)
    assert total == N, (total, N)
    if rank == 0:
        chars = sum(len(t) for p in parts for t in p.values())
        # digest of the gathered strings in line order: the same job must give the same digest on any number of ranks
        import hashlib
        allt = {}
        for p in parts:
            allt.update(p)
        assert sorted(allt) == list(range(N)), "gathered line indices are not 0..N-1 exactly once"
        digest = hashlib.sha256("\n".join(allt[i] for i in range(N)).encode("utf-8")).hexdigest()
        steps = max(1, (N + BATCH - 1) // BATCH)
        out = {"metric": "text-lines/sec at 32x280 crops (CRNN ocr-lstm head, 5530 classes)",
               "value": round(N / elapsed, 1), "unit": "text-lines/s", "n_gpus": world, "steps": steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
               "data": "synthetic (uniform uint8 lines generated on-device from a seed, seeded synthetic weights)",
               "config": {"workload": "configs[3]: %d synthetic 32x280 lines, generated on-device from seed %d, width-bucketed "
                                      "(256 lines) and sharded over %d GPU(s) by rank, CRNN %s; decoded strings gathered on "
                                      "the host (timed)" % (N, args.seed, world, args.precision),
                          "lines": N, "batch": BATCH, "width": WIDTH, "nclass": NCLASS, "lanes_per_gpu": NL,
                          "buckets_rank0": nb, "sharding": "independent lines, one replica per GPU, no collective",
                          "gpu_max_hw_queues": _hw_queues(), "split_timeouts_rank0": model.split_timeouts()},
               "elapsed_s": round(elapsed, 3), "decoded_chars": chars, "strings_sha256": digest,
               "lines_per_rank": [len(p) for p in parts]}
        print(json.dumps(out, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher: start N child processes of this script, one per GPU (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), wait for all of
    them and return the first non-zero exit code.  The children inherit stdout, so rank 0's JSON line is this
    process's output.  The parent never initialises HIP and never execs."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            c = p.wait()
            if c and not rc:
                rc = c
                for q in procs:          # one rank failed: the others would wait at a barrier for ever
                    if q.poll() is None:
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        rc = 130
    return rc


def _hw_queues():
    from ocr_amd.crnn import _capi
    return _capi.effective_hw_queues()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lanes", type=int, default=0,
                    help="independent stream/workspace lanes per GPU (2 since round 3: with the hidden-split recurrence in the "
                         "pipelined path a batch alone takes 2.9 ms instead of 3.9, and two in flight fill the chip; the bf16 "
                         "path keeps the one-workgroup recurrence and four lanes); 0 = that default")
    ap.add_argument("--lines", type=int, default=0,
                    help="BASELINE configs[3]: a fixed job of this many synthetic 32x280 lines (1000000 in the config), "
                         "generated on-device from --seed and sharded over the ranks; 0 = the default per-step mode")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="fp32 = the headline, parity-exact path; bf16 = bf16 MFMA inputs, fp32 accumulate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="one batch at a time (for rocprofv3 --kernel-trace --stats: per-kernel durations that are "
                         "not stretched by the other lanes, with the tile shapes the pipelined run uses)")
    args = ap.parse_args()
    if args.lanes <= 0:
        args.lanes = 2 if args.precision == "fp32" else 4

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called without a launcher: be the launcher.  Nothing in this process has touched HIP yet (torch is not even
        # imported), the ranks are fresh child processes, and rank 0's JSON line passes through on its stdout.
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "BENCH_FORCE_DEVICE" in os.environ:      # rehearsal of the N>1 path on a one-GPU box
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from ocr_amd.crnn import keys
    from ocr_amd.crnn.network_hip import CRNN
    from ocr_amd.crnn.synth import synth_state_dict

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sd = synth_state_dict(NCLASS, seed=0)
    model = CRNN(32, 1, NCLASS, 256, leakyRelu=False, lstmFlag=True, GPU=True, alphabet=keys.alphabetChinese,
                 device=local_rank, lanes=args.lanes, precision=args.precision)
    model.load_state_dict(sd)
    NL = model.lanes

    # synthetic lines, resident in HBM before the timed region; a different batch per step slot
    nslots = 4
    rs = np.random.RandomState(1000 + rank)
    host_lines = rs.randint(0, 256, size=(nslots, BATCH, 32, WIDTH)).astype(np.uint8)
    d_lines = torch.from_numpy(host_lines).to(dev)
    T = WIDTH // 4 + 1
    NS = NL + 1    # result slots: NL batches in flight + the one being read on the host

    def pipeline(mdl):
        d_lab = [torch.zeros((BATCH, T), dtype=torch.int32, device=dev) for _ in range(NS)]
        d_len = [torch.zeros((BATCH,), dtype=torch.int32, device=dev) for _ in range(NS)]
        h_lab = [torch.zeros((BATCH, T), dtype=torch.int32).pin_memory() for _ in range(NS)]
        h_len = [torch.zeros((BATCH,), dtype=torch.int32).pin_memory() for _ in range(NS)]
        ext = [torch.cuda.ExternalStream(mdl.stream_ptr(l), device=dev) for l in range(NL)]
        evs = [torch.cuda.Event() for _ in range(NS)]

        def launch(k):
            s = k % NS
            # --serial: one batch on the device at a time (sync), which also tells the library that its persistent
            # launches may take every CU (256 blocks instead of the pipelined 192)
            lane = mdl.forward_device(d_lines[k % nslots].data_ptr(), BATCH, WIDTH, d_lab[s].data_ptr(),
                                      d_len[s].data_ptr(), sync=args.serial)
            with torch.cuda.stream(ext[lane]):      # copy-back queued behind the batch on ITS lane
                h_lab[s].copy_(d_lab[s], non_blocking=True)
                h_len[s].copy_(d_len[s], non_blocking=True)
                evs[s].record(ext[lane])

        def collect(k):
            s = k % NS
            evs[s].synchronize()
            return mdl.texts_from_labels(h_lab[s].numpy(), h_len[s].numpy())

        def run(nsteps):
            """nsteps batches; up to NL in flight on the device while the host decodes finished ones."""
            texts = None
            for k in range(nsteps):
                launch(k)
                if args.serial:
                    mdl.synchronize()
                if k >= NL:
                    texts = collect(k - NL)
            for k in range(max(0, nsteps - NL), nsteps):
                texts = collect(k)
            return texts
        return run, d_lab, d_len

    if args.lines > 0:
        return lines_job(args, model, dev, rank, world, dist)

    run, d_lab, d_len = pipeline(model)

    def barrier():
        if world > 1:
            dist.barrier()

    first_texts = run(1)   # also allocates the workspace
    run(args.warmup)
    model.synchronize()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    model.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    split_tmo = model.split_timeouts()   # give-ups of the hidden-split recurrence so far (warm-up + timed region): 0 expected
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    # ---- per-kernel device times (HIP events on the library's stream), outside the timed region
    model.set_profiling(True)
    acc = {}
    order = []
    nprof = max(3, min(10, args.steps))
    for k in range(nprof):
        model.forward_device(d_lines[k % nslots].data_ptr(), BATCH, WIDTH, d_lab[0].data_ptr(), d_len[0].data_ptr())
        model.synchronize()
        for name, ms, fl, by in model.kernel_times():
            if name not in acc:
                acc[name] = [0.0, fl, by]
                order.append(name)
            acc[name][0] += ms / nprof
    # the preprocessing kernels of the host-image path (gray conversion + Pillow-exact resize + normalise): 256 BGR crops of
    # 64x560 in one bucket through predict_lines, per-kernel device times of that call (HBM stages of the north star)
    pre_stages = []
    if world == 1 and not args.serial:
        crops = [rs.randint(0, 256, size=(64, 560, 3)).astype(np.uint8) for _ in range(BATCH)]
        model.predict_lines(crops[:8], max_batch=BATCH)
        model.predict_lines(crops, max_batch=BATCH)
        for name, ms, fl, by in model.kernel_times():
            if name.startswith("resize"):
                pre_stages.append({"stage": name + " (256 BGR crops 64x560 -> 32x280)", "ms": round(ms, 4),
                                   "gbs": round(by / (ms * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})
    model.set_profiling(False)
    # like-for-like with the CPU baseline: the SAME 256 host lines the CPU baseline gets, from host memory to Python strings
    # through CRNN.predict_lines (upload, resize path, network, copy-back, decode), several batches kept in flight by threads
    host_path = None
    if world == 1 and not args.serial and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        hl = list(host_lines[0])
        model.predict_lines(hl, max_batch=BATCH)
        reps = 12
        with ThreadPoolExecutor(min(NL, 4)) as ex:
            th = time.perf_counter()
            res = list(ex.map(lambda _: model.predict_lines(hl, max_batch=BATCH), range(reps)))
            eh = time.perf_counter() - th
        host_path = {"value": round(BATCH * reps / eh, 1), "unit": "text-lines/s",
                     "what": "CRNN.predict_lines on the CPU baseline's own 256 host lines (host uint8 arrays in, strings out; "
                             "%d calls from %d threads)" % (reps, min(NL, 4)),
                     "strings_identical_to_headline_path": "%d/%d" % (sum(a == b for a, b in zip(first_texts, res[-1])), BATCH)}

    # ---- the same pipeline with parts of the arithmetic switched back to the f32 MFMA, for reference beside the headline
    # (rank 0 of a one-GPU run only; outside the timed region above): the classifier's arg-max from a plain fp32 GEMM (no
    # bf16 pre-filter), the Winograd / projection GEMMs on the f32 MFMA (no f16x3), and both
    plain = None
    if (args.precision == "fp32" and world == 1 and not args.serial and not args.no_cpu_baseline
            and os.environ.get("CRNN_CLS_FILTER", "1") != "0" and os.environ.get("CRNN_F16X3", "1") != "0"):
        # pinned result buffers remember the lane streams they were used on: free them (the closure) before the handle
        run = None
        gc.collect()
        model.__del__()          # frees the first handle's streams: both handles' lanes would share the hardware queues
        plain = {}
        for tag, env in (("conv3_6_gemms_loader_wave_form_of_round_2", {"CRNN_GEMM_PP": "0", "CRNN_LIN_PP": "0"}),   # (and the plain GEMMs on lin_gemm_h3_kernel)
                         ("classifier_plain_fp32_gemm", {"CRNN_CLS_FILTER": "0"}),
                         ("gemms_on_f32_mfma", {"CRNN_F16X3": "0"}),
                         ("all_on_f32_mfma", {"CRNN_CLS_FILTER": "0", "CRNN_F16X3": "0"})):
            os.environ.update(env)
            model2 = CRNN(32, 1, NCLASS, 256, leakyRelu=False, lstmFlag=True, GPU=True, alphabet=keys.alphabetChinese,
                          device=local_rank, lanes=args.lanes, precision=args.precision)
            model2.load_state_dict(sd)
            run2, _, _ = pipeline(model2)
            texts2 = run2(1)
            run2(args.warmup)
            model2.synchronize()
            t2 = time.perf_counter()
            run2(args.steps)
            model2.synchronize()
            e2 = time.perf_counter() - t2
            for k in env:             # (CRNN_GEMM_PP is read per launch, the others when the handle is created)
                del os.environ[k]
            plain[tag] = {"value": round(BATCH * args.steps / e2, 1), "unit": "text-lines/s",
                          "ms_per_step": round(e2 / args.steps * 1e3, 3),
                          "strings_identical_to_headline_path": "%d/%d" % (sum(a == b for a, b in zip(first_texts, texts2)), len(texts2))}
            run2 = None
            gc.collect()
            model2.__del__()
            del model2

    run = None          # same order at every exit: pinned buffers (held by the closure) go before the handle's streams
    gc.collect()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    fp32 = args.precision == "fp32"
    h3 = fp32 and os.environ.get("CRNN_F16X3", "1") != "0"         # Winograd / projection GEMMs on the fp16 MFMA (f16x3)
    lines_total = BATCH * args.steps * world
    value = lines_total / elapsed
    step_s = elapsed / args.steps
    # ---- GEMM kernel families.  acc[name] = [ms per launch (HIP events on the library's stream, batches run one at a
    # time), fp32 FLOPs of the launch as executed (2 M N K of the GEMMs the kernel really runs: Winograd component
    # GEMMs, not the direct form), algorithmic bytes].  An f16x3 kernel executes THREE fp16 MFMA products per fp32
    # product: its roofline is the fp16 MFMA's, with 3x those FLOPs.
    wino4_names = [n for n in order if n.endswith("_f4")]
    # the same GEMMs in the ping-pong form (round 3), and the plain GEMMs that run on that kernel (lstm0_ih_pp, ...)
    wino4pp_names = [n for n in order if n.endswith("_f4pp") or n.endswith("_pp")]
    fused4_names = [n for n in order if n.endswith("_fused4")]      # GEMMs + output transform in one kernel (f16x3 only)
    lin_on = fp32 and os.environ.get("CRNN_LIN_GEMM", "1") != "0"
    lin_names = [n for n in order if lin_on and n in ("lstm0_ih", "lstm1_ih", "emb0", "classifier")]
    other_names = [n for n in order if (n in ("lstm0_ih", "lstm1_ih", "emb0", "classifier") or
                                        (n.startswith("conv") and not n.startswith("conv0_pool0") and "_win" not in n and "_wout" not in n))
                   and n not in wino4_names + wino4pp_names + fused4_names + lin_names]

    def family(kernel, names, what, products, peak):
        ms = sum(acc[n][0] for n in names)
        fl = sum(acc[n][1] for n in names)
        eq = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0        # fp32 GEMM FLOPs per second
        direct = sum(DIRECT_MFLOP.get(n.split("_")[0], 0.0) for n in names) * 1e6 * BATCH
        f = {"kernel": "%s (%s; %d launches per step)" % (kernel, what, len(names)), "launches": names,
             "ms_per_step": round(ms, 4), "ms_per_launch_avg": round(ms / max(1, len(names)), 4),
             "achieved": round(eq * products, 2), "peak": peak, "frac": round(eq * products / peak, 4),
             "mfma_products_per_fp32_product": products,
             "fp32_equivalent_tflops": round(eq, 2),
             "fp32_equivalent_over_f32_mfma_peak": round(eq / PEAK_F32_MFMA_TFLOPS, 4),
             "flops_per_launch_avg": products * fl / max(1, len(names)),
             "algorithmic_bytes_per_launch_avg": sum(acc[n][2] for n in names) / max(1, len(names))}
        if direct:
            f["direct_form_tflops"] = round(direct / (ms * 1e-3) / 1e12, 2)
            f["direct_form_over_f32_mfma_peak"] = round(direct / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        return f
    fams = []
    if wino4pp_names:
        fams.append(family("wino4_gemm_h3_pp_kernel", wino4pp_names,
                           "the 36 Winograd F(4x4,3x3) component GEMMs of a layer (conv4, conv5: N = 512, K <= 512), and the plain "
                           "GEMMs of the LSTM input projections and the first embedding, each as one persistent "
                           "launch of 256 x 128 units on the fp16 MFMA: operands as (hi, lo) fp16 pairs, 3 MFMA products per fp32 "
                           "product, 64 x 64 wave tiles, two groups of four MFMA waves one slot apart (ping-pong), four loader waves",
                           3, PEAK_BF16_MFMA_TFLOPS))
    if wino4_names:
        if h3:
            fams.append(family("wino4_gemm_h3_lw_kernel", wino4_names,
                               "Winograd component GEMMs of a layer -- 36 for F(4x4,3x3) (conv3; the others when the ping-pong form is "
                               "off), 5 for conv6's 1-D F(4,2) -- as one persistent launch of 128x128 units on the fp16 "
                               "MFMA, operands as (hi, lo) fp16 pairs, 3 MFMA products per fp32 product, loader waves, product stores "
                               "through LDS (wino4_gemm_h3_kernel below K = 256)", 3, PEAK_BF16_MFMA_TFLOPS))
        else:
            fams.append(family("wino4_gemm_lw_kernel", wino4_names,
                               "Winograd component GEMMs of a layer as one persistent launch of 128x128 units on the f32 MFMA", 1,
                               PEAK_F32_MFMA_TFLOPS))
    # (one family per kernel SYMBOL: conv1 runs the POOL = 1 instantiation of the fused kernel, conv2 the POOL = 0 one)
    for inst, names in (("wino4_fused_h3_kernel<1>", [n for n in fused4_names if n.startswith("conv1")]),
                        ("wino4_fused_h3_kernel<0>", [n for n in fused4_names if not n.startswith("conv1")])):
        if names:
            fams.append(family(inst, names,
                               "the 36 Winograd component GEMMs of a layer AND its output transform (+ bias, activation%s) in one "
                               "persistent launch of 64x64 units, no product tensor; fp16 MFMA, f16x3 operands; bound by operand "
                               "delivery from L2 (its 64 x 64 unit needs 32 KB of operands per 384 MFMA cycles)"
                               % (", pool1" if inst.endswith("<1>") else ""), 3, PEAK_BF16_MFMA_TFLOPS))
    if lin_names:
        h3l = [n for n in lin_names if h3 and n != "classifier"]
        if h3l:
            fams.append(family("lin_gemm_h3_kernel", h3l, "plain GEMMs (" + ", ".join(h3l) + ") as persistent launches of 128x128 "
                               "units on the fp16 MFMA, f16x3 operands", 3, PEAK_BF16_MFMA_TFLOPS))
        f32l = [n for n in lin_names if n not in h3l]
        if f32l:
            fams.append(family("lin_gemm_kernel", f32l, "plain GEMMs on the f32 MFMA: " + ", ".join(f32l), 1, PEAK_F32_MFMA_TFLOPS))
    if other_names:
        fams.append(family("conv_gemm_kernel" if fp32 else "conv_gemm_bf16_kernel", other_names,
                           "implicit GEMM: " + ", ".join(other_names), 1, PEAK_F32_MFMA_TFLOPS if fp32 else PEAK_BF16_MFMA_TFLOPS))
    fams.sort(key=lambda f: -f["ms_per_step"])
    dom = fams[0]
    gemm_names = dom["launches"]
    kernels = {n: {"ms": round(acc[n][0], 4),
                   "tflops": round(acc[n][1] / (acc[n][0] * 1e-3) / 1e12, 2) if acc[n][1] else None,
                   "gbs_algorithmic": round(acc[n][2] / (acc[n][0] * 1e-3) / 1e9, 1)} for n in order}
    # HBM traffic of the same launches from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs, gfx950 x2 read correction; tools/pmc_traffic.py).  PMC counters cannot be collected from inside this process,
    # so the figure comes from profiles/ and the line says which file and which commit of it.
    traffic, traffic_src = None, None
    for rnd in ("r3", "r2", "r1"):
        tp = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, args.precision))
        if not os.path.exists(tp):
            continue
        with open(tp) as f:
            tj = json.load(f)
        if all(n in tj for n in gemm_names) and (not h3 or "wino4_gemm_h3" in tj[gemm_names[0]].get("kernel", "")):
            traffic = sum(tj[n]["hbm_bytes"] for n in gemm_names) / len(gemm_names)
            traffic_src = "profiles/" + os.path.basename(tp)
            try:
                import subprocess
                c = subprocess.run(["git", "log", "-1", "--format=%h", "--", tp], cwd=ROOT, capture_output=True, text=True, timeout=10)
                if c.stdout.strip():
                    traffic_src += " @ " + c.stdout.strip()
            except Exception:  # noqa: BLE001
                pass
            break
    # HBM-bound stages: achieved algorithmic GB/s and the fraction of the 8 TB/s peak (north star: preprocess, arg-max, decode)
    hbm_names = [n for n in order if (acc[n][1] == 0 and n != "cls_filter_bf16") or n.startswith("conv0_pool0")]
    hbm_stages = [{"stage": n, "ms": round(acc[n][0], 4), "gbs": round(acc[n][2] / (acc[n][0] * 1e-3) / 1e9, 1),
                   "frac_of_hbm_peak": round(acc[n][2] / (acc[n][0] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)} for n in hbm_names]
    # MFMA time at peak of every launch of a step / step time: how busy the matrix pipes are over the whole pipeline
    cls_flops = 2.0 * BATCH * (WIDTH // 4 + 1) * NCLASS * 512
    mfma_s = 0.0
    for n in order:
        if n == "cls_filter_bf16":
            mfma_s += cls_flops / (PEAK_BF16_MFMA_TFLOPS * 1e12)
        elif acc[n][1] and not n.startswith("conv0_pool0"):
            is_h3 = h3 and (n in wino4_names or n in wino4pp_names or n in fused4_names or n in ("lstm0_ih", "lstm1_ih", "emb0"))   # (wino4pp_names has the *_pp plain GEMMs)
            mfma_s += (3.0 * acc[n][1] / (PEAK_BF16_MFMA_TFLOPS * 1e12)) if is_h3 else acc[n][1] / ((PEAK_F32_MFMA_TFLOPS if fp32 else PEAK_BF16_MFMA_TFLOPS) * 1e12)
    serial_ms = sum(acc[n][0] for n in order)
    peak = dom["peak"]

    out = {
        "metric": "text-lines/sec at 32x280 crops (CRNN ocr-lstm head, 5530 classes)",
        "value": round(value, 1), "unit": "text-lines/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if fp32 else "bf16",
        "data": "synthetic (seeded uniform uint8 lines, seeded synthetic weights)",
        "config": {"workload": "configs[1]: CRNN %s, batch 256 synthetic 32x280 grayscale lines per step per GPU, "
                               "uint8 lines resident in HBM -> decoded strings on the host" % args.precision,
                   "batch": BATCH, "width": WIDTH, "nclass": NCLASS, "lines_per_step": BATCH * world,
                   "lanes_per_gpu": NL, "gpu_max_hw_queues": _hw_queues(), "split_timeouts": split_tmo,
                   "persistent_blocks": {"pipelined_launches": int(os.environ.get("CRNN_PIPELINE_BLOCKS", "192")),
                                         "alone_on_the_device (the per-kernel table and roofline)": 256},
                   "timed_region": "normalise -> conv stack -> 2x BiLSTM -> classifier arg-max -> CTC collapse on the GPU, label "
                                   "copy-back and string assembly on the host; NOT in it: upload and resize of source images "
                                   "(tools/bench_configs.py measures those: crop replay)",
                   "sharding": "independent lines, one replica per GPU, no collective"},
        "roofline": {
            "bound": "mfma", "kernel": dom["kernel"],
            "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s",
            "frac": dom["frac"],
            "what_achieved_counts": ("MFMA FLOPs the kernel executes per second, measured alone on the device: %d MFMA product(s) per fp32 "
                                     "product of the Winograd component GEMMs, against the dense peak of that MFMA type"
                                     % dom["mfma_products_per_fp32_product"]),
            "fp32_equivalent_tflops": dom["fp32_equivalent_tflops"],
            "fp32_equivalent_over_f32_mfma_peak": dom["fp32_equivalent_over_f32_mfma_peak"],
            "direct_form_tflops": dom.get("direct_form_tflops"),
            "direct_form_over_f32_mfma_peak": dom.get("direct_form_over_f32_mfma_peak"),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (avg)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch_avg": dom["algorithmic_bytes_per_launch_avg"],
            "flops_per_launch_avg": dom["flops_per_launch_avg"],
            "ms_per_launch_avg": dom["ms_per_launch_avg"]},
        "roofline_other": [{k: v for k, v in f.items() if k != "launches"} for f in fams[1:]],
        "hbm_stages": pre_stages + hbm_stages,
        # direct-form algorithmic FLOPs (4.283 GFLOP per line) x lines/s: an "effective" rate that Winograd, the
        # classifier pre-filter and the f16x3 GEMMs push past the f32 MFMA peak; it is not a roofline figure
        "pipeline_tflops": round(FLOP_PER_LINE * value / 1e12, 2),
        "pipeline_effective_frac_of_f32_mfma_peak": round(FLOP_PER_LINE * value / 1e12 / (PEAK_F32_MFMA_TFLOPS * world), 4),
        # sum over a step's launches of (MFMA FLOPs executed / dense peak of that MFMA type) / step time
        "pipeline_mfma_time_frac": round(mfma_s / step_s, 4),
        "serial_kernel_ms_per_step": round(serial_ms, 4),
        "kernels": kernels,
    }
    if fp32 and os.environ.get("CRNN_CLS_FILTER", "1") != "0":
        out["config"]["classifier"] = ("arg-max decided in fp32: a bf16 GEMM with a rigorous error bound only prunes the "
                                       "5530 classes to the candidates (about 4 per row here), which are scored exactly in fp32 "
                                       "(DESIGN.md section 4; CRNN_CLS_FILTER=0 = plain fp32 GEMM)")
    if h3:
        out["config"]["gemm_arithmetic"] = ("f16x3: every fp32 GEMM operand is carried as a (hi, lo) fp16 pair with a power-of-two "
                                            "scale per row block, products hi*hi + hi*lo + lo*hi on the fp16 MFMA with fp32 "
                                            "accumulation -- measured as accurate as the f32 MFMA against float64 "
                                            "(tools/split_mfma_test.hip; DESIGN.md section 4); CRNN_F16X3=0 = f32 MFMA")
    if plain:
        out["alternatives_same_run"] = plain
    if host_path:
        out["gpu_host_path_same_inputs_as_cpu_baseline"] = host_path
    if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N=1 only
        cb, cpu_texts = cpu_baseline(host_lines[0], sd, keys.alphabetChinese)
        out["cpu_baseline"] = cb
        same = sum(a == b for a, b in zip(first_texts, cpu_texts))
        out["cpu_baseline"]["strings_identical_to_gpu"] = "%d/%d" % (same, len(cpu_texts))
        out["cpu_baseline"]["char_agreement_gpu_vs_cpu_fp32"] = round(char_agreement(first_texts, cpu_texts), 4)
    print(json.dumps(out, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
