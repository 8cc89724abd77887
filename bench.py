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
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

# the lanes are HIP streams: give the runtime enough hardware queues for them (before HIP initialises;
# see ocr_amd/crnn/_capi.py _more_hw_queues)
if not os.environ.get("CRNN_KEEP_HW_QUEUES") and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, WIDTH, NCLASS = 256, 280, 5530
FLOP_PER_LINE = 4283.46e6          # 2*MACs per 32x280 line, nclass 5530
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lanes", type=int, default=4, help="independent stream/workspace lanes per GPU")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="fp32 = the headline, parity-exact path; bf16 = bf16 MFMA inputs, fp32 accumulate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="one batch at a time (for rocprofv3 --kernel-trace --stats: per-kernel durations that are "
                         "not stretched by the other lanes, with the tile shapes the pipelined run uses)")
    args = ap.parse_args()

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
            lane = mdl.forward_device(d_lines[k % nslots].data_ptr(), BATCH, WIDTH, d_lab[s].data_ptr(),
                                      d_len[s].data_ptr())
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
    model.set_profiling(False)

    # ---- the same pipeline with the classifier's arg-max taken from a plain fp32 GEMM (no bf16 pre-filter), for
    # reference beside the headline (rank 0 of a one-GPU run only; outside the timed region above)
    plain = None
    if (args.precision == "fp32" and world == 1 and not args.serial
            and os.environ.get("CRNN_CLS_FILTER", "1") != "0" and not args.no_cpu_baseline):
        # pinned result buffers remember the lane streams they were used on: free them (the closure) before the handle
        run = None
        gc.collect()
        model.__del__()          # frees the first handle's streams: both handles' lanes would share the hardware queues
        os.environ["CRNN_CLS_FILTER"] = "0"
        model2 = CRNN(32, 1, NCLASS, 256, leakyRelu=False, lstmFlag=True, GPU=True, alphabet=keys.alphabetChinese,
                      device=local_rank, lanes=args.lanes, precision=args.precision)
        del os.environ["CRNN_CLS_FILTER"]
        model2.load_state_dict(sd)
        run2, _, _ = pipeline(model2)
        texts2 = run2(1)
        run2(args.warmup)
        model2.synchronize()
        t2 = time.perf_counter()
        run2(args.steps)
        model2.synchronize()
        e2 = time.perf_counter() - t2
        plain = {"value": round(BATCH * args.steps / e2, 1), "unit": "text-lines/s",
                 "ms_per_step": round(e2 / args.steps * 1e3, 3),
                 "strings_identical_to_headline_path": "%d/%d" % (sum(a == b for a, b in zip(first_texts, texts2)), len(texts2))}
        run2 = None
        gc.collect()
        del model2

    run = None          # same order at every exit: pinned buffers (held by the closure) go before the handle's streams
    gc.collect()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    lines_total = BATCH * args.steps * world
    value = lines_total / elapsed
    gemm_names = [n for n in order if (n.startswith("conv") and n != "conv0_pool0" and "_win" not in n and "_wout" not in n)
                  or n in ("lstm0_ih", "lstm1_ih", "emb0", "classifier")]
    # GEMM kernels of the fp32 path: conv2-5 run as Winograd F(4x4) and conv6 as 1-D F(4,2) ("<layer>_f4":
    # wino4_gemm_lw_kernel), conv1 as fused F(2x2) (a "<layer>_win" launch before it: wino_gemm_dma_kernel), the LSTM input projections and the
    # classifier on lin_gemm_kernel (same persistent direct-to-LDS pipeline as wino4_gemm_kernel), conv6 and the
    # embedding on conv_gemm_kernel.  The roofline object is for whichever kernel takes most of the step
    # (rocprof's per-kernel average is over the same launches).
    wino_names = [n for n in gemm_names if (n.split("_")[0] + "_win") in order]
    wino4_names = [n for n in gemm_names if n.endswith("_f4")]
    lin_on = args.precision == "fp32" and os.environ.get("CRNN_LIN_GEMM", "1") != "0"
    lin_names = [n for n in gemm_names if lin_on and n in ("lstm0_ih", "lstm1_ih", "emb0", "classifier")]
    direct_names = [n for n in gemm_names if n not in wino_names + wino4_names + lin_names]
    direct_kernel = "conv_gemm_kernel" if args.precision == "fp32" else "conv_gemm_bf16_kernel"

    def family(kernel, names, what):
        ms = sum(acc[n][0] for n in names)
        fl = sum(acc[n][1] for n in names)
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"kernel": "%s (%s; %d launches per step)" % (kernel, what, len(names)), "launches": names,
                "ms_per_step": round(ms, 4), "achieved": round(ach, 2), "frac": round(ach / peak, 4),
                "flops_per_launch_avg": fl / max(1, len(names)),
                "algorithmic_bytes_per_launch_avg": sum(acc[n][2] for n in names) / max(1, len(names)),
                "ms_per_launch_avg": round(ms / max(1, len(names)), 4)}
    fams = []
    if wino_names:
        fams.append(family("wino_gemm_dma_kernel", wino_names,
                           "Winograd F(2x2,3x3) GEMM + output transform, executed MFMA FLOPs; a launch here is one layer = "
                           "the 128x128-tile launch over the full rounds plus the small-tile launch over the remaining tiles"))
    if wino4_names:
        fams.append(family("wino4_gemm_lw_kernel", wino4_names,
                           "Winograd component GEMMs of a layer -- 36 for F(4x4,3x3) on conv2-5, 5 for conv6's 1-D F(4,2) -- "
                           "as one persistent launch of 128x128 units, executed MFMA FLOPs"))
    if lin_names:
        fams.append(family("lin_gemm_kernel", lin_names, "plain GEMMs as persistent launches of 128x128 units: " +
                           ", ".join(lin_names)))
    if direct_names:
        fams.append(family(direct_kernel, direct_names, "implicit GEMM: " + ", ".join(direct_names)))
    fams.sort(key=lambda f: -f["ms_per_step"])
    dom = fams[0]
    gemm_names = dom["launches"]
    kernels = {n: {"ms": round(acc[n][0], 4),
                   "tflops": round(acc[n][1] / (acc[n][0] * 1e-3) / 1e12, 2) if acc[n][1] else None,
                   "gbs_algorithmic": round(acc[n][2] / (acc[n][0] * 1e-3) / 1e9, 1)} for n in order}
    # HBM traffic of the same launches from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, gfx950 x2 read correction; tools/pmc_traffic.py).  PMC counters
    # cannot be collected from inside this process, so the figure comes from profiles/.
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "r1_pmc_traffic_%s.json" % args.precision)
    if os.path.exists(tp):
        with open(tp) as f:
            tj = json.load(f)
        if all(n in tj for n in gemm_names):
            # a Winograd layer is two launches (full rounds + "<layer>_tail"): count both
            traffic = sum(tj[n]["hbm_bytes"] + tj.get(n + "_tail", {}).get("hbm_bytes", 0.0) for n in gemm_names) / len(gemm_names)
            traffic_src = "profiles/" + os.path.basename(tp)

    out = {
        "metric": "text-lines/sec at 32x280 crops (CRNN ocr-lstm head, 5530 classes)",
        "value": round(value, 1), "unit": "text-lines/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16",
        "data": "synthetic (seeded uniform uint8 lines, seeded synthetic weights)",
        "config": {"workload": "configs[1]: CRNN %s, batch 256 synthetic 32x280 grayscale lines per step per GPU, "
                               "uint8 lines resident in HBM -> decoded strings on the host" % args.precision,
                   "batch": BATCH, "width": WIDTH, "nclass": NCLASS, "lines_per_step": BATCH * world,
                   "lanes_per_gpu": NL,
                   "sharding": "independent lines, one replica per GPU, no collective"},
        "roofline": {
            "bound": "mfma", "kernel": dom["kernel"],
            "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch (avg)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch_avg": dom["algorithmic_bytes_per_launch_avg"],
            "flops_per_launch_avg": dom["flops_per_launch_avg"],
            "ms_per_launch_avg": dom["ms_per_launch_avg"]},
        "roofline_other": [{k: v for k, v in f.items() if k != "launches"} for f in fams[1:]],
        # direct-form algorithmic FLOPs (4.283 GFLOP per line) x lines/s.  The fp32 path runs
        # conv1-5 as Winograd F(2x2,3x3) (2.25x fewer multiplies), so this "effective" rate can approach
        # or pass the MFMA peak; roofline.achieved above counts the FLOPs the MFMAs actually execute.
        "pipeline_tflops": round(FLOP_PER_LINE * value / 1e12, 2),
        "pipeline_effective_frac_of_mfma_peak": round(FLOP_PER_LINE * value / 1e12 / (peak * world), 4),
        # FLOPs the MFMAs execute per step (every GEMM launch + the recurrences) / step time: what the whole
        # pipeline, lanes overlapped, sustains.  roofline.achieved is the dominant kernel measured ALONE on the
        # device; with several lanes the library picks the 128x128 Winograd tile, which is slower alone at this
        # batch (2.25 rounds of one block per CU) but faster in the pipeline, where other lanes fill its tail.
        "pipeline_executed_mfma_tflops": round(sum(acc[n][1] for n in order if n != "conv0_pool0") * world
                                               / (elapsed / args.steps) / 1e12, 2),
        "kernels": kernels,
    }
    if args.precision == "fp32" and os.environ.get("CRNN_CLS_FILTER", "1") != "0":
        out["config"]["classifier"] = ("arg-max decided in fp32: a bf16 GEMM with a rigorous error bound only prunes the "
                                       "5530 classes to the candidates (about 4 per row here), which are scored exactly in fp32 "
                                       "(DESIGN.md section 4; CRNN_CLS_FILTER=0 = plain fp32 GEMM)")
        if plain:
            out["classifier_plain_fp32_gemm"] = plain
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
