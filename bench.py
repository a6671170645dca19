#!/usr/bin/env python
"""bench.py — fragments/sec of the MI355X per-fragment 3D path on BASELINE.json's config 2.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A bare `python bench.py --gpus N` (no WORLD_SIZE in the environment) spawns the N ranks itself — one process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT set per child — the launch model of the
reference's main.py:67-76 (`torch.multiprocessing.spawn`-style, one rank per device).

One "step" = one pass of the hot path over ONE synthetic 9-view 640x480 fragment window whose
inputs (feature pyramids, projection matrices, voxel lists) are already resident in HBM:
the stages listed in `Cfg2Step` (eprecon_amd.fragment_step).  With N ranks every rank processes its
own fragment per step (independent fragment windows shard one-per-GPU), so
value = N * K / max-over-ranks(time) and scaling is "weak".

Prints ONE JSON line on rank 0 with the driver's keys plus
  roofline      the kernel BASELINE.json's metric names (bp_gather on the dense 96^3 / C=24 / 120x160 level), HIP-event
                timed on its launch stream; `traffic` from the newest committed PMC passes of that kernel
  roofline_conv one launch of the gather-GEMM convolution family (the largest share of kernel time), HIP-event timed,
                against the dense fp32-MFMA peak
  extra         cfg3 / cfg4 (whole NeuConNet.forward per fragment) timed in the same process, launches per fragment
  cpu_baseline  the CPU oracle (C + OpenMP port of the reference algorithm) on the same windows.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured streaming ceiling)
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, f32 in / f32 accumulate
DOMINANT_KERNEL = "bp_gather_mlp_kernel<256,MEAN,6,1>"
L1_PEAK_LINES = 256 * 2.4e9  # nominal: 256 CUs x one 64-byte segment per clock x 2.4 GHz (used only when no probe file is committed)


def l1_peak_from_probe():
    """The vector-L1 segment rate the roofline's `l1` record is measured against: CALIBRATED by tools/probes/l1_line_rate.hip on
    the gather's own access shape (16-byte loads, six lanes per 96-byte texel, L1-resident table), read from the newest committed
    profiles/rNN/l1_line_rate.txt -> (segments per second, source)."""
    import re
    path = newest_profile("l1_line_rate.txt")
    if path:
        text = open(path).read()
        m = re.search(r"# PMC-calibrated peak \(mode 0, L1-resident\): ([0-9.]+) G accesses/s = ([0-9.]+) per clock per CU", text)
        if m:   # the probe's rate counted by the same counter as the gather's accesses (TCP_TOTAL_CACHE_ACCESSES)
            return float(m.group(1)) * 1e9, (f"{os.path.relpath(path, ROOT)}: probe mode 0 (96-byte texels, 6 lanes each, L1-resident), "
                                             f"TCP_TOTAL_CACHE_ACCESSES per second = {m.group(1)} G/s = {m.group(2)} per clock per CU")
        block = text.split("## L1-resident")[1].split("##")[0] if "## L1-resident" in text else ""
        m = re.search(r"mode 0:.*?([0-9.]+) G segments/s = ([0-9.]+) per clk per CU", block)
        if m:
            return float(m.group(1)) * 1e9, (f"{os.path.relpath(path, ROOT)}: mode 0 (96-byte texels, 6 lanes each, L1-resident) "
                                             f"{m.group(1)} G segments/s = {m.group(2)} per clock per CU")
    return L1_PEAK_LINES, "nominal 256 CUs x 1 segment / clk x 2.4 GHz (no committed probe output found)"


# The cfg4 workloads time the drop-in contract: every fragment complete when forward() returns.  The pipelined serving mode
# (the panoptic branch of fragment k issued by a worker thread on its own stream while the main thread runs fragment k + 1:
# NeuConNet.panoptic_stream / panoptic_worker, identical results, tests/test_cfg4_gpu.py) bought 2.3 ms per fragment while
# the fragment was host-bound (round 2) and nothing since it is GPU-bound (round 4: 14.4 against 14.0 ms).
# EPRECON_CFG4_PIPELINE=1: `--workload cfg4` and the cfg5 leg run pipelined; =0: the `extra` leg skips its pipelined figure
# (it reports both by default).
_PIPELINE_ENV = os.environ.get("EPRECON_CFG4_PIPELINE")
PIPELINE = _PIPELINE_ENV == "1"             # standalone cfg4 / cfg5 workloads
PIPELINE_FIGURE = _PIPELINE_ENV != "0"      # the additional pipelined figure of the `extra` leg


def newest_profile(name):
    """newest committed profiles/rNN/<name> (the per-round directories sort by round number)"""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", name)))
    return hits[-1] if hits else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="approximate CPU time to spend on the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3 / cfg4 timings reported under `extra`")
    ap.add_argument("--scenes-per-gpu", type=int, default=0,
                    help="with --workload cfg4: K independent scenes on ONE GPU at the same time, each in its own process with "
                         "its own NeuConNet, map handles and streams (SURVEY.md 8e: fragments of different scenes are pure "
                         "replicas); value = fragments/s over all K scenes")
    ap.add_argument("--scene-seed-base", type=int, default=0, help="scene k of --scenes-per-gpu runs Cfg4Step(seed = base + k)")
    ap.add_argument("--scene-child", type=int, default=-1, help=argparse.SUPPRESS)   # (internal: one scene of --scenes-per-gpu)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "train"],
                    help="cfg2 (default) = BASELINE.json configs[1], the headline metric; cfg4 = the full "
                         "coarse-to-fine forward over 4 sequential fragments (extra measurement, no roofline)")
    return ap.parse_args()


def cpu_baseline(step, seconds):
    """The same step on the host cores (`kind` = "port"): oracle/c/*.c, a C + OpenMP port of the reference's back-projection;
    the spconv layers as PyTorch-CPU gather -> matmul -> index_add_ per kernel offset (oracle/torch_cpu.py: the shape the
    reference's own CPU path would take); the 2D fusion convolutions as the same PyTorch modules on the CPU threads; the stage-0
    selection in numpy.  Protocol (SURVEY.md 8d with one warm-up step instead of three: a step takes about a second): whole
    steps until `seconds` are spent and at least 5 are done (10 at most); the MEDIAN step is reported, with its per-stage split."""
    import torch
    from oracle import back_project as O
    from oracle import grid_ops as OG
    from oracle import torch_cpu as TC
    from eprecon_amd import synthetic as S
    from eprecon_amd.fragment_step import LEVELS

    w = step.window
    origin = w["vol_origin_partial"][None]
    feats = [f.cpu().numpy() for f in step.feats]
    kr = [np.ascontiguousarray(w["proj_matrices"][:, l][:, None]) for l in range(3)]
    coords = {iv: S.dense_coords(w["n_vox"], iv) for iv in (4, 2, 1)}
    from eprecon_amd.config import CH_IMG, CH_INIT_DOWN, N_VIEWS
    from eprecon_amd.occupancy_initialization import Occupancy_Initialization
    net = Occupancy_Initialization(CH_IMG, CH_INIT_DOWN, N_VIEWS)   # same weights, on the host
    net.load_state_dict({k: v.detach().cpu() for k, v in step.init_net.state_dict().items()})
    net.train()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    f_init = [[t.cpu() for t in view] for view in step.features_init]
    # one thread budget for all CPU engines (OpenMP oracle, torch, numpy BLAS): more than ~64 threads only adds fork / join
    # overhead on these sizes
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    O.set_threads(threads)
    from threadpoolctl import threadpool_limits
    blas_limit = threadpool_limits(limits=threads)

    def one_step():
        t = [time.perf_counter()]
        lap = lambda: t.append(time.perf_counter())
        with torch.no_grad():
            fused = net.feat_fusion_pre(torch.stack([f[2][0] for f in f_init]), torch.stack([f[1][0] for f in f_init]),
                                        torch.stack([f[0][0] for f in f_init])).unsqueeze(1).numpy()
            lap()
            r = O.back_project(coords[2], origin, w["voxel_size"], fused, kr[1], 2, O.MODE_VARIANCE)
            lap()
            pairs = TC.kernel_map_pairs(r["coords"], 2)
            lap()
            logit = TC.sparse_stack(sd, torch.from_numpy(r["feats"]), pairs).numpy()
            lap()
        OG.init_select(logit, r["coords"], 1)
        lap()
        for _, lvl, interval, mv in LEVELS:
            O.back_project(coords[interval], origin, w["voxel_size"], feats[lvl], kr[lvl], mv)
            lap()
        names = ["fusion2d_torch_cpu", "variance_back_project_c_omp", "kernel_map_numpy", "sparse_stack_torch_cpu", "init_select_numpy",
                 "bp24_c_omp", "bp48_c_omp", "bp96_c_omp"]
        return t[-1] - t[0], {n: (b_ - a_) * 1e3 for n, a_, b_ in zip(names, t[:-1], t[1:])}

    one_step()
    runs, t0 = [], time.perf_counter()
    while len(runs) < 5 or (len(runs) < 10 and time.perf_counter() - t0 < seconds):
        runs.append(one_step())
    dt = time.perf_counter() - t0
    blas_limit.restore_original_limits()
    runs.sort(key=lambda r: r[0])
    med, stages = runs[len(runs) // 2]
    return {"value": 1.0 / med, "unit": "fragments/s", "cores": threads, "kind": "port",
            "sample": f"{len(runs)} whole steps of the same workload (same window, seed {step.seed}) in {dt:.1f} s, median step",
            "ms_per_step": med * 1e3, "stages_ms": {k: round(v, 1) for k, v in stages.items()},
            "protocol": "1 warm-up step, then 5-10 whole steps, median (SURVEY.md 8d's protocol with one warm-up instead of "
                        "three); C + OpenMP back-projection, PyTorch-CPU sparse layers (gather / matmul / index_add_ per "
                        "offset) and 2D convolutions, numpy kernel map and stage-0 selection"}


def conv_family_roofline(name="conv_shapes.txt"):
    """The figure that tracks fragments/s: the whole gather-GEMM family over the 30 distinct 3x3x3 shapes of
    a cfg4 fragment, from the newest committed profiles/rNN/conv_shapes.txt (tools/conv_shapes_ab.py: HIP-event time of each
    shape alone on the device, flops on LIVE kernel-map pairs) — sum of flops / sum of time against the fp32-MFMA peak."""
    import re
    path = newest_profile(name)
    if not path:
        return None
    us, flops, rows = 0.0, 0.0, []
    for line in open(path):
        m = re.search(r"N=\s*(\d+)\s+(\d+)->\s*(\d+)\s+([0-9.]+) us\s+([0-9.]+) TF", line)
        if m:
            t, tf = float(m.group(4)), float(m.group(5))
            us += t
            flops += tf * 1e12 * t * 1e-6
            rows.append((line.split("N=")[0].strip(), int(m.group(1)), int(m.group(2)), int(m.group(3)), t, tf))
    if not rows:
        return None
    lead = max(rows, key=lambda r: r[4])
    return {"bound": "mfma", "shapes": len(rows), "sum_us": round(us, 1), "sum_gflop_live": round(flops / 1e9, 2),
            "achieved": flops / (us * 1e-6) / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": flops / (us * 1e-6) / 1e12 / F32_MFMA_PEAK_TF,
            "slowest_shape": {"name": lead[0], "rows": lead[1], "cin": lead[2], "cout": lead[3], "us": lead[4], "tflops": lead[5]},
            "source": os.path.relpath(path, ROOT) + " (recorded: each shape timed alone by tools/conv_shapes_ab.py)"}


def _conv_roofline_record(samples, kvol, cin, cout, what):
    """armed launches of the gather-GEMM family on ONE voxel set -> the roofline record (mean time of the samples); live
    kernel-map pairs are counted by the library on the launch stream right behind each timed launch
    (eprecon_profile_conv_pairs)"""
    rows, pairs, issued, name = samples[-1]["rows"], samples[-1]["pairs"], samples[-1]["issued"], samples[-1]["name"]
    if issued <= 0:
        issued = rows * kvol
    t = float(np.mean([x["ms"] for x in samples]))
    flops = 2.0 * pairs * cin * cout
    return {"bound": "mfma", "kernel": f"{name.decode()} ({what}, {rows} voxels)", "flops": flops,
            "executed_flops": 2.0 * issued * cin * cout, "executed_over_live": issued / max(pairs, 1),
            "output_stationary_flops": 2.0 * rows * kvol * cin * cout, "live_pairs": pairs, "avg_launch_ms": t,
            "launches_timed": len(samples), "achieved": flops / (t * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": flops / (t * 1e-3) / 1e12 / F32_MFMA_PEAK_TF}


def _armed_conv_samples(lib, run, kvol, cin, cout, min_rows, repeats):
    """one armed launch per call of `run`: [{ms, rows, pairs, issued, name}]"""
    import ctypes
    out = []
    for _ in range(repeats):
        lib.eprecon_profile_conv_arm(kvol, cin, cout, min_rows)
        run()
        r, k = ctypes.c_int64(0), ctypes.c_char_p()
        t = float(lib.eprecon_profile_conv_ms(ctypes.byref(r), ctypes.byref(k)))
        if t > 0:
            out.append({"ms": t, "rows": int(r.value), "name": k.value or b"", "pairs": int(lib.eprecon_profile_conv_pairs()),
                        "issued": int(lib.eprecon_profile_conv_executed_pairs())})
    return out


def conv_roofline(step, lib, repeats=3):
    """HIP-event time of ONE launch of the convolution family: the first 3x3x3 32->32 submanifold layer of the
    initialisation stack on the ~94k-voxel valid set (the layer DESIGN.md 3b analyses), measured in untimed
    extra steps.  flops = 2 * (live kernel-map pairs) * C_in * C_out (algorithmic; the output-stationary kernels
    also issue the MFMAs of empty neighbours: `executed_flops`)."""
    samples = _armed_conv_samples(lib, step.run, 27, 32, 32, 20000, repeats)
    if not samples or step.last.get("init") is None:
        return None
    return _conv_roofline_record(samples, 27, 32, 32,
                                 "submanifold 3x3x3, 32->32 + fused LayerNorm epilogue, valid set of the dense 48^3 grid")


def conv_roofline_cfg4(step, lib, cycles=2):
    """the same for the instance that leads the cfg4 profile: the 3x3x3 48->24 convolutions of the finest-level ConvGRU
    ([h, x] -> gate, models/modules.py:178-222).  One armed launch per fragment (the first that matches) over `cycles` passes
    of the scene; the fragments of a scene have different voxel sets, so only the launches on the LARGEST one (the scene's
    last fragment) enter the record — time, rows and live pairs of the same instance."""
    samples = _armed_conv_samples(lib, step.run, 27, 48, 24, 20000, cycles * step.n_fragments)
    if not samples:
        return None
    top = max(x["rows"] for x in samples)
    rec = _conv_roofline_record([x for x in samples if x["rows"] == top], 27, 48, 24,
                                "ConvGRU gate convolution 3x3x3, 48->24, finest level, in situ")
    # In situ the voxel- and the image-ConvGRU issue this same convolution on two streams at the same time: the two launches
    # share the chip and each takes up to twice as long as alone.  The launch alone on the device is measured by
    # tools/conv_cfg4_instance.py under rocprofv3 (committed rows); reported next to the in-situ figure while it names the same set.
    rec["in_situ_note"] = "the twin ConvGRU issues the same convolution on a second stream at the same time: the launches share the chip"
    inst = newest_profile("conv_cfg4_instance.txt")
    if inst:
        import re
        text = open(inst).read()
        m_rows = re.search(r"rows (\d+) .* live pairs (\d+)", text)
        m_avg = re.search(r"rocprofv3 kernel trace: (\d+) launches .* avg ([0-9.]+) us", text)
        if m_rows and m_avg and abs(int(m_rows.group(1)) - top) <= top // 100:
            us = float(m_avg.group(2))
            flops = 2.0 * int(m_rows.group(2)) * 48 * 24
            rec["alone"] = {"avg_launch_ms": us * 1e-3, "launches": int(m_avg.group(1)), "rows": int(m_rows.group(1)),
                            "achieved": flops / (us * 1e-6) / 1e12, "frac": flops / (us * 1e-6) / 1e12 / F32_MFMA_PEAK_TF,
                            "source": os.path.relpath(inst, ROOT) + " (rocprofv3 --kernel-trace rows of the launch alone)"}
    return rec


def _timed(run, steps, sync, after=None):
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    if after is not None:
        after()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


def extra_workloads(device, lib, steps3=12, steps4=16, warm4=4):
    """cfg3 / cfg4 (BASELINE.json configs[2] / [3]), the whole drop-in boundary (NeuralRecon.forward) and one optimisation
    step, timed in the same process after the headline.  Not the headline metric (that is cfg2), reported under `extra`.
    Every leg is guarded on its own: a failing leg reports `<leg>_error` and never costs the others."""
    import torch
    from eprecon_amd.fragment_step import Cfg4Step, E2EStep, TrainStep
    sync = torch.cuda.synchronize
    out = {}

    def leg(name, fn):
        try:
            # Every leg starts with the previous leg's networks really gone.  They sit in reference cycles, and until the
            # collector runs their HIP-graph executables (three parallel branches each) and streams stay alive: the
            # multi-stream parts of the NEXT leg then measured 23-27 ms per e2e fragment instead of 18.5-19.6 (four boxes;
            # bisected down to this line).
            import gc
            sync()
            gc.collect()
            # (what survives is long-lived: out of the collector's way — a full collection over the heap the earlier legs left
            # behind, triggered from inside the ~7,000-launch optimisation step, cost that leg 8 ms per step)
            if os.environ.get("EPRECON_BENCH_GC_FREEZE", "1") == "1":
                gc.freeze()
            torch.cuda.empty_cache()
            fn()
        except Exception as exc:  # noqa: BLE001
            out[f"{name}_error"] = f"{type(exc).__name__}: {exc}"

    def cfg34():
        step = Cfg4Step(seed=0, device=device, pipeline=False)
        for _ in range(2):
            step.run_cfg3()
        out["cfg3_ms_per_fragment"] = _timed(step.run_cfg3, steps3, sync)
        out["cfg3_workload"] = "one 9-view 640x480 fragment, empty map: occupancy init + 3 x [Back_Project, SPVCNN, GRU " \
                               "fusion, TSDF / occupancy heads, sparsify], no panoptic decoder"
        for _ in range(warm4):
            step.run()
        step.voxels.clear()
        # step.run raises if a fragment returns before the finest level.  The drop-in figure: every fragment complete
        # (panoptic_info included) when NeuConNet.forward returns, nothing carried across fragments.
        from eprecon_amd import _lib as L
        reads0 = L.HOST_READS
        out["cfg4_unpipelined_ms_per_fragment"] = _timed(step.run, steps4, sync)
        # device -> host reads the host thread blocks on, counted at their call sites (eprecon_amd._lib.count_host_read):
        # element counts the next launches are sized by, the reference's guards, the panoptic post-processing
        out["blocking_reads_per_fragment"] = (L.HOST_READS - reads0) / steps4
        out["cfg4_ms_per_fragment"] = out["cfg4_unpipelined_ms_per_fragment"]
        out["cfg4_fragments_per_sec"] = 1e3 / out["cfg4_ms_per_fragment"]
        out["cfg4_finest_voxels_min_max"] = [min(step.voxels), max(step.voxels)]
        out["cfg4_early_returns"] = step.early_returns
        out["cfg4_workload"] = step.describe()["workload"]
        out["roofline_conv_cfg4"] = conv_roofline_cfg4(step, lib)
        # beside, never instead of, the exact-fp32 figure: the direct gather kernel's opt-in bf16x3 operand form (read per
        # launch; error budget in tests/test_sparse_gpu.py::test_bf16x3_opt_in_is_within_its_error_budget)
        os.environ["EPRECON_CONV_BF16X3"] = "1"
        try:
            for _ in range(2):
                step.run()
            out["cfg4_bf16x3_opt_in_ms_per_fragment"] = _timed(step.run, max(4, steps4 // 2), sync)
        finally:
            del os.environ["EPRECON_CONV_BF16X3"]
        if PIPELINE_FIGURE:
            # throughput mode (opt-in for NeuConNet, other outputs contract): the panoptic branch of fragment k issued by a
            # worker thread on its own stream while the main thread runs fragment k + 1; the last fragment's branch
            # belongs to the timed region (flush)
            step.set_pipeline(True)
            for _ in range(warm4):
                step.run()
            out["cfg4_pipelined_ms_per_fragment"] = _timed(step.run, steps4, sync, after=step.flush)
            out["cfg4_pipelined"] = step.describe()["pipelined"]
            step.set_pipeline(False)

    def e2e():
        step = E2EStep(seed=0, device=device)
        for _ in range(4):
            step.run()
        step.voxels.clear()
        step.early_returns = 0
        out["e2e_ms_per_fragment"] = _timed(step.run, 12, sync)
        out["e2e_early_returns"] = step.early_returns
        out["e2e_finest_voxels_min_max"] = [min(step.voxels), max(step.voxels)] if step.voxels else None
        out["e2e_workload"] = step.describe()["workload"]

    def train():
        # lr 1e-6: with seeded-random weights on noise features the occupancy logits of a level differ by ~1e-2 between
        # voxels, so ONE Adam step at the reference's 1e-4 (every parameter moves by lr in a coherent direction) shifts
        # them all by about a standard deviation and the next forward keeps 80 % instead of 35 % of a level (measured:
        # step 2 runs into the reference's 1.5 x cap guard).  Adam's arithmetic does not depend on lr; a small one keeps
        # the nine steps in the calibrated sparsity regime so that every timed step is the full workload.
        step = TrainStep(seed=0, device=device, lr=1e-6)
        step.raise_on_early_return = False     # counted instead: the number below is reported only over FULL steps
        for _ in range(3):
            step.run()
        full_ms, n_early = [], 0
        for _ in range(6):
            before = step.early_returns
            ms = _timed(step.run, 1, sync)
            if step.early_returns == before:
                full_ms.append(ms)
            else:
                n_early += 1
        out["train_ms_per_step"] = float(np.mean(full_ms)) if full_ms else None
        out["train_full_steps"] = len(full_ms)
        out["train_early_returns"] = n_early
        out["train_finest_voxels_min_max"] = [min(step.voxels), max(step.voxels)] if step.voxels else None
        out["train_workload"] = step.describe()["workload"] + " (lr 1e-6, see bench.py)"
        # the reference trains at BATCH_SIZE 4 (config/train.yaml:2): the same step on a batch of TWO consecutive windows
        del step
        import gc
        gc.collect()
        step = TrainStep(seed=0, device=device, lr=1e-6, batch=2)
        step.raise_on_early_return = False
        for _ in range(2):
            step.run()
        full_ms = []
        for _ in range(4):
            before = step.early_returns
            ms = _timed(step.run, 1, sync)
            if step.early_returns == before:
                full_ms.append(ms)
        out["train_b2_ms_per_step"] = float(np.mean(full_ms)) if full_ms else None
        out["train_b2_full_steps"] = len(full_ms)
        out["train_b2_finest_voxels_min_max"] = [min(step.voxels), max(step.voxels)] if step.voxels else None

    def scenes():
        # a cfg4 fragment keeps the GPU "busy" with ~600 launches that occupy a fraction of its 256 CUs;
        # K independent scenes side by side (one process each: K Python threads would share one interpreter lock) fill it.
        # This process keeps its context but queues nothing while the children run.
        sync()
        for k in (1, 2, 4):
            rec = scenes_per_gpu(k, 16, 8)
            out[f"cfg4_fragments_per_sec_k{k}"] = rec["fragments_per_sec"]
            out[f"cfg4_ms_per_fragment_by_scene_k{k}"] = rec["ms_per_fragment_by_scene"]
        # the same for the configuration the headline is quoted on: K streams of the cfg2 step on the one GPU
        for k in (2, 4):
            rec = scenes_per_gpu(k, 100, 10, workload="cfg2")
            out[f"cfg2_fragments_per_sec_k{k}"] = rec["fragments_per_sec"]
            out[f"cfg2_ms_per_step_by_stream_k{k}"] = rec["ms_per_fragment_by_scene"]
        out["cfg4_scenes_per_gpu_note"] = ("K independent scenes on ONE GPU, one process each, started together: aggregate "
                                            "fragments/s = K x 16 / (last finish - first start); per-scene outputs are "
                                            "bit-identical to the solo run (tests/test_bench_records.py)")

    leg("cfg34", cfg34)
    leg("e2e", e2e)
    leg("train", train)
    leg("scenes", scenes)
    prof = newest_profile("cfg4_kernel_stats.json")
    if prof:
        rec = json.load(open(prof))
        out["launches_per_fragment"] = rec.get("launches_per_fragment")
        out["launches_source"] = os.path.relpath(prof, ROOT) + " (rocprofv3 --kernel-trace --stats of bench.py --workload cfg4)"
    else:
        out["launches_per_fragment"] = None
    return out


def fragment_summary(extra):
    """The figure of merit north_star names — the WHOLE coarse-to-fine fragment (BASELINE configs[3]: 4 sequential fragments,
    persistent GRU map, panoptic head) — at the top level of the line, next to the cfg2 headline the metric is quoted on:
    ms per fragment, launches and blocking host reads per fragment, the leading convolution's fraction of the fp32-MFMA peak
    (alone on the device), and the whole drop-in boundary NeuralRecon.forward."""
    conv = extra.get("roofline_conv_cfg4") or {}
    return {"cfg4_ms_per_fragment": extra.get("cfg4_ms_per_fragment"),
            "cfg4_fragments_per_sec": extra.get("cfg4_fragments_per_sec"),
            "cfg4_launches_per_fragment": extra.get("launches_per_fragment"),
            "cfg4_blocking_reads_per_fragment": extra.get("blocking_reads_per_fragment"),
            "cfg4_fragments_per_sec_per_gpu_by_scenes": {str(k): extra.get(f"cfg4_fragments_per_sec_k{k}") for k in (1, 2, 4)},
            "cfg4_roofline_conv_alone_frac": (conv.get("alone") or {}).get("frac"),
            "cfg4_roofline_conv_in_situ_frac": conv.get("frac"),
            "e2e_ms_per_fragment": extra.get("e2e_ms_per_fragment")}


def bench_cfg5(device, rank, world, dist, steps=8, warmup=4):
    """whole NeuConNet.forward per fragment with the boundary exchange on; fragments/s over all ranks"""
    import torch
    from eprecon_amd.fragment_step import Cfg4Step
    step = Cfg4Step(seed=0, device=device, rank=rank, world=world, force_exchange=True, pipeline=PIPELINE)
    step.raise_on_early_return = False
    for _ in range(warmup):
        step.run()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.run()
    step.flush()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0, float(step.early_returns)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    xch = step.net.gru_fusion._xchg
    # what every rank put on the wire (rows and bytes of its packed boundary payloads, per fragment): makes a scaling curve
    # interpretable — a rank that waits in the payload all-gather waits for the LARGEST of these
    mine = torch.tensor([float(xch.rows_sent), float(xch.bytes_sent)] if xch is not None else [0.0, 0.0], dtype=torch.float64, device=device)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    n_frag = steps + warmup
    return {"cfg5_fragments_per_sec": world * steps / elapsed, "cfg5_ms_per_step": elapsed / steps * 1e3,
            "cfg5_exchange_rows_per_fragment_by_rank": [round(float(v[0]) / n_frag, 1) for v in per_rank],
            "cfg5_exchange_bytes_per_fragment_by_rank": [round(float(v[1]) / n_frag, 1) for v in per_rank],
            "cfg5_steps": steps, "cfg5_early_returns_max_over_ranks": int(t[1].item()),
            "cfg5_collectives_per_fragment": (xch.collectives / (steps + warmup)) if xch is not None else 0,
            "cfg5_workload": f"{world} ranks x 1 fragment per step of one scene (fragments dealt round-robin), whole "
                             "NeuConNet.forward, RCCL boundary-voxel exchange (boxes, counts, and when any rank has rows to send the packed "
                             "payload of all three scales) before every fragment"}


def bench_cfg4(args, step, world, rank, dist, use_dist=False):
    """extra measurement: whole-forward fragments/s (not the headline line; no roofline / cpu legs)"""
    import torch
    for _ in range(args.warmup):
        step.run()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    from eprecon_amd import _lib as L
    reads0 = L.HOST_READS
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run()
    if hasattr(step, "flush"):
        step.flush()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0 and args.workload == "train":
        print(json.dumps({"metric": "train_fragments_per_sec", "value": world * args.steps / elapsed, "unit": "fragments/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": step.describe(), "losses_last_step": step.last}), flush=True)
    elif rank == 0:
        n_out = int(step.last["coords"].shape[0])   # Cfg4Step.run raises on an early return
        print(json.dumps({"metric": "fragments_per_sec", "value": world * args.steps / elapsed,
                          "unit": "fragments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": step.describe(),
                          "last_fragment_voxels": n_out,
                          "blocking_reads_per_fragment": (L.HOST_READS - reads0) / args.steps,
                          "finest_voxels_min_max": [min(step.voxels), max(step.voxels)]}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def _fragment_digest(out):
    """sha1 over the outputs of one fragment (voxel list, TSDF, panoptic labels + segments): what 'bit-identical' compares"""
    import hashlib
    h = hashlib.sha1()
    h.update(out["coords"].cpu().numpy().tobytes())
    h.update(out["tsdf"].cpu().numpy().tobytes())
    for info in out.get("panoptic_info", []):
        seg, segments = info["panoptic_seg"]
        h.update(seg.cpu().numpy().tobytes())
        h.update(repr([(d.get("id"), d.get("isthing"), d.get("category_id")) for d in segments]).encode())
    return h.hexdigest()


def scene_child(args):
    """one scene of `--scenes-per-gpu K`: build, warm up, hash one pass over the scene's fragments, report READY, wait for GO on
    stdin, time `--steps` fragments, report DONE with wall-clock start / end (one host: the parent compares the clocks)"""
    import torch
    from eprecon_amd import _lib as L
    from eprecon_amd.fragment_step import Cfg2Step, Cfg4Step
    torch.cuda.set_device(0)
    if args.workload == "cfg2":
        return window_child(args, Cfg2Step(seed=args.scene_seed_base + args.scene_child, device=torch.device("cuda", 0)))
    step = Cfg4Step(seed=args.scene_seed_base + args.scene_child, device=torch.device("cuda", 0), pipeline=PIPELINE)
    for _ in range(max(args.warmup, step.n_fragments)):
        step.run()
    while step.k != 0:          # align on the first fragment of the scene
        step.run()
    digests = [_fragment_digest(step.run()) for _ in range(step.n_fragments)]
    step.voxels.clear()
    torch.cuda.synchronize()
    print("READY " + json.dumps({"scene": args.scene_child, "seed": step.seed, "digests": digests}), flush=True)
    if sys.stdin.readline().strip() != "GO":
        raise SystemExit("scene child: no GO")
    reads0 = L.HOST_READS
    t_wall0, t0 = time.time(), time.perf_counter()
    for _ in range(args.steps):
        step.run()
    step.flush()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print("DONE " + json.dumps({"scene": args.scene_child, "t_start": t_wall0, "t_end": t_wall0 + elapsed, "elapsed": elapsed,
                                "ms_per_fragment": elapsed / args.steps * 1e3,
                                "blocking_reads_per_fragment": (L.HOST_READS - reads0) / args.steps,
                                "finest_voxels_min_max": [min(step.voxels), max(step.voxels)], "workload": step.describe()}), flush=True)


def window_child(args, step):
    """one window stream of `--workload cfg2 --scenes-per-gpu K`: the headline step (its own window, seed = base + index) in a
    process of its own, deferred reads as in the headline loop"""
    import torch
    defer = os.environ.get("EPRECON_CFG2_DEFER", "1") == "1"
    for _ in range(args.warmup):
        step.run()
    out = step.run()
    digest = [str(int(out["init"][0].shape[0])) if out.get("init") else "none"] + [str(int(out[k]["n_valid"])) for k in ("bp24", "bp48", "bp96") if out.get(k)]
    torch.cuda.synchronize()
    print("READY " + json.dumps({"scene": args.scene_child, "seed": step.seed, "digests": digest}), flush=True)
    if sys.stdin.readline().strip() != "GO":
        raise SystemExit("window child: no GO")
    step.defer_reads = defer
    t_wall0, t0 = time.time(), time.perf_counter()
    for _ in range(args.steps):
        step.run()
    step.flush()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print("DONE " + json.dumps({"scene": args.scene_child, "t_start": t_wall0, "t_end": t_wall0 + elapsed, "elapsed": elapsed,
                                "ms_per_fragment": elapsed / args.steps * 1e3, "blocking_reads_per_fragment": None,
                                "finest_voxels_min_max": None, "workload": step.describe()}), flush=True)


def scenes_per_gpu(k, steps, warmup, seed_base=0, timeout=900, workload="cfg4"):
    """K scene processes on device 0, started together -> the record of `bench.py --workload cfg4 --scenes-per-gpu K`
    (workload="cfg2": K streams of the headline step, one window each)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           "--scene-seed-base", str(seed_base)]
    kids = [subprocess.Popen(cmd + ["--scene-child", str(i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(k)]

    def wait_for(proc, tag):
        while True:
            line = proc.stdout.readline()
            if not line:
                raise RuntimeError(f"scene process exited before {tag} (code {proc.poll()})")
            if line.startswith(tag + " "):
                return json.loads(line[len(tag) + 1:])
    try:
        ready = [wait_for(p_, "READY") for p_ in kids]
        for p_ in kids:
            p_.stdin.write("GO\n")
            p_.stdin.flush()
        done = [wait_for(p_, "DONE") for p_ in kids]
        for p_ in kids:
            p_.wait(timeout=timeout)
    finally:
        for p_ in kids:
            if p_.poll() is None:
                p_.kill()
    span = max(d["t_end"] for d in done) - min(d["t_start"] for d in done)
    return {"scenes_per_gpu": k, "fragments_per_sec": k * steps / span, "span_s": span, "steps_per_scene": steps,
            "ms_per_fragment_by_scene": [round(d["ms_per_fragment"], 3) for d in done],
            "start_skew_ms": round((max(d["t_start"] for d in done) - min(d["t_start"] for d in done)) * 1e3, 3),
            "finest_voxels_min_max_by_scene": [d["finest_voxels_min_max"] for d in done],
            "blocking_reads_per_fragment": done[0]["blocking_reads_per_fragment"],
            "digests_by_scene": {str(r["seed"]): r["digests"] for r in ready}, "workload": done[0]["workload"]}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this one (same command line,
    the rendezvous variables torch.distributed.run would set) and wait for them.  Rank 0's JSON line goes to this process's
    stdout unchanged; the exit code is the first non-zero one of the children."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [c.wait() for c in children]
    return next((c for c in codes if c), 0)


def main():
    args = parse()
    if args.scene_child >= 0:
        return scene_child(args)
    if args.scenes_per_gpu > 0:      # (--workload cfg2 by default: state it)
        if args.workload not in ("cfg4", "cfg2") or args.gpus != 1:
            raise SystemExit("--scenes-per-gpu K goes with --workload cfg4 (or cfg2) --gpus 1")
        rec = scenes_per_gpu(args.scenes_per_gpu, args.steps, args.warmup, args.scene_seed_base, workload=args.workload)
        cfg = dict(rec.pop("workload"), scenes_per_gpu=args.scenes_per_gpu,
                   scenes="K independent scenes (seeds base .. base + K - 1), one process each on device 0, started together; "
                          "value = K x steps / (last finish - first start)")
        print(json.dumps({"metric": "fragments_per_sec", "value": rec["fragments_per_sec"], "unit": "fragments/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["span_s"] / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": cfg, **rec}), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a GPU (the HIP path has no CPU fallback) [rank {rank} of {world}]")
    torch.cuda.set_device(local_rank)
    # EPRECON_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, MAX all-reduce) at world size 1
    use_dist = world > 1 or os.environ.get("EPRECON_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from eprecon_amd import _lib
    from eprecon_amd.fragment_step import Cfg2Step

    lib = _lib.load()
    # every rank owns a different fragment window (seed = rank)
    if args.workload == "train":
        from eprecon_amd.fragment_step import TrainStep
        # data parallel: one fragment per rank and step, gradients all-reduced by DDP over RCCL
        # (lr 1e-6: see extra_workloads.train — at the reference's 1e-4 the synthetic fragment leaves its sparsity regime after a step)
        return bench_cfg4(args, TrainStep(seed=0, device=torch.device("cuda", local_rank), rank=rank, world=world, lr=1e-6),
                          world, rank, dist, use_dist)
    if args.workload == "cfg4":
        from eprecon_amd.fragment_step import Cfg4Step
        # all ranks work on ONE scene (seed 0), fragments dealt round-robin, boundary voxels exchanged
        return bench_cfg4(args, Cfg4Step(seed=0, device=torch.device("cuda", local_rank), rank=rank, world=world, pipeline=PIPELINE),
                          world, rank, dist, use_dist)
    step = Cfg2Step(seed=rank, device=torch.device("cuda", local_rank))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # EPRECON_CFG2_DEFER=0: every step reads its own counts (stage-0 selection, valid voxels per level) before it returns;
    # default: they are read after the next step is queued (Cfg2Step.defer_reads) — the same reads, the same K steps, but the
    # GPU does not idle at every step boundary while the host starts issuing the next step
    defer = os.environ.get("EPRECON_CFG2_DEFER", "1") == "1"
    for _ in range(args.warmup):
        step.run()
    step.profile_dominant = True  # one-shot event pair around the dense 96^3 gather of every step
    gather_ms = []
    barrier()
    step.defer_reads = defer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run()
        # the dense 96^3 level's gather kernel is the dominant kernel; the step's initialisation branch has already waited
        # for everything queued before it (the valid-voxel count it needs), so reading the event pair does not stall anything
        gather_ms.append(step.dominant_kernel_ms(lib))
    step.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    describe = step.describe()
    step.defer_reads = False
    step.profile_dominant = False
    lib.eprecon_profile_enable(0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: BASELINE.json configs[4] — the fragments of ONE scene dealt round-robin to the ranks, every fragment
    # preceded by the RCCL boundary-voxel exchange (eprecon_amd/distributed.py).  All ranks take part; reported
    # under `extra` (the headline stays the cfg2 metric on independent windows).
    cfg5 = None
    if use_dist and not args.no_extra:   # (EPRECON_BENCH_FORCE_DIST=1 exercises this leg in a single-rank group)
        cfg5 = bench_cfg5(torch.device("cuda", local_rank), rank, world, dist)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        gm = float(np.mean([g for g in gather_ms if g > 0])) if any(g > 0 for g in gather_ms) else None
        alg = step.dominant_kernel_bytes()
        traffic, traffic_src, l1 = None, None, None
        pmc = newest_profile("pmc_traffic_bp_gather.json")
        if pmc:
            # HBM-side bytes per launch from the newest committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
            # in separate runs, gfx950 FETCH_SIZE x2 correction applied).  PMC collection cannot run inside the timed
            # process, so this is a recorded measurement: it is reported only while it names the kernel timed here
            rec = json.load(open(pmc))
            if rec.get("kernel") == DOMINANT_KERNEL:
                traffic, traffic_src = rec["traffic_bytes"], os.path.relpath(pmc, ROOT)
                if rec.get("l1_accesses") and gm:
                    # what actually limits this gather: vector-L1 line accesses per launch (PMC,
                    # TCP_TOTAL_CACHE_ACCESSES) against 256 CUs x 1 line / clk x 2.4 GHz
                    rate = rec["l1_accesses"] / (gm * 1e-3)
                    peak, peak_src = l1_peak_from_probe()
                    l1 = {"line_accesses": rec["l1_accesses"], "achieved_glines_per_s": rate / 1e9,
                          "peak_glines_per_s": peak / 1e9, "frac": rate / peak, "peak_source": peak_src}
            else:
                traffic_src = f"{os.path.relpath(pmc, ROOT)} measured {rec.get('kernel')!r}, not this kernel: traffic withheld"
        roof = {"bound": "hbm", "kernel": DOMINANT_KERNEL + " (dense 96^3, C=24, 120x160)",
                "achieved": (alg / (gm * 1e-3) / 1e9) if gm else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (alg / (gm * 1e-3) / 1e9 / HBM_PEAK_GBS) if gm else None,
                "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": alg, "avg_launch_ms": gm,
                "limiter": "vector-L1 bandwidth, not HBM: see l1_bandwidth (bytes the taps request from L1) and l1 (line accesses, PMC)",
                "l1": l1, "l1_bandwidth": step.dominant_kernel_l1_bandwidth(gm)}
        out = {"metric": "fragments_per_sec", "value": value, "unit": "fragments/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": describe, "roofline": roof}
        if world == 1:
            out["roofline_conv"] = conv_roofline(step, lib)
            out["roofline_conv_family"] = conv_family_roofline()
            bfx = conv_family_roofline("conv_shapes_bf16x3.txt")      # the opt-in: beside the exact-fp32 record, priced the same way
            if bfx:
                bfx["opt_in"] = "EPRECON_CONV_BF16X3=1 (bf16 matrix pipe, three products, fp32 accumulate; priced against the fp32-MFMA peak like the exact path)"
                out["roofline_conv_family_bf16x3_opt_in"] = bfx
            if not args.no_extra:
                try:   # the extras must never cost the headline line (Cfg4Step.run raises on an early-returning fragment)
                    out["extra"] = extra_workloads(torch.device("cuda", local_rank), lib)
                except Exception as exc:  # noqa: BLE001
                    out["extra"] = {"error": f"{type(exc).__name__}: {exc}"}
                if cfg5 is not None:
                    out["extra"].update(cfg5)
                out.update(fragment_summary(out["extra"]))
        elif cfg5 is not None:
            out["extra"] = cfg5
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(step, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
