"""bench.py's record arithmetic and the committed evidence it quotes (CPU): the convolution roofline record from armed
launches, the figure of the cfg4-leading instance alone recomputed from the committed rocprofv3 rows, the defaults of the
driver's command line."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_conv_record_from_armed_launches():
    samples = [{"ms": 0.25, "rows": 1000, "pairs": 15000, "issued": 20000, "name": b"spconv_direct16_kernel"},
               {"ms": 0.35, "rows": 1000, "pairs": 15000, "issued": 20000, "name": b"spconv_direct16_kernel"}]
    r = bench._conv_roofline_record(samples, 27, 48, 24, "what")
    assert r["launches_timed"] == 2 and r["avg_launch_ms"] == pytest.approx(0.30)
    assert r["flops"] == 2.0 * 15000 * 48 * 24
    assert r["executed_flops"] == 2.0 * 20000 * 48 * 24
    assert r["executed_over_live"] == pytest.approx(20000 / 15000)
    assert r["output_stationary_flops"] == 2.0 * 1000 * 27 * 48 * 24
    assert r["achieved"] == pytest.approx(r["flops"] / 0.30e-3 / 1e12)
    assert r["frac"] == pytest.approx(r["achieved"] / bench.F32_MFMA_PEAK_TF)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and "1000 voxels" in r["kernel"]


def test_conv_record_without_an_executed_count_falls_back_to_every_offset():
    r = bench._conv_roofline_record([{"ms": 1.0, "rows": 10, "pairs": 100, "issued": 0, "name": b"k"}], 27, 8, 8, "w")
    assert r["executed_flops"] == 2.0 * 10 * 27 * 8 * 8


def test_instance_alone_recomputes_from_the_committed_rows():
    """`roofline_conv_cfg4.alone` is parsed from profiles/rNN/conv_cfg4_instance.txt: the fraction the file states must follow
    from its own rocprofv3 average and live-pair count (the round-3 figure could not be recomputed from profiles/)"""
    path = bench.newest_profile("conv_cfg4_instance.txt")
    assert path and int(re.search(r"profiles/r(\d+)/", path).group(1)) >= 4
    text = open(path).read()
    rows, pairs = map(int, re.search(r"rows (\d+) .* live pairs (\d+)", text).groups())
    launches, avg_us = re.search(r"rocprofv3 kernel trace: (\d+) launches .* avg ([0-9.]+) us", text).groups()
    stated = float(re.search(r"= ([0-9.]+) of the fp32-MFMA peak", text).group(1))
    hip_us = float(re.search(r"([0-9.]+) us per launch \(HIP events", text).group(1))
    flops = 2.0 * pairs * 48 * 24
    assert int(launches) >= 10 and rows > 100000
    assert flops / (hip_us * 1e-6) / 1e12 / bench.F32_MFMA_PEAK_TF == pytest.approx(stated, abs=2e-3)
    assert abs(float(avg_us) - hip_us) / hip_us < 0.03       # rocprofv3's average agrees with the HIP-event time
    assert pairs <= 27 * rows


def test_driver_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload) == (1, "cfg2") and a.steps >= 10 and a.warmup >= 1
    assert bench.HBM_PEAK_GBS == 8000.0 and 150 < bench.F32_MFMA_PEAK_TF < 160


def test_headline_roofline_recomputes_from_the_committed_profiles():
    """the bench line's `roofline` against profiles/rNN: algorithmic bytes / the rocprofv3 average of the same kernel, and the
    PMC traffic from the raw counters with the guide's gfx950 correction (FETCH_SIZE x2, KiB)"""
    import glob
    import json
    # the newest round that committed the driver's line, its rocprofv3 summary and its PMC record together
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "bench_cfg2_r*_final.json")))
    assert lines and int(re.search(r"profiles/r(\d+)/", lines[-1]).group(1)) >= 4
    rdir, tag = os.path.dirname(lines[-1]), re.search(r"bench_cfg2_(r\d+_final)\.json", lines[-1]).group(1)
    line = json.load(open(lines[-1]))
    roof = line["roofline"]
    assert line["metric"] and line["n_gpus"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"]) and roof["peak"] == bench.HBM_PEAK_GBS
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes"] / (roof["avg_launch_ms"] * 1e-3) / 1e9)
    # 16 N + 4 V C H W + n_valid (4 C + 16) at N = 96^3, V = 9, C = 24, 120 x 160, every voxel valid
    n = 96 ** 3
    assert roof["algorithmic_bytes"] == 16 * n + 4 * 9 * 24 * 120 * 160 + n * (4 * 24 + 16)
    summary = open(os.path.join(rdir, f"bench_cfg2_{tag}_rocprof_summary.txt")).read()
    avg_us = float(re.search(r"bp_gather_mlp_kernel<256, 0, 6, 1>.*?\| \d+ \| [0-9.]+ \| [0-9.]+ \| ([0-9.]+) \|", summary).group(1))
    assert abs(avg_us - roof["avg_launch_ms"] * 1e3) / avg_us < 0.05     # HIP events in bench.py vs rocprofv3's average
    pmc = json.load(open(os.path.join(rdir, "pmc_traffic_bp_gather.json")))
    fetch = float(re.search(r"bp_gather_mlp_kernel<256, 0, 6, 1> \| grid \d+ \| FETCH_SIZE \| ([0-9.]+)", summary).group(1))
    assert pmc["fetch_size_kib_raw"] == pytest.approx(fetch, rel=1e-4)
    assert pmc["traffic_bytes"] == pytest.approx(2 * pmc["fetch_size_kib_raw"] * 1024 + pmc["write_size_kib"] * 1024)
    assert os.path.exists(os.path.join(ROOT, pmc["source"]))
    # (the line quotes the newest PMC record committed BEFORE it ran: the previous round's when the round's own passes are collected
    # by the same script behind it — the two collections agree)
    assert roof["traffic"] == pytest.approx(pmc["traffic_bytes"], rel=0.02)
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]


def test_bare_multi_gpu_command_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` without a launcher (no WORLD_SIZE): the parent spawns the two ranks itself (main.py:67-76 of the
    reference launches one process per device the same way); on a box without a GPU every rank reaches the "needs a GPU" exit,
    and the parent returns their non-zero code instead of refusing the command line"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""       # (also on a GPU box: this test is about the launch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    for rank in (0, 1):
        assert f"needs a GPU (the HIP path has no CPU fallback) [rank {rank} of 2]" in r.stderr, r.stderr
    assert "launch with torch.distributed.run" not in r.stderr


def test_whole_fragment_figures_are_first_class():
    """the cfg4 figures (ms per fragment, launches, blocking reads, the leading convolution alone) sit at the top level of the line"""
    extra = {"cfg4_ms_per_fragment": 12.5, "cfg4_fragments_per_sec": 80.0, "launches_per_fragment": 800.0,
             "blocking_reads_per_fragment": 8.0, "e2e_ms_per_fragment": 18.0,
             "roofline_conv_cfg4": {"frac": 0.2, "alone": {"frac": 0.33}}, "cfg4_fragments_per_sec_k2": 150.0}
    s = bench.fragment_summary(extra)
    assert s == {"cfg4_ms_per_fragment": 12.5, "cfg4_fragments_per_sec": 80.0, "cfg4_launches_per_fragment": 800.0,
                 "cfg4_blocking_reads_per_fragment": 8.0, "cfg4_roofline_conv_alone_frac": 0.33,
                 "cfg4_fragments_per_sec_per_gpu_by_scenes": {"1": None, "2": 150.0, "4": None},
                 "cfg4_roofline_conv_in_situ_frac": 0.2, "e2e_ms_per_fragment": 18.0}
    assert bench.fragment_summary({"cfg34_error": "x"})["cfg4_ms_per_fragment"] is None


def test_conv_family_record_recomputes_from_the_committed_shape_list():
    """`roofline_conv_family`: sum of live flops / sum of time over the newest profiles/rNN/conv_shapes.txt"""
    r = bench.conv_family_roofline()
    assert r and r["shapes"] == 30 and r["bound"] == "mfma"
    assert r["frac"] == pytest.approx(r["sum_gflop_live"] * 1e9 / (r["sum_us"] * 1e-6) / 1e12 / bench.F32_MFMA_PEAK_TF, rel=1e-3)
    assert 0.05 < r["frac"] < 1.0 and r["slowest_shape"]["us"] <= r["sum_us"]


def test_scenes_per_gpu_is_a_cfg4_single_gpu_mode(monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--scenes-per-gpu", "2", "--workload", "train"])
    with pytest.raises(SystemExit, match="--scenes-per-gpu K goes with --workload cfg4"):
        bench.main()
