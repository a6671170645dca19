"""Every EPRECON_* switch that changes which code runs ("Switches") against the default, on a WHOLE cfg4 scene
(four sequential fragments with the persistent map): same finest voxel sets up to a handful of threshold flips, same TSDF
to fp32 round-off of a different summation order.  The library's selection switches are read per launch, the Python ones
are attributes set at construction from the environment: both are flipped in-process here."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from eprecon_amd.fragment_step import Cfg4Step
    step = Cfg4Step(seed=0, device=torch.device("cuda"), height=240, width=320)
    return step


def run_scene(step):
    step.k = 0
    out = []
    for _ in range(step.n_fragments):
        o = step.run()
        info = o.get("panoptic_info")
        out.append({"coords": o["coords"].clone(), "tsdf": o["tsdf"].clone(),
                    "segments": None if info is None else len(info[0]["panoptic_seg"][1])})
    torch.cuda.synchronize()
    return out


def assert_same_scene(a, b, exact=False):
    for fa, fb in zip(a, b):
        na, nb = fa["coords"].shape[0], fb["coords"].shape[0]
        if exact:
            assert torch.equal(fa["coords"], fb["coords"]) and torch.equal(fa["tsdf"], fb["tsdf"])
            continue
        assert abs(na - nb) <= max(8, na // 500), (na, nb)       # occupancy logits within round-off of the threshold may flip
        if na == nb and torch.equal(fa["coords"], fb["coords"]):
            assert torch.allclose(fa["tsdf"], fb["tsdf"], rtol=2e-3, atol=2e-3), float((fa["tsdf"] - fb["tsdf"]).abs().max())
        else:
            assert abs(float(fa["tsdf"].mean()) - float(fb["tsdf"].mean())) < 5e-3


@pytest.fixture(scope="module")
def default_run(scene):
    return run_scene(scene)


def test_default_is_deterministic(scene, default_run):
    assert_same_scene(default_run, run_scene(scene), exact=True)


@pytest.mark.parametrize("env", [{"EPRECON_CONV_DIRECT": "0"}, {"EPRECON_CONV_WIDEK": "0"}, {"EPRECON_CONV_SPLITK": "0"},
                                 {"EPRECON_CONV_DENSE3D": "0"}, {"EPRECON_CONV_DENSE3D": "1"},
                                 {"EPRECON_CONV_TAIL8": "0"}, {"EPRECON_CONV_PERSIST": "1"}, {"EPRECON_CONV_BF16X3": "1"}, {"EPRECON_CONV_INTERLEAVE": "0"}, {"EPRECON_CONV_SPLITK_FAST": "0"},
                                 {"EPRECON_CONV_DIRECT": "0", "EPRECON_CONV_WIDEK": "0", "EPRECON_CONV_SPLITK": "0",
                                  "EPRECON_CONV_DENSE3D": "0"}])
def test_convolution_selection_switches(scene, default_run, env, monkeypatch):
    """the library picks another kernel of the family for every layer the switched-off one used to take"""
    from eprecon_amd import _lib, sparse as SP
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = run_scene(scene)
    assert_same_scene(default_run, got)
    # and the switch really switches: a 3x3x3 48 -> 24 layer on a long list is the direct kernel's by default
    vs = SP.VoxelSet(torch.stack(torch.meshgrid(torch.zeros(1, dtype=torch.int32), torch.arange(40, dtype=torch.int32),
                                                torch.arange(40, dtype=torch.int32), torch.arange(40, dtype=torch.int32),
                                                indexing="ij"), -1).reshape(-1, 4).cuda(), 1)
    SP.sparse_conv(torch.randn(vs.n, 48, device="cuda"), torch.randn(27, 48, 24, device="cuda"), vs.kernel_map(3))
    if env.get("EPRECON_CONV_DIRECT") == "0":
        assert _lib.last_conv_kernel() != "spconv_direct16_kernel"


def test_conv_log_lists_every_launch(tmp_path):
    """EPRECON_CONV_LOG=<file> (read once per process): one line per convolution launch with the kernel that took it"""
    import subprocess
    import sys
    log = tmp_path / "conv.log"
    code = ("import torch; from eprecon_amd import sparse as SP;"
            "x = torch.randn(5000, 32, device='cuda'); w = torch.randn(32, 16, device='cuda');"
            "SP.sparse_conv(x, w); SP.sparse_conv(x, w); torch.cuda.synchronize()")
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, EPRECON_CONV_LOG=str(log)),
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lines = log.read_text().strip().splitlines()
    assert len(lines) == 2 and lines[0].split()[:4] == ["5000", "1", "32", "16"]


def test_gru_stage_call_off(scene, default_run, monkeypatch):
    """EPRECON_GRU_STAGE=0: the level's bookkeeping as separate calls with a host read each"""
    monkeypatch.setattr(scene.net.gru_fusion, "stage_call", False)
    assert_same_scene(default_run, run_scene(scene), exact=True)


def test_prefetch_off(scene, default_run, monkeypatch):
    """EPRECON_PREFETCH=0: every SPVCNN pass reads its own voxel-set sizes and the panoptic pruning its own counts (14 blocking
    reads per fragment) instead of riding on the previous compaction's read (10): bit-identical fragments, four reads apart"""
    import eprecon_amd.neucon_network as NN
    from eprecon_amd import _lib
    r0 = _lib.HOST_READS
    on = run_scene(scene)
    r1 = _lib.HOST_READS
    monkeypatch.setattr(NN, "_PREFETCH", False)
    off = run_scene(scene)
    r2 = _lib.HOST_READS
    assert_same_scene(default_run, on, exact=True)
    assert_same_scene(default_run, off, exact=True)
    n = scene.n_fragments
    # three SPVCNN size reads + the pruning's read per fragment (a level that falls back to the seeded random drop keeps its own)
    assert 3 * n <= (r2 - r1) - (r1 - r0) <= 4 * n, ((r1 - r0) / n, (r2 - r1) / n)


def test_spvcnn_issued_from_python(scene, default_run, monkeypatch):
    """EPRECON_SPVCNN_NATIVE=0: the body of every SPVCNN pass issued launch by launch from Python instead of by one library call
    (eprecon_spvcnn_forward_async): the same entry points with the same descriptors in the same order — bit-identical"""
    import eprecon_amd.modules as M
    monkeypatch.setattr(M, "_NATIVE_SPVCNN", False)
    assert_same_scene(default_run, run_scene(scene), exact=True)


def test_gru_one_stream(scene, default_run, monkeypatch):
    """EPRECON_GRU_STREAMS=0: the two ConvGRUs of a level on one stream"""
    monkeypatch.setattr(scene.net.gru_fusion, "two_streams", False)
    assert_same_scene(default_run, run_scene(scene), exact=True)


def test_fused_heads_off(scene, default_run, monkeypatch):
    """EPRECON_FUSED_HEADS=0: Linear4xTrans on separate Linear / LayerNorm launches"""
    import eprecon_amd.modules as M
    monkeypatch.setattr(M, "_FUSED_HEADS", False)
    assert_same_scene(default_run, run_scene(scene))


def test_decoder_on_pytorch_ops_and_without_graphs(scene, default_run, monkeypatch):
    """EPRECON_DECODER_FUSED=0 (module path with graph-replayed query side), then EPRECON_NO_GRAPH=1 on top"""
    dec = scene.net.panoptic
    monkeypatch.setattr(dec, "use_fused_voxel_side", False)
    a = run_scene(scene)
    monkeypatch.setattr(dec, "use_hip_graph", False)
    monkeypatch.setattr(scene.net.initialization, "use_hip_graph", False)
    b = run_scene(scene)
    assert_same_scene(default_run, a)
    assert_same_scene(default_run, b)
    assert [f["segments"] for f in a] == [f["segments"] for f in default_run]


def test_2d_stack_on_miopen(scene, default_run, monkeypatch):
    """EPRECON_MIOPEN_CONV2D=1: the 2D fusion stack of the occupancy initialisation on PyTorch-ROCm convolutions"""
    monkeypatch.setattr(scene.net.initialization, "use_hip_conv", False)
    assert_same_scene(default_run, run_scene(scene))


def test_convgru_with_fresh_indices_runs(scene, monkeypatch):
    """EPRECON_CONVGRU_LITERAL=0 is ANOTHER function (every SConv3d devoxelises with the indices of its own voxelisation,
    DESIGN.md 5b): it must run to the finest level; its values are checked against the oracle in test_modules / test_gru"""
    import eprecon_amd.torchsparse_utils as TU
    monkeypatch.setattr(TU, "LITERAL_CONVR", False)
    got = run_scene(scene)
    assert all(f["coords"].shape[0] > 0 and bool(torch.isfinite(f["tsdf"]).all()) for f in got)


def test_pipelined_modes(scene, default_run):
    """EPRECON_CFG4_PIPELINE=1 (bench.py): the pipelined serving mode, with and without the worker thread — same fragments"""
    try:
        for mode in (True, "inline"):
            scene.set_pipeline(mode)
            got = run_scene(scene)
            scene.flush()
            assert_same_scene(default_run, got, exact=True)
    finally:
        scene.set_pipeline(False)


def test_exchange_forced_at_one_rank(scene, default_run, monkeypatch):
    """EPRECON_FORCE_EXCHANGE=1 (+ EPRECON_XCHG_STREAM=0): the boundary exchange path (selection kernels on the map handles,
    RCCL collectives in a single-rank group, stamps) with nobody to exchange with — on its own stream and on the main one"""
    import socket
    import torch.distributed as dist
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        monkeypatch.setattr(scene.net, "distributed_exchange", True)
        assert_same_scene(default_run, run_scene(scene), exact=True)
        monkeypatch.setenv("EPRECON_XCHG_STREAM", "0")
        assert_same_scene(default_run, run_scene(scene), exact=True)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


# EPRECON_CFG2_DEFER: tests/test_cfg2_step_gpu.py (deferred reads and level order do not change the outputs).


def test_bench_line_through_a_single_rank_process_group():
    """EPRECON_BENCH_FORCE_DIST=1: bench.py's multi-rank legs (RCCL init, barrier, MAX all-reduce of the elapsed time, the
    exchange leg with per-rank rows / bytes) at world size 1"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EPRECON_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1",
                                   "--no-cpu-baseline", "--no-extra"], env=env, cwd=root, stderr=subprocess.DEVNULL)
    line = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])   # (RCCL prints its banner at exit)
    assert line["metric"] == "fragments_per_sec" and line["n_gpus"] == 1 and line["value"] > 0 and "roofline" in line
