"""bench.py --workload cfg4 --scenes-per-gpu K: K independent scenes on ONE GPU, one process each (own
NeuConNet, map handles, streams), started together.  Every scene's fragments must come out bit-identical to the same scene
running alone on the device (sha1 over voxel list, TSDF, panoptic labels and segments of each fragment)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(k, base):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg4", "--scenes-per-gpu", str(k),
                        "--scene-seed-base", str(base), "--steps", "8", "--warmup", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_scenes_side_by_side_give_the_solo_outputs():
    both = _run(2, 0)
    assert both["config"]["scenes_per_gpu"] == 2 and both["n_gpus"] == 1 and both["value"] > 0
    assert both["value"] == pytest.approx(2 * 8 / both["span_s"])
    assert len(both["ms_per_fragment_by_scene"]) == 2 and set(both["digests_by_scene"]) == {"0", "1"}
    for seed in (0, 1):
        solo = _run(1, seed)
        assert solo["digests_by_scene"][str(seed)] == both["digests_by_scene"][str(seed)]
        assert len(solo["digests_by_scene"][str(seed)]) == 4
    assert both["digests_by_scene"]["0"] != both["digests_by_scene"]["1"]
