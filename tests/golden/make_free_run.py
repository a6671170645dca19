"""Regenerates tests/golden/free_run.npz: ONE fragment through the FREE-RUNNING CPU oracle (oracle/free_run.py) at cfg1
image size (9 views of 320x240, 96^3 fragment volume, empty scene map), SURVEY.md section 4 bullet 3 — and, with
`--size 640`, tests/golden/free_run_640.npz: the same at BASELINE's full image size (9 views of 640x480, cfg3), so that the
error accumulation across the stages of models/neucon_network.py:230-624 is pinned at the size the metric is quoted on.

    python tests/golden/make_free_run.py [--size 640]     (CPU only, a few minutes; does not touch /root/reference)

This fixture pins the HIP path against the oracle END TO END (no teacher forcing): tests/test_free_run_gpu.py loads the
calibrated occupancy heads stored here into a NeuConNet built from the same torch seed and compares its free-running
forward.  The oracle's sparse layers restate torchsparse / spconv semantics (parity unpinned, oracle/sparse.py), so this is
a HIP-vs-restatement pin, not a reference pin.  Stored: the three calibrated occupancy heads' last layers, per stage the
voxel lists, occupancy decisions, logits' margins (so the test can tell a legitimate near-zero flip from a real
difference) and sampled TSDF / logit values; the finest-level output."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from eprecon_amd import synthetic as S  # noqa: E402
from eprecon_amd.config import ModelCfg  # noqa: E402
from oracle import free_run as FR  # noqa: E402

SEED, WINDOW, FEAT_SEED = 7, dict(seed=0, width=320, height=240), 3
# (size key -> the windows of the batch; "320b2": TWO consecutive windows of one scene in one forward, round 6 — the reference
# trains at BATCH_SIZE 4, config/train.yaml:2, and loops `for b in range(bs)` in every stage)
WINDOWS = {320: [WINDOW], 640: [dict(seed=0, width=640, height=480)],
           "320b2": [dict(seed=0, width=320, height=240), dict(seed=1, width=320, height=240, advance=0.32)]}
FILES = {320: "free_run.npz", 640: "free_run_640.npz", "320b2": "free_run_b2.npz"}
KEEP = (0.45, 0.35, 0.25)


def build(size=320):
    """network and inputs exactly as tests/test_free_run_gpu.py rebuilds them"""
    from eprecon_amd.neucon_network import NeuConNet
    torch.manual_seed(SEED)
    net = NeuConNet(ModelCfg())
    net.train()
    windows = [S.make_window(**w) for w in WINDOWS[size]]
    feats, feats2, inputs = S.make_model_inputs(windows, feat_seed=FEAT_SEED)
    return net, feats, feats2, inputs


def main(size=320):
    net, feats, feats2, inputs = build(size)
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    init = net.initialization
    with torch.no_grad():   # the dense 2D fusion stack: the same PyTorch modules on the CPU (pinned to the reference's, dense_blocks.npz)
        bs = feats[0][0].shape[0]
        fused = torch.stack([init.feat_fusion_pre(torch.stack([torch.from_numpy(v[2][b]) for v in feats]),
                                                  torch.stack([torch.from_numpy(v[1][b]) for v in feats]),
                                                  torch.stack([torch.from_numpy(v[0][b]) for v in feats])) for b in range(bs)], 1).numpy()
    rec = FR.forward(sd, fused, feats2, inputs, keep_fraction=KEEP)
    assert "early" not in rec, rec.get("early")
    out = {"seed": np.array(SEED), "feat_seed": np.array(FEAT_SEED), "keep_fraction": np.array(KEEP),
           "init_n_valid": np.array(rec["init"]["n_valid"]), "init_n_selected": np.array(rec["init"]["n_selected"]),
           "init_min_margin": np.array(rec["init"]["sigmoid_margin"].min()),
           "coords": rec["coords"].astype(np.int32), "tsdf": rec["tsdf"].astype(np.float32)}
    for i, st in enumerate(rec["stages"]):
        out[f"head{i}_weight"] = sd[f"occ_preds.{i}.linear3.weight"]
        out[f"head{i}_bias"] = sd[f"occ_preds.{i}.linear3.bias"]
        out[f"s{i}_coords"] = st["coords"].astype(np.int32)
        out[f"s{i}_occ"] = st["occ"].astype(np.float32)
        out[f"s{i}_tsdf"] = st["tsdf"].astype(np.float32)
        out[f"s{i}_counts"] = np.array([st["n_in"], st["n_fused"], st["n_occ"]])
        print(f"stage {i}: in {st['n_in']} fused {st['n_fused']} occupied {st['n_occ']}  "
              f"|logit| < 1e-3: {(np.abs(st['occ']) < 1e-3).sum()}  min |logit| {np.abs(st['occ']).min():.2e}")
    out["size"] = np.array(WINDOWS[size][0]["width"])
    out["batch"] = np.array(len(WINDOWS[size]))
    path = os.path.join(HERE, FILES[size])
    np.savez_compressed(path, **out)
    print(f"free_run: {os.path.getsize(path) / 1024:.0f} KiB, finest voxels {len(rec['coords'])}")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="320", choices=[str(k) for k in WINDOWS])
    a = ap.parse_args().size
    main(int(a) if a.isdigit() else a)
