"""CPU: the dense PyTorch blocks of eprecon_amd.modules reproduce the reference's blocks
(models/modules.py:273-399) given the reference's own state_dict (same parameter names)."""
import os

import numpy as np
import pytest
import torch

from eprecon_amd import modules as M

CASES = {"fusion8": lambda: M.Fusion_Block(8), "res6": lambda: M.Conv2d_Residual_Block(6, 3),
         "l4x_12_1": lambda: M.Linear4xTrans(12, 1), "l4x_12_12": lambda: M.Linear4xTrans(12, 12),
         "linres10": lambda: M.Linear_Residual(10)}


@pytest.mark.parametrize("name", list(CASES))
def test_block_matches_reference(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "dense_blocks.npz"))
    mod = CASES[name]()
    prefix = name + "__sd__"
    sd = {k[len(prefix):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(prefix)}
    missing = mod.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    mod.train()
    x = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(gold[name + "__shape"])).astype(np.float32))
    with torch.no_grad():
        y = mod(x).numpy()
    np.testing.assert_allclose(y, gold[name + "__out"], atol=1e-5, rtol=0)
