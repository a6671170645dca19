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


def test_batched_backbone_equals_per_view_calls():
    """MnasMulti.forward_views (all views of a fragment as one channels-last batch, BatchNorm
    statistics per view) against the reference's loop of per-view calls in train mode (models/neuralrecon.py:53-54,
    main.py:357); the stacked per-level maps alias one tensor (no copy before the back-projection)."""
    import torch
    from eprecon_amd.backbone import MnasMulti, stack_views
    torch.manual_seed(0)
    net = MnasMulti(1.0).train()
    for batch in (1, 2):
        imgs = [torch.randn(batch, 3, 64, 96) * 40 for _ in range(3)]
        with torch.no_grad():
            ref = [net(i) for i in imgs]
            got = net.forward_views(imgs)
        for lvl in range(3):
            scale = max(r[lvl].abs().max().item() for r in ref)
            err = max((r[lvl] - g[lvl]).abs().max().item() for r, g in zip(ref, got))
            assert err < 1e-4 * max(scale, 1.0), (batch, lvl, err, scale)
            st = stack_views([g[lvl] for g in got])
            assert st.shape == (3, batch) + tuple(ref[0][lvl].shape[1:])
            assert st.data_ptr() == got[0][lvl].data_ptr() and torch.equal(st, torch.stack([g[lvl] for g in got]))
    # unrelated tensors still stack (by copy)
    a, b = torch.randn(1, 4, 3, 3), torch.randn(1, 4, 3, 3)
    assert torch.equal(stack_views([a, b]), torch.stack([a, b]))
