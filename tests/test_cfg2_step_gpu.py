"""GPU: the scheduling options of the cfg2 workload (eprecon_amd.fragment_step.Cfg2Step) change WHEN the host reads a step's
counts and where the Back_Project levels are queued, never what a step computes."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _same(a, b):
    assert set(a) == set(b)
    assert torch.equal(a["stage0_coords"], b["stage0_coords"])
    for x, y in zip(a["init"], b["init"]):
        assert torch.equal(x, y)
    for lvl in ("bp24", "bp48", "bp96"):
        assert a[lvl]["n_valid"] == b[lvl]["n_valid"]
        assert torch.equal(a[lvl]["feats"], b[lvl]["feats"]) and torch.equal(a[lvl]["coords"], b[lvl]["coords"])


def test_deferred_reads_and_level_order_do_not_change_the_outputs():
    from eprecon_amd.fragment_step import Cfg2Step
    step = Cfg2Step(seed=0)
    ref = {k: v for k, v in step.run().items()}
    # the levels queued at the start of the step (round 2's order) instead of at the initialisation branch's wait
    step.levels_inside = not step.levels_inside
    _same(ref, step.run())
    step.levels_inside = not step.levels_inside
    # counts of step k read after step k + 1 is queued: run() returns the previous step's outputs, flush() the last one's
    step.defer_reads = True
    assert step.run() == {}
    second = step.run()
    _same(ref, second)
    _same(ref, step.flush())
    assert step.flush() is step.last            # nothing in flight any more
    step.defer_reads = False
    _same(ref, step.run())
