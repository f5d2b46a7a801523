"""-m gpu: EditGrid region growing on the device (csrc/editgrid.hip) vs the reference's Python run on CPU tensors
(tests/golden/editgrid.npz) and vs the oracle on further cases."""
import numpy as np
import pytest
import torch

from conftest import golden
from gpu_util import DEV, N, T
from test_oracle_golden_cpu import _editgrid_case

pytestmark = pytest.mark.gpu


def queue_array(eg):
    return np.array([list(c) + [l] for c, l in eg.growing_queue], np.int32).reshape(-1, 4)


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_editgrid_matches_reference_capture(tag):
    from laenerf_amd.editing import EditGrid
    g = golden("editgrid")
    dens, grid0, grid1 = _editgrid_case(g, tag)
    cascade = int(g[f"{tag}_cascade"])
    eg = EditGrid()
    eg.new_from_points(T(g[f"{tag}_pts"]), torch.zeros(cascade * 128 ** 3 // 8, dtype=torch.uint8, device=DEV), cascade, float(g[f"{tag}_bound"]))
    assert np.array_equal(N(eg.grid), grid0)
    assert np.array_equal(queue_array(eg), g[f"{tag}_queue0"])
    popped = eg.grow_region_queue(T(dens), 12.0, grow_iterations=int(g[f"{tag}_iters"]))
    assert popped == int(g[f"{tag}_iters"])
    assert np.array_equal(N(eg.grid), grid1)
    assert np.array_equal(queue_array(eg), g[f"{tag}_queue1"])


def test_editgrid_vs_oracle_budget_and_exhaustion(O):
    """call sequences the GUI produces: several grow calls in a row, tiny budgets (batches shorter than 32), growth until
    the queue runs dry; the selection never leaves the dense region"""
    from laenerf_amd.editing import EditGrid
    g = golden("editgrid")
    dens, grid0, _ = _editgrid_case(g, "c1")
    eg = EditGrid()
    eg.new_from_points(T(g["c1_pts"]), torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV), 1, 1.0)
    grid, queue = grid0, [tuple(r) for r in g["c1_queue0"]]
    for budget in (1, 5, 31, 32, 33, 700, 100000):
        popped = eg.grow_region_queue(T(dens), 12.0, grow_iterations=budget)
        grid, queue, ref_popped = O.grow_region_queue(grid, dens, 12.0, queue, grow_iterations=budget)
        assert popped == ref_popped
        assert np.array_equal(N(eg.grid), grid), budget
        assert np.array_equal(queue_array(eg), np.array(queue, np.int32).reshape(-1, 4)), budget
    assert eg.queue_length() == 0                                     # the last call drained the queue
    sel = np.nonzero(np.unpackbits(N(eg.grid), bitorder="little"))[0]
    assert sel.size > 1000 and (dens.reshape(-1)[sel[1:]] >= 12.0).sum() >= sel.size - 2      # only the seed may lie outside
    assert eg.grow_region_queue(T(dens), 12.0) == 0                   # empty queue: no-op like the reference
