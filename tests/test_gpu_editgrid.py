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


def test_editgrid_whole_grid_helpers(tmp_path):
    """xor / and_ / bw_and / save / load / morphological / get_selection_points (editing/editgrid.py:60-78, 145-164, 343-369)
    against numpy restatements on the unpacked level-0 grid"""
    from laenerf_amd.editing import EditGrid
    from laenerf_amd import raymarching as rm
    rng = np.random.default_rng(4)
    eg = EditGrid()
    # a selection: two blobs, one of them across the 64-cell block boundary the reference's loops are cut at
    sel = np.zeros((128, 128, 128), bool)
    sel[60:68, 30:34, 62:66] = True; sel[10:13, 100:103, 5:8] = True; sel[0, 0, 0] = True; sel[127, 127, 127] = True
    c = np.argwhere(sel).astype(np.int32)
    idx = N(rm.morton3D(T(c))).astype(np.int64)
    bits = np.zeros(128 ** 3, np.uint8); bits[idx] = 1
    eg.grid = T(np.packbits(bits, bitorder="little"))
    other = T(rng.integers(0, 256, 128 ** 3 // 8).astype(np.uint8))
    g0 = N(eg.grid).copy()
    eg.xor(other);  assert np.array_equal(N(eg.grid), g0 & (g0 ^ N(other)))
    eg.grid = T(g0); eg.and_(other); assert np.array_equal(N(eg.grid), g0 | N(other))
    eg.grid = T(g0); eg.bw_and(other); assert np.array_equal(N(eg.grid), g0 & N(other))
    eg.grid = T(g0)
    f = str(tmp_path / "grid.pt")
    eg.save_grid_as_torch(f); eg.grid = None; eg.load_grid_as_torch(f)
    assert np.array_equal(N(eg.grid), g0) and eg.grid.is_cuda
    # selection points: cell centres in [0, 1]^3
    pts = eg.get_selection_points()
    assert pts.shape == (c.shape[0], 3)
    want = (c.astype(np.float32) + 0.5) / 128
    srt = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert np.allclose(srt(np.asarray(pts, np.float32)), srt(want))
    # morphological: one 6-neighbour dilation, walked in the reference's 64^3 blocks in order (a later block sees the cells
    # an earlier block has just added): numpy restatement of that exact procedure
    ref = sel.copy()
    for x0 in (0, 64):
        for y0 in (0, 64):
            for z0 in (0, 64):
                hit = np.argwhere(ref[x0:x0 + 64, y0:y0 + 64, z0:z0 + 64]) + np.array([x0, y0, z0])
                for d in ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)):
                    n = hit + np.array(d)
                    n = n[((n >= 0) & (n < 128)).all(1)]
                    ref[n[:, 0], n[:, 1], n[:, 2]] = True
    eg.morphological()
    got = np.unpackbits(N(eg.grid), bitorder="little").astype(bool)
    cr = np.argwhere(ref).astype(np.int32)
    want_bits = np.zeros(128 ** 3, bool); want_bits[N(rm.morton3D(T(cr))).astype(np.int64)] = True
    assert np.array_equal(got, want_bits) and want_bits.sum() > sel.sum()
