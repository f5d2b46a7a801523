"""Region selection grid of the LAENeRF editor: the queue-based flood fill of the reference's `EditGrid`
(editing/editgrid.py:56-340: `new_from_points`, `grow_region_queue`, bit helpers) on the device.

The reference keeps a `collections.deque` of (coords tensor, level) pairs and pops 32 of them per Python iteration; here
the queue is a device array and `grow_region_queue` is ONE kernel that walks it in the reference's FIFO order
(csrc/editgrid.hip).  Bitfields use the layout of `density_bitfield` ([cascade * 128^3 / 8] uint8, Morton order).
"""
import torch

from .. import _lib, raymarching

GRIDSIZE = 128                      # EDIT_GRIDSIZE() (editgrid.py:14-15)
GRIDVOLUME = GRIDSIZE ** 3
_NEIGHBOURS = ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0))     # editgrid.py:118-125


def get_bitfield_at(cell_idx, level, bitfield):
    """editgrid.py:30-33: the selected bit (still in its bit position) of cells `cell_idx` at cascade `level`"""
    return bitfield[cell_idx // 8 + (GRIDVOLUME * level) // 8] & (1 << (cell_idx % 8)).to(torch.uint8)


def set_bits(bitfield, cell_idx, level):
    """set the bits of the given cells (every one of them: duplicates and byte-sharing cells included)"""
    byte = (cell_idx // 8 + (GRIDVOLUME * level) // 8).long()
    for b in range(8):
        sel = (cell_idx % 8) == b
        if bool(sel.any()):
            bitfield[byte[sel]] |= (1 << b)
    return bitfield


def pack_cells(coords, level):
    """queue entries x | y << 8 | z << 16 | level << 24 (int32 storage of the uint32 pattern)"""
    c = coords.to(torch.int64)
    lv = torch.as_tensor(level, device=coords.device).to(torch.int64).expand(c.shape[0])
    return (c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16) | (lv << 24)).to(torch.int32)


def unpack_cells(entries):
    e = entries.to(torch.int64) & 0xffffffff
    return torch.stack([e & 0xff, (e >> 8) & 0xff, (e >> 16) & 0xff], -1).to(torch.int32), (e >> 24).to(torch.int32)


class EditGrid:
    def __init__(self):
        self.reset()

    def reset(self):
        """editgrid.py:138-142"""
        self.grid = None
        self.pts = None
        self.palette = None
        self._queue = None               # device int32 [capacity] of packed cells
        self._state = None               # device int32 [4]: head, tail, popped by the last call, overflow

    def get_grid(self):
        return self.grid

    # ---- whole-grid helpers of the reference (editgrid.py:60-78, 145-164, 204-230, 343-369)
    def save_grid_as_torch(self, f):
        torch.save(self.grid.cpu(), f)

    def load_grid_as_torch(self, f):
        self.grid = torch.load(f).to("cuda")

    def xor(self, negative_grid):
        """remove `negative_grid` from the selection: grid & (grid ^ negative) (editgrid.py:66-68)"""
        self.grid = torch.bitwise_and(self.grid, torch.bitwise_xor(self.grid, negative_grid))

    def and_(self, negative_grid):
        """the reference's `and_` is a union (bitwise_or, editgrid.py:70-71); kept under its name"""
        self.grid = torch.bitwise_or(self.grid, negative_grid)

    def bw_and(self, other_grid):
        torch.bitwise_and(self.grid, other_grid, out=self.grid)

    @staticmethod
    def _block_coords(xs, ys, zs):
        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
        return torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)

    def add_neighbors(self, coords_hit, grid):
        """editgrid.py:204-230: select the in-bounds face neighbours of `coords_hit` (cascade level 0).  Every neighbour's
        bit is set; the reference's indexed byte assignment keeps one of several bits that share a byte in one call."""
        nb = torch.tensor(_NEIGHBOURS, dtype=torch.int32, device=coords_hit.device)
        c = (coords_hit[:, None, :].to(torch.int32) + nb[None]).reshape(-1, 3)
        c = c[((c >= 0) & (c < GRIDSIZE)).all(-1)]
        idx = raymarching.morton3D(c.contiguous()).long()
        set_bits(grid, idx % GRIDVOLUME, idx // GRIDVOLUME)

    def morphological(self):
        """editgrid.py:145-164: one dilation of the level-0 selection by the 6-neighbourhood, walked in the reference's
        64^3 blocks in its order (a block sees the bits earlier blocks have just added, as in the reference)"""
        dev = self.grid.device
        parts = torch.arange(GRIDSIZE, dtype=torch.int32, device=dev).split(64)
        for xs in parts:
            for ys in parts:
                for zs in parts:
                    coords = self._block_coords(xs, ys, zs)
                    idx = raymarching.morton3D(coords.contiguous()).long()
                    bits = get_bitfield_at(idx % GRIDVOLUME, idx // GRIDVOLUME, self.grid)
                    hit = bits.nonzero(as_tuple=True)[0]
                    if hit.numel():
                        self.add_neighbors(coords[hit], self.grid)

    def get_selection_points(self):
        """editgrid.py:343-369: centres of the selected level-0 cells in [0, 1]^3 (numpy [n, 3]); `pts` if it was set"""
        if self.pts is not None:
            return self.pts
        dev = self.grid.device
        out = []
        parts = torch.arange(GRIDSIZE, dtype=torch.int32, device=dev).split(32)
        for xs in parts:
            for ys in parts:
                for zs in parts:
                    coords = self._block_coords(xs, ys, zs)
                    idx = raymarching.morton3D(coords.contiguous()).long()
                    level, pos = idx // GRIDVOLUME, idx % GRIDVOLUME
                    hit = get_bitfield_at(pos, level, self.grid).nonzero(as_tuple=True)[0]
                    if hit.numel():
                        c = raymarching.morton3D_invert(pos[hit].to(torch.int32).contiguous()).float()
                        out.append(((c + 0.5) / GRIDSIZE - 0.5) * torch.pow(2.0, level[hit].float())[:, None] + 0.5)
        import numpy as np
        return np.concatenate([o.cpu().numpy() for o in out]) if out else np.zeros((0, 3), np.float32)

    # ---- the deque of the reference, as seen from Python
    @property
    def growing_queue(self):
        """remaining queue as a list of ((x, y, z), level) -- diagnostics / tests (one D2H copy)"""
        if self._queue is None:
            return []
        head, tail = (int(v) for v in self._state[:2].tolist())
        coords, lvl = unpack_cells(self._queue[head:tail])
        return [(tuple(c), int(l)) for c, l in zip(coords.tolist(), lvl.tolist())]

    def queue_length(self):
        if self._queue is None:
            return 0
        head, tail = (int(v) for v in self._state[:2].tolist())
        return tail - head

    def _push(self, entries, reserve):
        dev = entries.device
        n_old = self.queue_length()
        cap = n_old + entries.numel() + reserve
        q = torch.empty(cap, dtype=torch.int32, device=dev)
        if n_old:
            head, tail = (int(v) for v in self._state[:2].tolist())
            q[:n_old] = self._queue[head:tail]
        q[n_old:n_old + entries.numel()] = entries
        self._queue = q
        self._state = torch.tensor([0, n_old + entries.numel(), 0, 0], dtype=torch.int32, device=dev)

    def new_from_points(self, pts, density_bitfield, cascade, bound=1.0):
        """editgrid.py:80-136: start a selection from seed points `pts` [n,3]: their cells are selected and the face
        neighbours of every seed are queued at the seed's cascade level.  (The reference takes `trainer` and reads
        `trainer.model.density_bitfield` / `.cascade` from it.)"""
        pts = pts.to(density_bitfield.device).float().reshape(-1, 3)
        mx = pts.max(dim=-1).values                                                     # editgrid.py:23-26 (no abs, like the reference)
        level = torch.clamp(torch.frexp(mx).exponent, 0, cascade - 1)
        mip_bound = torch.minimum(torch.pow(2.0, level.float()), torch.tensor(float(bound), device=pts.device))
        grid_pos = torch.clamp(0.5 * (pts / mip_bound[:, None] + 1) * GRIDSIZE, 0.0, GRIDSIZE - 1).int()
        cell = raymarching.morton3D(grid_pos).long()
        self.grid = set_bits(torch.zeros_like(density_bitfield), cell % GRIDVOLUME, (GRIDVOLUME * level + cell) // GRIDVOLUME)
        self._queue, self._state = None, None
        nb = torch.tensor(_NEIGHBOURS, dtype=torch.int32, device=pts.device)
        coords = (grid_pos[:, None, :] + nb[None]).reshape(-1, 3)                        # point-major, offset order
        lv = level.to(torch.int32)[:, None].expand(-1, 6).reshape(-1)
        ok = ((coords >= 0) & (coords < GRIDSIZE)).all(-1)
        self._push(pack_cells(coords[ok], lv[ok]), 0)

    @torch.no_grad()
    def grow_region_queue(self, density_grid, density_thresh, occ_grid=None, grow_iterations=5000):
        """editgrid.py:274-340.  One kernel launch; returns the number of cells popped."""
        if self._queue is None or self.queue_length() == 0:
            print("Growing Queue is for some reason empty")                              # :279-281
            return 0
        _lib.need_cuda(self.grid, density_grid, self._queue)
        density_grid = density_grid.float().contiguous()
        C = density_grid.shape[0]
        # every popped cell pushes at most 6: give the queue room for the whole call
        head, tail = (int(v) for v in self._state[:2].tolist())
        self._push(torch.empty(0, dtype=torch.int32, device=self.grid.device), 6 * grow_iterations + 192)
        lib = _lib.load()
        _lib.check(lib.lae_grow_region(self.grid.data_ptr(), density_grid.data_ptr(), C, GRIDSIZE, float(density_thresh),
                                       self._queue.data_ptr(), self._queue.numel(), self._state.data_ptr(), int(grow_iterations), 32,
                                       _lib.stream()), "grow_region")
        popped, overflow = (int(v) for v in self._state[2:4].tolist())
        if overflow:
            raise RuntimeError("EditGrid.grow_region_queue: queue capacity exceeded")
        return popped
