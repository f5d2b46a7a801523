"""Contraction independence of the path's INTEGER outputs (SURVEY.md section 7 "Hard parts", raymarching.cu:345-398).

The reference is compiled by nvcc, which fuses `a*b+c`; which sites it fuses is a model here (the CUDA sources cannot
run in this image).  The oracle is therefore built twice -- FMA(a,b,c) = fmaf (the flavour every parity test uses) and
FMA(a,b,c) = a*b+c with two roundings (liblae_oracle_nofma.so) -- and this file measures what the choice can change:

  * Morton codes, bitfields, ray/box intervals: no FMA site at all -> identical by construction, asserted.
  * marcher: a sample position is one rounding (fused) or two (un-fused) away from o + t*d, so positions may differ by
    1 ulp, and a position that sits within that ulp of a voxel face can land in the neighbouring voxel.  The tests pin
    how often: on every march fixture (C in {1,2,4}, dt_gamma on/off, perturb on/off, training + inference + distill)
    per-ray sample COUNTS, offsets, `counter`, alive lists and `edit_occ` are asserted IDENTICAL between the flavours
    (12 training fixtures x 1024 rays, 3 inference traces x 512 rays), and every ray has positions within 1 ulp of
    the larger operand of o + t*d (4.8e-7 absolute) and bit-identical t-sequences (`deltas`) unless the start is
    jittered (`near + dt*noise` is itself an FMA site, :351/:746: the t-sequence then carries a last-bit difference).
"""
import numpy as np
import pytest

from laenerf_amd import synthetic as S


def _scene(O, C, bound, n_rays, seed):
    grid = S.sphere_density_grid(cascade=C, bound=bound)
    bits = S.pack_bits_np(grid, 10.0)
    o, d = S.lego_like_rays(n_rays, seed=seed, radius=3.2 if bound == 1 else 2.6)
    nears, fars = O.near_far_from_aabb(o, d, [-bound] * 3 + [bound] * 3, 0.2)
    return o, d, bits, nears, fars


def test_flavours_are_really_different_builds(O):
    with O.flavour("fma"):
        assert O.lib().orc_flavour_fma() == 1
    with O.flavour("nofma"):
        assert O.lib().orc_flavour_fma() == 0
    # and the default is restored
    assert O.lib().orc_flavour_fma() == 1


def test_integer_kernels_have_no_fma_site(O):
    rng = np.random.default_rng(0)
    c = rng.integers(0, 128, (20000, 3)).astype(np.int32)
    g = rng.uniform(-1, 30, 128 ** 3 // 8).astype(np.float32); g[::5] = 10.0
    o = rng.uniform(-3, 3, (4096, 3)).astype(np.float32)
    d = rng.standard_normal((4096, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:16, 0] = 0
    res = {}
    for fl in ("fma", "nofma"):
        with O.flavour(fl):
            m = O.morton3D(c)
            res[fl] = (m, O.morton3D_invert(m), O.packbits(g, 10.0), *O.near_far_from_aabb(o, d, [-1, -1, -1, 1, 1, 1], 0.2))
    for a, b in zip(res["fma"], res["nofma"]):
        assert np.array_equal(a, b)


MARCH_CASES = [(C, bound, dtg, perturb) for (C, bound) in ((1, 1.0), (2, 2.0), (4, 8.0))
               for dtg in (0.0, 1 / 128) for perturb in (False, True)]


@pytest.mark.parametrize("C,bound,dtg,perturb", MARCH_CASES)
def test_march_train_counts_offsets_independent_of_contraction(O, C, bound, dtg, perturb):
    n = 1024
    o, d, bits, nears, fars = _scene(O, C, bound, n, seed=10 + C)
    noises = np.random.default_rng(3).random(n).astype(np.float32) if perturb else np.zeros(n, np.float32)
    out = {}
    for fl in ("fma", "nofma"):
        with O.flavour(fl):
            out[fl] = O.march_rays_train(o, d, bound, bits, C, 128, nears, fars, noises, dt_gamma=dtg, max_steps=1024)
    xa, _, da, ra, ca = out["fma"]; xb, _, db, rb, cb = out["nofma"]
    assert np.array_equal(ra[:, 0], rb[:, 0]) and ca[1] == cb[1] == n
    same = ra[:, 2] == rb[:, 2]
    # a voxel decision could flip only where a position lies within 1 ulp of a voxel face: none on these fixtures
    assert same.all(), f"{(~same).sum()} of {n} rays change their sample count with the contraction model"
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb)                # offsets = scan of counts
    worst = 0.0
    for i in np.nonzero(same & (ra[:, 2] > 0))[0]:
        sa = slice(ra[i, 1], ra[i, 1] + ra[i, 2]); sb = slice(rb[i, 1], rb[i, 1] + rb[i, 2])
        if perturb:      # the jittered start near + dt*noise IS an FMA site (:351): every later t inherits its last bit
            assert np.abs(da[sa] - db[sb]).max() <= (4.8e-7 if dtg == 0 else 1e-5)
        else:
            assert np.array_equal(da[sa], db[sb])                            # t-sequence: no FMA site between samples
        worst = max(worst, float(np.abs(xa[sa] - xb[sb]).max()))
    # o + t*d with |o|, |t*d| < 8: the un-fused product is off by <= half an ulp of [4, 8) = 2.4e-7 before the sum rounds
    assert worst <= 4.8e-7, f"positions differ by {worst}"


@pytest.mark.parametrize("C,bound,dtg,perturb", [(1, 1.0, 0.0, False), (2, 2.0, 1 / 128, True), (4, 8.0, 0.0, True)])
def test_inference_loop_alive_lists_and_edit_marks_independent_of_contraction(O, C, bound, dtg, perturb):
    """the reference's inference loop (renderer.py:352-379 / 430-466) with constant sigma: alive lists, rays_t and edit_occ"""
    n = 512
    o, d, bits, nears, fars = _scene(O, C, bound, n, seed=20 + C)
    edit_bits = bits & np.random.default_rng(1).integers(0, 256, bits.shape[0]).astype(np.uint8)
    traces = {}
    for fl in ("fma", "nofma"):
        with O.flavour(fl):
            rng = np.random.default_rng(9)
            alive = np.arange(n, dtype=np.int32); rays_t = nears.copy()
            ws = np.zeros(n, np.float32); dep = np.zeros(n, np.float32); img = np.zeros((n, 3), np.float32)
            wse = np.zeros(n, np.float32); depe = np.zeros(n, np.float32)
            step, trace = 0, []
            while step < 1024 and len(alive):
                n_alive = len(alive); n_step = max(min(n // n_alive, 8), 1)
                noises = rng.random(n_alive).astype(np.float32) if (perturb and step == 0) else np.zeros(n_alive, np.float32)
                x, dd, dl, eo = O.march_rays(n_alive, n_step, alive, rays_t, o, d, bound, bits, C, 128, nears, fars, noises,
                                             align=128, dt_gamma=dtg, edit_bitfield=edit_bits)
                sig = np.full(x.shape[0], 4.0, np.float32); rgb = np.full((x.shape[0], 3), 0.5, np.float32)
                O.composite_rays(n_alive, n_step, alive, rays_t, sig, rgb, dl, ws, dep, img, 1e-4, wse, depe, eo)
                trace.append((alive.copy(), eo[:n_alive * n_step].copy(), (dl[:n_alive * n_step, 0] > 0).copy()))
                alive = alive[alive >= 0]
                step += n_step
            traces[fl] = (trace, rays_t.copy())
    ta, tb = traces["fma"], traces["nofma"]
    assert len(ta[0]) == len(tb[0])
    flips = 0
    for (aa, ea, va), (ab, eb, vb) in zip(ta[0], tb[0]):
        if not (np.array_equal(aa, ab) and np.array_equal(va, vb) and np.array_equal(ea, eb)):
            flips += 1
    assert flips == 0, f"{flips} of {len(ta[0])} iterations differ between the contraction models"
    if perturb:
        # jittered start: an FMA site (:746); with dt_gamma > 0 the recurrence t += t*gamma amplifies that last bit by
        # (1 + gamma)^steps
        assert np.abs(ta[1] - tb[1]).max() <= (4.8e-7 if dtg == 0 else 1e-5)
    else:
        assert np.array_equal(ta[1], tb[1])
