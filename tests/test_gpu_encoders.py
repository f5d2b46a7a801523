"""-m gpu: gridencoder / shencoder HIP kernels vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import golden

from gpu_util import DEV, N, T, bits_from_half, half_from_bits

pytestmark = pytest.mark.gpu


def grid_case(O, D=3, C=2, L=16, T_log2=19, desired=2048, base=16, B=3000, seed=0, gridtype=0, align=False, scale=1e-4):
    rng = np.random.default_rng(seed)
    offsets, pls = O.grid_offsets(input_dim=D, num_levels=L, level_dim=C, base_resolution=base, log2_hashmap_size=T_log2,
                                  desired_resolution=desired, align_corners=align)
    table = rng.uniform(-scale, scale, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0; x[1] = 1; x[2, 0] = 1.0000001; x[3, 1] = -1e-7; x[4] = 0.5          # edges, OOB, cell boundary
    return offsets, pls, table, x


@pytest.mark.parametrize("kw", [dict(), dict(L=4), dict(desired=4096), dict(D=2, L=4), dict(C=1, L=8, T_log2=14),
                                dict(C=4, L=8, T_log2=14, desired=512), dict(C=8, L=8, T_log2=12, desired=256),
                                dict(gridtype=1, T_log2=15, desired=1024), dict(align=True), dict(D=4, L=8, T_log2=14, desired=128),
                                dict(D=5, L=8, T_log2=14, desired=64, B=500)])
def test_grid_forward_fp32_bit_exact(O, kw):
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, **kw)
    D, C, L = x.shape[1], table.shape[1], offsets.shape[0] - 1
    gt, al = kw.get("gridtype", 0), kw.get("align", False)
    for interp in (0, 1):
        ref, ref_dd = O.grid_encode_forward(x, table, offsets, pls, kw.get("base", 16), calc_dy_dx=True, gridtype=gt,
                                            align_corners=al, interp=interp)
        out = torch.empty(L, x.shape[0], C, device=DEV); dd = torch.empty(x.shape[0], L * D * C, device=DEV)
        G.grid_encode_forward(T(x), T(table), T(offsets), out, x.shape[0], D, C, L, np.log2(pls), kw.get("base", 16), dd, gt, al, interp)
        assert np.array_equal(N(out), ref)
        assert np.allclose(N(dd), ref_dd, rtol=1e-5, atol=1e-7)
        out2 = torch.empty(x.shape[0], L * C, device=DEV)
        G.grid_encode_forward(T(x), T(table), T(offsets), out2, x.shape[0], D, C, L, np.log2(pls), kw.get("base", 16), None, gt, al, interp, blc=True)
        assert np.array_equal(N(out2).reshape(-1, L, C).transpose(1, 0, 2), ref)


@pytest.mark.parametrize("kw", [dict(scale=0.5), dict(L=8, T_log2=14, C=4, scale=0.5), dict(D=2, L=4, scale=0.5)])
def test_grid_forward_fp16_bit_exact(O, kw):
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, **kw)
    D, C, L = x.shape[1], table.shape[1], offsets.shape[0] - 1
    th = O.to_f16_bits(table)
    ref, ref_dd = O.grid_encode_forward(x, th, offsets, pls, 16, calc_dy_dx=True, f16=True)
    out = torch.empty(L, x.shape[0], C, device=DEV, dtype=torch.half)
    dd = torch.empty(x.shape[0], L * D * C, device=DEV, dtype=torch.half)
    G.grid_encode_forward(T(x), half_from_bits(th), T(offsets), out, x.shape[0], D, C, L, np.log2(pls), 16, dd, 0, False, 0)
    assert np.array_equal(bits_from_half(out), ref)
    assert np.allclose(N(dd), O.from_f16_bits(ref_dd), rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("kw", [dict(scale=0.5), dict(scale=0.5, B=70001), dict(scale=0.5, B=40000, T_log2=14), dict(scale=0.5, L=8, B=20000),
                                dict(scale=0.5, L=24, desired=8192, B=17000), dict(scale=0.5, B=30000, desired=4096, base=8),
                                dict(scale=0.5, B=257, L=3), dict(scale=0.5, B=66000, T_log2=21, L=12)])
def test_grid_forward_hot_configuration_all_modes(O, kw):
    """fp16 table, D=3, C=2, no dy_dx: the specialised kernel under the balanced schedule (mode 0), under the (l, l+8) map
    (mode 1) and the generic kernel (mode 2) -- each bit-identical to the oracle, in both output layouts, with and without
    the coordinate map, including out-of-range inputs and batches smaller than the schedule's granularity"""
    from laenerf_amd import _lib
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, **kw)
    L, B, base = offsets.shape[0] - 1, x.shape[0], kw.get("base", 16)
    th = O.to_f16_bits(table)
    ref, _ = O.grid_encode_forward(x, th, offsets, pls, base, f16=True)
    xm = (x * 2 - 1).astype(np.float32)                                   # the same points in [-1, 1]: in_map = (shift 1, scale 0.5)
    ref_m, _ = O.grid_encode_forward(((xm + np.float32(1.0)) * np.float32(0.5)).astype(np.float32), th, offsets, pls, base, f16=True)
    try:
        for mode in (0, 1, 2):
            assert _lib.load().lae_grid_set_forward_mode(mode) == 0
            out = torch.full((L, B, 2), 7.0, device=DEV, dtype=torch.half)
            G.grid_encode_forward(T(x), half_from_bits(th), T(offsets), out, B, 3, 2, L, np.log2(pls), base, None, 0, False, 0)
            assert np.array_equal(bits_from_half(out), ref), mode
            out2 = torch.full((B, L * 2), 7.0, device=DEV, dtype=torch.half)
            G.grid_encode_forward(T(x), half_from_bits(th), T(offsets), out2, B, 3, 2, L, np.log2(pls), base, None, 0, False, 0, blc=True)
            assert np.array_equal(bits_from_half(out2).reshape(B, L, 2).transpose(1, 0, 2), ref), mode
            out3 = torch.full((L, B, 2), 7.0, device=DEV, dtype=torch.half)
            G.grid_encode_forward(T(xm), half_from_bits(th), T(offsets), out3, B, 3, 2, L, np.log2(pls), base, None, 0, False, 0,
                                  in_map=(1.0, 0.5))
            assert np.array_equal(bits_from_half(out3), ref_m), mode
    finally:
        _lib.load().lae_grid_set_forward_mode(0)
    assert _lib.load().lae_grid_set_forward_mode(3) == -1


@pytest.mark.parametrize("kw", [dict(B=20000), dict(L=4, B=5000), dict(D=2, L=4), dict(C=4, L=8, T_log2=14, desired=512),
                                dict(C=1, L=8, T_log2=14), dict(gridtype=1, T_log2=15, desired=1024)])
def test_grid_backward_fp32(O, kw):
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, **kw)
    D, C, L, B = x.shape[1], table.shape[1], offsets.shape[0] - 1, x.shape[0]
    gt = kw.get("gridtype", 0)
    rng = np.random.default_rng(11)
    g = rng.standard_normal((L, B, C)).astype(np.float32)
    _, dd = O.grid_encode_forward(x, table, offsets, pls, 16, calc_dy_dx=True, gridtype=gt)
    ref, ref_gi = O.grid_encode_backward(g, x, table.shape, offsets, pls, 16, dy_dx=dd, gridtype=gt)
    ge = torch.zeros(table.shape, device=DEV); gi = torch.zeros(B, D, device=DEV)
    G.grid_encode_backward(T(g), T(x), T(table), T(offsets), ge, B, D, C, L, np.log2(pls), 16, T(dd), gi, gt, False, 0)
    # float atomics: order differs from the sequential oracle -> rounding-level tolerance
    assert np.allclose(N(ge), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    assert np.allclose(N(gi), ref_gi, rtol=1e-4, atol=1e-5 * (1 + np.abs(ref_gi).max()))
    ge2 = torch.zeros(table.shape, device=DEV)
    gb = np.ascontiguousarray(g.transpose(1, 0, 2).reshape(B, L * C))
    G.grid_encode_backward(T(gb), T(x), T(table), T(offsets), ge2, B, D, C, L, np.log2(pls), 16, None, None, gt, False, 0, blc=True)
    assert np.allclose(N(ge2), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("kw", [dict(B=20000), dict(L=8, T_log2=12, desired=512, B=6000), dict(gridtype=1, T_log2=15, desired=1024),
                                dict(L=8, T_log2=16, desired=32768, B=8000), dict(L=4, T_log2=22, desired=4096, B=6000),
                                dict(B=20001), dict(B=4098, L=4), dict(B=1027, L=4), dict(B=7, L=2), dict(B=5, L=2)])
def test_grid_backward_modes_agree(O, kw):
    """binned LDS pipeline vs generic global-atomic kernel vs oracle (fp32), incl. hashed levels smaller than a partition,
    (desired=32768) levels finer than a partition, whose x-corner pairs may straddle partitions (two single-corner items),
    and (T_log2=22) a level with more partitions than the binned path handles, which is left to the generic kernel"""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, **kw)
    D, C, L, B = 3, 2, offsets.shape[0] - 1, x.shape[0]
    gt = kw.get("gridtype", 0)
    g = np.random.default_rng(5).standard_normal((L, B, C)).astype(np.float32)
    ref, _ = O.grid_encode_backward(g, x, table.shape, offsets, pls, 16, gridtype=gt)
    outs = []
    try:
        for mode in (0, 1):
            G.set_backward_mode(mode)
            ge = torch.zeros(table.shape, device=DEV)
            G.grid_encode_backward(T(g), T(x), T(table), T(offsets), ge, B, D, C, L, np.log2(pls), 16, None, None, gt, False, 0)
            assert np.allclose(N(ge), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max()), mode
            outs.append(N(ge))
    finally:
        G.set_backward_mode(0)
    # "accumulated into": a second call adds on top of the existing gradient
    G.grid_encode_backward(T(g), T(x), T(table), T(offsets), ge, B, D, C, L, np.log2(pls), 16, None, None, gt, False, 0)


@pytest.mark.parametrize("B,gscale", [(30000, 1e-1), (30001, 1e-1), (29999, 1e-1), (1022, 1e-1), (30000, 60.0), (20000, 2e-6), (400001, 1e-1)])
def test_grid_backward_fp16_exact_sum(O, B, gscale):
    """fp16 mode of the binned pipeline = correctly rounded exact sum of the fp16-rounded contributions (int64 fixed point):
    compare with a float64 accumulation of the same rounded contributions; also deterministic across runs.  Batch sizes that
    are not a multiple of the 4 samples a lane reads at once: the last lane's 1-3 samples must not be lost.  From 400 k samples on
    the accumulate pass deals 64 instead of 32 workgroups per level (more sub-ranges per partition, same exact sums)."""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=B)
    L, C = 16, 2
    # gscale 60: many contributions at or above 128 in magnitude (the accumulate pass converts those on its integer path,
    # smaller ones through the float pipeline); 2e-6: fp16 subnormals
    g = (np.random.default_rng(2).standard_normal((L, B, C)) * gscale).astype(np.float32)
    gh = O.to_f16_bits(g)
    ge = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(half_from_bits(gh), T(x), T(table).half(), T(offsets), ge, B, 3, C, L, np.log2(pls), 16, None, None, 0, False, 0)
    ge2 = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(half_from_bits(gh), T(x), T(table).half(), T(offsets), ge2, B, 3, C, L, np.log2(pls), 16, None, None, 0, False, 0)
    ref32, _ = O.grid_encode_backward(O.from_f16_bits(gh), x, table.shape, offsets, pls, 16)
    assert torch.equal(ge, ge2)                    # every level: exact sums, rounded once, independent of any order
    err = np.abs(N(ge) - ref32)
    assert err.max() < 2e-3 * np.abs(ref32).max() + 1e-6        # one fp16 rounding of the sum (+ per-contribution rounding)


@pytest.mark.parametrize("B", [50000, 4099, 400003])
def test_grid_backward_fp16_bit_exact_on_lattice_points(O, B):
    """samples ON the vertices of a level whose scale is a power of two (resolution 17 -> scale 16, x = (k + 0.5) / 16):
    the interpolation weights are exactly 1 and 0, so every queue item is one of the fp16 gradients itself and the table
    entry must be RN_half(exact sum) BIT for bit.  The gradients mix subnormals, ordinary values, values at or above 128
    (the accumulate pass's integer path) and zeros; the expectation is computed in integers (value * 2^24)."""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, _, table, _ = grid_case(O, L=2, base=17, desired=34, B=8)
    offsets, table, pls = offsets[:2].copy(), table[:offsets[1]], 1.0      # the first level alone: resolution 17, scale 16
    rng = np.random.default_rng(7)
    k = rng.integers(0, 16, (B, 3))
    # neighbouring samples in different cells: a lane of the fill pass merges the (up to 4) consecutive samples it walks when
    # they sit in one cell -- their fp32 sum becomes ONE fp16 item (test_..._same_cell_neighbours_merge_in_fp32 below)
    while True:
        same = np.zeros(B, bool)
        for lag in (1, 2, 3): same[lag:] |= (k[lag:] == k[:-lag]).all(axis=1)
        if not same.any(): break
        k[same, 0] = (k[same, 0] + rng.integers(1, 16, int(same.sum()))) % 16
    x = ((k + 0.5) / 16.0).astype(np.float32)
    assert np.array_equal(x.astype(np.float64) * 16 + 0.5, k + 1.0)
    kinds = rng.integers(0, 4, (1, B, 2))
    mag = np.where(kinds == 0, 3e-6, np.where(kinds == 1, 0.05, np.where(kinds == 2, 700.0, 0.0)))
    gh = (rng.standard_normal((1, B, 2)) * mag).astype(np.float16)
    ge = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(T(gh), T(x), T(table).half(), T(offsets), ge, B, 3, 2, 1, np.log2(pls), 17, None, None, 0, False, 0)
    stride = 18                                                      # (resolution + 1) vertices per axis, x fastest (gridencoder.cu:66-84)
    idx = (k[:, 0] + 1) + (k[:, 1] + 1) * stride + (k[:, 2] + 1) * stride * stride
    fix = np.round(gh[0].astype(np.float64) * 2.0 ** 24).astype(np.int64)
    assert np.array_equal(fix / 2.0 ** 24, gh[0].astype(np.float64))   # every fp16 value is a multiple of 2^-24
    acc = np.zeros((table.shape[0], 2), dtype=np.int64)
    np.add.at(acc, idx, fix)
    want = (acc.astype(np.float64) / 2.0 ** 24).astype(np.float16)    # one rounding, ties to even
    got_bits, want_bits = bits_from_half(ge), want.view(np.uint16)
    zero = ((got_bits & 0x7fff) == 0) & ((want_bits & 0x7fff) == 0)               # +0 and -0 are the same gradient
    assert np.array_equal(np.where(zero, 0, got_bits), np.where(zero, 0, want_bits))
    assert (np.abs(want.astype(np.float64)) >= 128).any() and (np.abs(fix) < 1024).any()    # both conversion paths and subnormals were in play


def test_grid_backward_same_cell_neighbours_merge_in_fp32(O):
    """what a queue item is when consecutive samples share a cell: the lane that walks samples 4i .. 4i+3 adds the w*g of a run
    of same-cell samples in fp32 and rounds that sum to fp16 ONCE (fewer roundings than one per sample, and far fewer than the
    reference's one per atomicAdd(half2), gridencoder.cu:248-340).  Lattice points again (weights exactly 1 and 0): groups of
    four samples in one cell with gradients whose fp32 sum is not an fp16 number."""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, _, table, _ = grid_case(O, L=2, base=17, desired=34, B=8)
    offsets, table = offsets[:2].copy(), table[:offsets[1]]
    B = 4 * 3000
    rng = np.random.default_rng(11)
    kc = rng.permutation(4096)[:B // 4]                                  # every group of four its own cell
    k = np.repeat(np.stack([kc % 16, (kc // 16) % 16, kc // 256], axis=1), 4, axis=0)
    x = ((k + 0.5) / 16.0).astype(np.float32)
    gh = (rng.standard_normal((1, B, 2)) * np.array([300.0, 0.07, 5.0, 0.07])[None, np.arange(B) % 4, None]).astype(np.float16)
    ge = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(T(gh), T(x), T(table).half(), T(offsets), ge, B, 3, 2, 1, np.log2(1.0), 17, None, None, 0, False, 0)
    idx = (k[::4, 0] + 1) + (k[::4, 1] + 1) * 18 + (k[::4, 2] + 1) * 324
    g32 = gh[0].astype(np.float32).reshape(-1, 4, 2)
    run = ((g32[:, 0] + g32[:, 1]) + g32[:, 2]) + g32[:, 3]              # the lane's order, fp32
    want = np.zeros((table.shape[0], 2), np.float16); want[idx] = run.astype(np.float16)
    exact = np.zeros((table.shape[0], 2), np.float16); exact[idx] = gh[0].astype(np.float64).reshape(-1, 4, 2).sum(axis=1).astype(np.float16)
    got = N(ge)
    assert np.array_equal(got, want)
    assert np.abs(got.astype(np.float64) - exact.astype(np.float64)).max() <= 1.0       # at most one fp16 ulp (sums stay below 2048) from RN(exact sum)


def test_grid_backward_reports_stored_nonfinite_values(O):
    """nonfinite_flag of the backward (include/laenerf.h): set when a non-finite gradient is STORED -- an inf contribution, a
    sum that overflows fp16, a non-finite value already in the buffer that the call adds to -- and left alone otherwise;
    refused (LAE_EINVAL) when a level is too large for the binned pipeline, which is the only one that can tell"""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=20000)
    B, L, C = 20000, 16, 2
    oh = np.ascontiguousarray(offsets.astype(np.int32))
    g = (np.random.default_rng(2).standard_normal((L, B, C)) * 1e-1).astype(np.float16)

    def run(grad, ge=None, **kw):
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        ge = torch.zeros(table.shape, device=DEV, dtype=torch.half) if ge is None else ge
        G.grid_encode_backward(T(grad), T(x), T(table).half(), T(offsets), ge, B, 3, C, L, np.log2(pls), 16, None, None, 0, False, 0,
                               offsets_host=oh, nonfinite_flag=flag.data_ptr(), **kw)
        return int(flag.item()), ge

    f, ge = run(g)
    assert f == 0 and torch.isfinite(ge.float()).all()
    bad = g.copy(); bad[11, 777, 1] = np.inf
    f, ge = run(bad)
    assert f == 1 and not torch.isfinite(ge.float()).all()
    big = g.copy(); big[0] = 60000.0                                   # level 0: dozens of samples per entry -> the sum overflows
    f, ge = run(big)
    assert f == 1 and torch.isinf(ge.float()).any()
    f, _ = run(g, ge=ge)                                               # adds on top of the overflowed buffer: still there -> reported
    assert f == 1
    plan = G.grid_backward_plan(T(x), T(offsets), B, 3, C, L, np.log2(pls), 16, 0, False, 0, True)
    f, _ = run(bad, plan=plan)
    assert f == 1
    # a level with more than 2^21 entries goes through the generic atomic kernel: the flag cannot be promised
    offsets2, pls2, table2, x2 = grid_case(O, L=4, T_log2=22, desired=4096, B=6000)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        G.grid_encode_backward(T(np.zeros((4, 6000, 2), np.float16)), T(x2), T(table2).half(), T(offsets2),
                               torch.zeros(table2.shape, device=DEV, dtype=torch.half), 6000, 3, 2, 4, np.log2(pls2), 16, None, None, 0,
                               False, 0, offsets_host=np.ascontiguousarray(offsets2.astype(np.int32)), nonfinite_flag=flag.data_ptr())


def test_grid_backward_fp16(O):
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=4000, L=8, T_log2=14, desired=512)
    B, L, C = 4000, 8, 2
    g = (np.random.default_rng(2).standard_normal((L, B, C)) * 1e-2).astype(np.float32)
    gh = O.to_f16_bits(g)
    ref32, _ = O.grid_encode_backward(O.from_f16_bits(gh), x, table.shape, offsets, pls, 16)
    ge = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(half_from_bits(gh), T(x), T(table).half(), T(offsets), ge, B, 3, C, L, np.log2(pls), 16, None, None, 0, False, 0)
    # fp16 atomics round every partial sum: compare with the fp32 sum at fp16 resolution of the largest bins
    assert np.abs(N(ge) - ref32).max() < 2e-2 * np.abs(ref32).max()


def test_grid_autograd_module(O):
    """GridEncoder module: forward [B, L*C], backward -> embeddings.grad; fp32 and autocast(fp16)"""
    from laenerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(num_levels=8, log2_hashmap_size=14, desired_resolution=512).to(DEV)
    rng = np.random.default_rng(3)
    enc.embeddings.data = T(rng.uniform(-0.5, 0.5, enc.embeddings.shape).astype(np.float32))
    x = rng.uniform(-1, 1, (1000, 3)).astype(np.float32)
    y = enc(T(x), bound=1)
    ref, _ = O.grid_encode_forward((x + 1) / 2, N(enc.embeddings), N(enc.offsets), enc.per_level_scale, 16, out_blc=True)
    assert y.shape == (1000, 16) and np.allclose(N(y), ref, atol=1e-6)
    g = rng.standard_normal((1000, 16)).astype(np.float32)
    y.backward(T(g))
    refg, _ = O.grid_encode_backward(g, (x + 1) / 2, tuple(enc.embeddings.shape), N(enc.offsets), enc.per_level_scale, 16, grad_blc=True)
    assert np.allclose(N(enc.embeddings.grad), refg, rtol=1e-4, atol=1e-4 * np.abs(refg).max())
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = enc(T(x), bound=1)
    assert y16.dtype == torch.float16 and np.abs(N(y16) - ref).max() < 3e-3
    # input gradients (calc_grad_inputs) through the module
    xi = T(x).requires_grad_()
    enc(xi, bound=1).backward(T(g))
    _, dd = O.grid_encode_forward((x + 1) / 2, N(enc.embeddings), N(enc.offsets), enc.per_level_scale, 16, calc_dy_dx=True, out_blc=True)
    _, gi = O.grid_encode_backward(g, (x + 1) / 2, tuple(enc.embeddings.shape), N(enc.offsets), enc.per_level_scale, 16, dy_dx=dd, grad_blc=True)
    assert np.allclose(N(xi.grad), gi / 2, rtol=1e-3, atol=1e-3 * np.abs(gi).max())


def test_grid_total_variation(O):
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, L=8, T_log2=14, desired=512, B=2000, scale=0.5)
    grad = np.zeros_like(table)
    O.grad_total_variation(x, table, grad, offsets, 1e-3, pls, 16)
    g = torch.zeros(table.shape, device=DEV)
    G.grad_total_variation(T(x), T(table), g, T(offsets), 1e-3, 2000, 3, 2, 8, np.log2(pls), 16, 0, False)
    assert np.allclose(N(g), grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("degree", range(1, 9))
def test_sh_all_degrees(O, degree):
    from laenerf_amd.shencoder import sh_encode
    rng = np.random.default_rng(degree)
    for B in (1, 255, 1000):
        d = rng.standard_normal((B, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[0] = [0, 0, 1]
        ref, ref_dd = O.sh_encode_forward(d, degree, True)
        out = sh_encode(T(d), degree, False)
        assert out.shape == (B, degree * degree)
        assert np.allclose(N(out), ref, rtol=2e-5, atol=2e-6)
        di = T(d).requires_grad_()
        y = sh_encode(di, degree, True)
        g = rng.standard_normal(ref.shape).astype(np.float32)
        y.backward(T(g))
        assert np.allclose(N(di.grad), O.sh_encode_backward(g, ref_dd, degree), rtol=1e-4, atol=1e-4 * (1 + np.abs(ref_dd).max()))


def test_freq_encoder_matches_the_references_own_torch_encoder():
    """K18 pinned by execution of the reference: its pure-torch FreqEncoder (encoding.py:5-43, run by make_golden.py::gen_freq)
    has the row layout of the CUDA encoder (inputs, then per frequency sines | cosines)"""
    from laenerf_amd.freqencoder import freq_encode
    g = golden("freq_encoder")
    for D, deg in ((3, 4), (3, 10), (2, 6), (5, 1)):
        x, ref = g[f"x_{D}_{deg}"], g[f"y_{D}_{deg}"]
        y = freq_encode(T(x), deg, D + 2 * D * deg)
        assert y.shape == ref.shape
        assert np.abs(N(y) - ref).max() < 2e-6 * 2 ** deg + 2e-6              # fast-sine error grows with the argument, as above


@pytest.mark.parametrize("D,deg", [(3, 6), (3, 10), (2, 4), (5, 1)])
def test_freq_encoder(O, D, deg):
    """K18/K19 (freqencoder.cu:30-94) through the C ABI vs the oracle; the kernels use the fast sine like the reference"""
    from laenerf_amd.encoding import get_encoder
    from laenerf_amd.freqencoder import FreqEncoder, freq_encode
    rng = np.random.default_rng(D * 100 + deg)
    for B in (1, 255, 4097):
        x = rng.uniform(-1, 1, (B, D)).astype(np.float32)
        ref = O.freq_encode_forward(x, deg)
        xi = T(x).requires_grad_()
        y = freq_encode(xi, deg, D + 2 * D * deg)
        assert y.shape == ref.shape
        assert np.array_equal(N(y)[:, :D], x)
        assert np.abs(N(y) - ref).max() < 2e-6 * 2 ** deg + 2e-6                # fast-sine error grows with the argument
        g = rng.standard_normal(ref.shape).astype(np.float32)
        y.backward(T(g))
        rg = O.freq_encode_backward(g, N(y), D, deg)                             # same saved outputs -> same arithmetic
        assert np.allclose(N(xi.grad), rg, rtol=1e-5, atol=1e-5 * (1 + np.abs(rg).max()))
    enc, dim = get_encoder("frequency", input_dim=D, multires=deg)
    assert isinstance(enc, FreqEncoder) and dim == D + 2 * D * deg
    with torch.autocast("cuda", dtype=torch.float16):                           # forced to fp32 like the reference (freq.py:17)
        out = enc(T(x).half().reshape(-1, 1, D))
    assert out.dtype == torch.float32 and out.shape == (x.shape[0], 1, dim)
    with pytest.raises(RuntimeError):
        freq_encode(torch.zeros(4, D), deg, D + 2 * D * deg)                     # CPU tensor: no fallback


def test_grid_backward_accumulates_exactly_on_top(O):
    """`grad_embeddings` is accumulated into (gridencoder.cu:473-503): a second execution adds the same exact
    per-partition sums on top of the first, entry = RN(old + sum) with ONE rounding, so twice == 2 * once bit for bit"""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=30000)
    B, L, C = x.shape[0], offsets.shape[0] - 1, 2
    g = (np.random.default_rng(7).standard_normal((L, B, C)) * 1e-2).astype(np.float32)
    gh = half_from_bits(O.to_f16_bits(g))
    S_ = np.log2(pls)
    once = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(gh, T(x), None, T(offsets), once, B, 3, C, L, S_, 16, None, None, 0, False, 0)
    assert float(once.float().abs().sum()) > 0
    for rep in range(4):             # repeated: the sub-ranges of the small levels hand their partial sums over between workgroups
        twice = torch.zeros(table.shape, device=DEV, dtype=torch.half)
        for _ in range(2):
            G.grid_encode_backward(gh, T(x), None, T(offsets), twice, B, 3, C, L, S_, 16, None, None, 0, False, 0, offsets_host=offsets)
        assert torch.equal(twice, (once.float() * 2).half()), rep           # every level, dense ones included


def test_grid_backward_plan_can_be_executed_twice(O):
    """a plan (counting pass + scans) serves any number of fill / accumulate executions: every execution resets the work
    queue / arrival counters it uses, so a second execution on the same plan adds the same gradient again"""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=30000)
    B, L, C = x.shape[0], offsets.shape[0] - 1, 2
    g = (np.random.default_rng(7).standard_normal((L, B, C)) * 1e-2).astype(np.float32)
    gh = half_from_bits(O.to_f16_bits(g))
    S_ = np.log2(pls)
    plan = G.grid_backward_plan(T(x), T(offsets), B, 3, 2, L, S_, 16, 0, False, 0, True)
    ref = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(gh, T(x), None, T(offsets), ref, B, 3, C, L, S_, 16, None, None, 0, False, 0)
    once = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(gh, T(x), None, T(offsets), once, B, 3, C, L, S_, 16, None, None, 0, False, 0, plan=plan)
    assert torch.equal(once, ref)                                   # planned == unplanned, every level
    twice = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    for _ in range(2):
        G.grid_encode_backward(gh, T(x), None, T(offsets), twice, B, 3, C, L, S_, 16, None, None, 0, False, 0, plan=plan)
    assert torch.equal(twice, (once.float() * 2).half())


def test_grid_backward_host_offsets_never_change_a_result(O):
    """offsets_host (the module's host copy of the level sizes, include/laenerf.h) only steers launches: with and without
    it the gradients are identical -- also for a table whose last level (2^22 entries) exceeds what the binned path
    handles and goes to the generic atomic kernel, and for two same-L encoders of different sizes used back to back
    (round 1 cached level sizes keyed on the device pointer of `offsets`; nothing is cached now)"""
    from laenerf_amd.backend import gridencoder_backend as G
    rng = np.random.default_rng(3)
    B = 6000
    x = rng.random((B, 3)).astype(np.float32)
    outs = {}
    for T_log2 in (22, 12, 22):                                   # big, small, big again: same L, different level sizes
        offsets, pls = O.grid_offsets(num_levels=4, log2_hashmap_size=T_log2, desired_resolution=4096)
        g = (rng.standard_normal((4, B, 2)) * 1e-2).astype(np.float32)
        gh = half_from_bits(O.to_f16_bits(g))
        toff = T(offsets)                                          # a fresh device tensor each round (addresses get reused)
        res = []
        for host in (None, offsets):
            ge = torch.zeros(int(offsets[-1]), 2, device=DEV, dtype=torch.half)
            G.grid_encode_backward(gh, T(x), None, toff, ge, B, 3, 2, 4, np.log2(pls), 16, None, None, 0, False, 0, offsets_host=host)
            res.append(ge)
        ref32, _ = O.grid_encode_backward(O.from_f16_bits(O.to_f16_bits(g)), x, (int(offsets[-1]), 2), offsets, pls, 16)
        lo = int(offsets[3])
        assert float(res[0][lo:].float().abs().sum()) > 0          # the last level got its gradient (binned or generic)
        assert np.abs(N(res[0]) - ref32).max() < 2e-2 * np.abs(ref32).max()
        for lvl in range(4):
            a, b = int(offsets[lvl]), int(offsets[lvl + 1])
            parts = -(-(b - a) // 4096)
            if parts <= 512:             # exact sums, one rounding, sub-buckets merged in a fixed order: bit-identical
                assert torch.equal(res[0][a:b], res[1][a:b]), lvl
            else:                        # generic kernel: fp16 atomics, rounding order
                assert (res[0][a:b].float() - res[1][a:b].float()).abs().max().item() <= 2e-2 * float(np.abs(ref32).max())
        del toff


def test_grid_backward_nonfinite_gradient_stays_visible(O):
    """an Inf / NaN in the incoming gradient (overflowed loss scale) must reach the table gradient as a non-finite value
    (the reference's half atomicAdd keeps it; GradScaler relies on finding it): the integer accumulators of the binned
    path mark such entries and write NaN"""
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=5000)
    B, L, C = x.shape[0], offsets.shape[0] - 1, 2
    g = (np.random.default_rng(9).standard_normal((L, B, C)) * 1e-2).astype(np.float16)
    g[3, 100, 0] = np.inf; g[9, 200, 1] = -np.inf; g[14, 300, 0] = np.nan
    ge = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(torch.from_numpy(g).to(DEV), T(x), None, T(offsets), ge, B, 3, C, L, np.log2(pls), 16, None, None, 0, False, 0)
    bad = ~torch.isfinite(ge.float())
    for lvl in (3, 9, 14):
        lo, hi = int(offsets[lvl]), int(offsets[lvl + 1])
        assert bad[lo:hi].any(), f"level {lvl}: the non-finite contribution vanished"
    for lvl in (0, 5, 12, 15):
        lo, hi = int(offsets[lvl]), int(offsets[lvl + 1])
        assert not bad[lo:hi].any()


def test_fused_adam_backward_helper_equals_loss_backward():
    """FusedAdam.backward(loss) = loss.backward() without autograd's ones_like fill: same gradients"""
    from laenerf_amd.network import NeRFNetwork
    from laenerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = NeRFNetwork(bound=1, log2_hashmap_size=14).to(DEV)
    net.encoder.embeddings.data.uniform_(-0.3, 0.3)
    opt = FusedAdam(net, param_groups=net.get_params(1e-2))
    x = torch.rand(4096, 3, device=DEV) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(4096, 3, device=DEV), dim=-1)
    grads = []
    for helper in (False, True):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color = net(x, d)
            loss = opt.scale((color.float() ** 2).mean() + 1e-3 * sigma.float().mean())
        opt.backward(loss) if helper else loss.backward()
        grads.append([m.shadow.grad_half.clone() for m in (net.encoder, net.sigma_net, net.color_net)])
    assert float(grads[0][0].float().abs().sum()) > 0
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_library_workspace_growth_never_invalidates_a_captured_graph(O):
    """a graph captured at one size bakes the library's scratch pointers in; a later, larger eager call must not
    free that memory (buffers are retired, not freed), growth inside a capture is refused, and a replay stays correct"""
    from laenerf_amd import _lib
    from laenerf_amd.backend import gridencoder_backend as G
    offsets, pls, table, x = grid_case(O, B=8192)
    L, C = offsets.shape[0] - 1, 2
    S_ = np.log2(pls)
    g_small = half_from_bits(O.to_f16_bits((np.random.default_rng(1).standard_normal((L, 8192, C)) * 1e-2).astype(np.float32)))
    tx, toff = T(x), T(offsets)
    lib = _lib.load()
    lib.lae_free_workspaces()
    out = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        G.grid_encode_backward(g_small, tx, None, toff, out, 8192, 3, C, L, S_, 16, None, None, 0, False, 0)      # eager warm-up
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = out.clone()
    out.zero_()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        G.grid_encode_backward(g_small, tx, None, toff, out, 8192, 3, C, L, S_, 16, None, None, 0, False, 0)
    live0 = lib.lae_workspace_bytes(0)
    # a much larger eager call grows the workspace ...
    xb = T(np.random.default_rng(2).random((200000, 3)).astype(np.float32))
    gb = torch.zeros(L, 200000, C, device=DEV, dtype=torch.half)
    big = torch.zeros(table.shape, device=DEV, dtype=torch.half)
    G.grid_encode_backward(gb, xb, None, toff, big, 200000, 3, C, L, S_, 16, None, None, 0, False, 0)
    torch.cuda.synchronize()
    assert lib.lae_workspace_bytes(0) > live0 and lib.lae_workspace_bytes(1) >= live0       # ... and the old buffer is retired, not freed
    out.zero_()
    gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)                                     # the replay still writes through the retired buffer
    # growth inside a capture is refused with a message that says what to do
    lib.lae_free_workspaces()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="warm-up"):
        with torch.cuda.graph(g2):
            G.grid_encode_backward(g_small, tx, None, toff, out, 8192, 3, C, L, S_, 16, None, None, 0, False, 0)
    torch.cuda.synchronize()
