"""Host-side logic that needs no GPU: sizing rules, parameter layout, error behaviour, no CPU fallback."""
import os

import numpy as np
import pytest
import torch

from conftest import golden


def test_level_offsets_match_reference_gridencoder(O):
    from laenerf_amd.gridencoder.grid import GridEncoder
    g = golden("grid_offsets")
    cfgs = [dict(), dict(desired_resolution=2048), dict(desired_resolution=4096), dict(num_levels=4, desired_resolution=2048),
            dict(input_dim=2, num_levels=4, log2_hashmap_size=19, desired_resolution=2048),
            dict(desired_resolution=2048, align_corners=True),
            dict(desired_resolution=1024, gridtype="tiled", log2_hashmap_size=15),
            dict(num_levels=8, level_dim=4, base_resolution=8, per_level_scale=1.5, log2_hashmap_size=14)]
    for i, kw in enumerate(cfgs):
        assert str(g[f"cfg{i}_kw"]) == repr(sorted(kw.items()))
        e = GridEncoder(**kw)
        assert np.array_equal(e.offsets.numpy(), g[f"cfg{i}_offsets"]), kw
        assert e.per_level_scale == pytest.approx(float(g[f"cfg{i}_pls"]), rel=0, abs=0)
        kw2 = {k: v for k, v in kw.items() if k != "gridtype"}
        off2, pls2 = O.grid_offsets(**kw2)
        assert np.array_equal(off2, g[f"cfg{i}_offsets"])
    # headline table: 6,119,864 entries at bound 1
    assert GridEncoder(desired_resolution=2048).embeddings.shape == (6119864, 2)
    assert GridEncoder(desired_resolution=4096).embeddings.shape == (6328848, 2)


def test_ffmlp_layout_init_and_padding_match_reference(hip_lib):
    from laenerf_amd.ffmlp import FFMLP
    g = golden("ffmlp_init")
    for name, args in (("sigma", (32, 16, 64, 2)), ("color", (32, 3, 64, 3)), ("wide", (48, 3, 128, 2))):
        m = FFMLP(*args)
        assert m.num_parameters == int(g[name + "_n"])
        assert m.padded_output_dim == int(g[name + "_padded_out"])
        w = m.weights.detach().numpy()
        assert np.array_equal(w[:256], g[name + "_head"])          # manual_seed(42) + U(+-sqrt(3/hidden))
        assert float(w.astype(np.float64).sum()) == pytest.approx(float(g[name + "_sum"]), abs=1e-9)
    # `pad = 128 - B % 128` always pads (ffmlp.py:157-159)
    for B, Bb in zip(g["pad_B_in"], g["pad_B_backend"]):
        assert B + (128 - B % 128) == Bb


def test_round_up_rule_of_march_rays_train():
    from laenerf_amd.raymarching.raymarching import _round_up_always
    g = golden("ops_wrappers")
    for mc in (1000, 1024, 5000):
        assert _round_up_always(mc, 128) == int(g[f"mc{mc}_M"])
    assert _round_up_always(int(g["counter"][0]), 128) == int(g["M_trimmed"])
    assert _round_up_always(77, -1) == 77


def test_no_cpu_fallback(hip_lib):
    """CPU tensors must be rejected loudly: the product path is HIP only."""
    from laenerf_amd.backend import raymarching_backend, gridencoder_backend, shencoder_backend, ffmlp_backend
    a = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        raymarching_backend.near_far_from_aabb(a, a, torch.zeros(6), 4, 0.2, torch.zeros(4), torch.zeros(4))
    with pytest.raises(RuntimeError, match="GPU"):
        shencoder_backend.sh_encode_forward(a, torch.zeros(4, 16), 4, 3, 4, None)
    with pytest.raises(RuntimeError, match="GPU"):
        gridencoder_backend.grid_encode_forward(a, torch.zeros(8, 2), torch.zeros(2, dtype=torch.int32), torch.zeros(1, 4, 2),
                                                4, 3, 2, 1, 0.5, 16, None, 0, False, 0)
    h = torch.zeros(128, 32, dtype=torch.half)
    with pytest.raises(RuntimeError, match="GPU"):
        ffmlp_backend.ffmlp_forward(h, h, 128, 32, 16, 64, 2, 0, 6, h, h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from laenerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    import os, re
    from conftest import ROOT
    for dirpath, _, files in os.walk(os.path.join(ROOT, "laenerf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|liblae_oracle|orc_", src, re.M), f


def test_synthetic_generators_are_seeded():
    from laenerf_amd import synthetic as S
    o1, d1 = S.lego_like_rays(64, seed=3)
    o2, d2 = S.lego_like_rays(64, seed=3)
    assert np.array_equal(o1, o2) and np.array_equal(d1, d2)
    assert np.allclose(np.linalg.norm(d1, axis=-1), 1, atol=1e-6)
    o, d = S.frame_rays(8, 8)
    assert o.shape == (64, 3) and d.shape == (64, 3)


def test_grid_forward_schedule_covers_every_chunk_once_and_balances(hip_lib):
    """host logic of the specialised grid forward (no GPU needed): every (level, chunk) appears in exactly one segment, no XCD
    holds more than 12 segments, whole hashed levels stay on one XCD (their table lives in that XCD's L2), and the
    modelled cost is balanced; small launches fall back to level l on XCD l mod 8"""
    import ctypes
    from laenerf_amd.gridencoder.grid import level_offsets
    for L, log2_T, desired, nb in ((16, 19, 2048, 973), (16, 19, 4096, 8100), (8, 14, 512, 300), (24, 19, 8192, 700), (16, 19, 2048, 12)):
        pls = np.exp2(np.log2(desired / 16) / (L - 1))
        offs = level_offsets(3, L, pls, 16, log2_T, False)
        nseg = (ctypes.c_uint32 * 8)()
        segs = (ctypes.c_uint32 * (8 * 12 * 3))()
        per_xcd = hip_lib.lae_grid_forward_schedule(offs.ctypes.data_as(ctypes.c_void_p), L, ctypes.c_float(np.log2(pls)), 16, nb, nseg, segs)
        assert per_xcd > 0
        seen = np.zeros((L, nb), np.int32)
        owners = [set() for _ in range(L)]
        blocks = []
        for x in range(8):
            assert nseg[x] <= 12
            tot = 0
            for q in range(nseg[x]):
                lvl, c0, n = segs[(x * 12 + q) * 3], segs[(x * 12 + q) * 3 + 1], segs[(x * 12 + q) * 3 + 2]
                seen[lvl, c0:c0 + n] += 1
                owners[lvl].add(x)
                tot += n
            blocks.append(tot)
        assert (seen == 1).all()
        assert max(blocks) == per_xcd
        res = [int(np.ceil(16 * pls ** l)) for l in range(L)]
        hashed = [(res[l] + 1) ** 3 > offs[l + 1] - offs[l] for l in range(L)]
        assert all(len(owners[l]) == 1 for l in range(L) if hashed[l])
        if nb < 64:                                                        # fallback map
            assert all(owners[l] == {l % 8} for l in range(L))
        elif L == 16:
            assert any(len(owners[l]) > 1 for l in range(L) if not hashed[l])   # dense levels are dealt out in pieces


def test_bench_refuses_a_rank_count_it_was_not_launched_with():
    """`bench.py --gpus N` under a launcher that set another WORLD_SIZE, or on a node with fewer than N GPUs, exits non-zero
    BEFORE any GPU call instead of printing a line for a different rank count (`--gpus 8` used to run one rank,
    print n_gpus 1 and exit 0)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LAE_BENCH_SINGLE_DEVICE")}
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()
    out = subprocess.run([sys.executable, bench, "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and not out.stdout.strip()
    import torch
    if torch.cuda.device_count() < 64:
        out = subprocess.run([sys.executable, bench, "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 2 and "asks for 64" in out.stderr and not out.stdout.strip()


def test_bench_imports_touch_no_gpu():
    """bench.py starts its ranks as a child process BEFORE any GPU call of the parent (round 5); what it imports at module level
    -- laenerf_amd.streams among it -- must therefore not initialise the GPU, and the stream helper has no CPU fallback."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys, torch; sys.path.insert(0, %r); import bench; import laenerf_amd.streams as st; "
            "print('init', torch.cuda.is_initialized()); "
            "\ntry:\n    st.concurrent_side_stream()\n    print('no error')\nexcept Exception as e:\n    print('raised', type(e).__name__)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-500:]
    assert "init False" in out.stdout
    import torch
    if not torch.cuda.is_available():
        assert "raised" in out.stdout and "no error" not in out.stdout
