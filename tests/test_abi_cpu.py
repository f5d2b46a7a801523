"""The C-ABI library loads without a GPU and exports every symbol include/laenerf.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "laenerf.h")).read()
    return sorted(set(re.findall(r"LAE_API\s+[\w\s\*]+?\b(lae_\w+)\s*\(", src)))


def test_header_declares_the_reference_backend_surface():
    names = set(header_symbols())
    # one symbol per function of the reference's pybind modules
    ref = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "march_rays_distill",
           "composite_rays", "composite_rays_distill", "grid_encode_forward", "grid_encode_backward",
           "grad_total_variation", "sh_encode_forward", "sh_encode_backward", "ffmlp_forward", "ffmlp_inference",
           "ffmlp_backward", "allocate_splitk", "free_splitk"]
    for r in ref:
        assert "lae_" + r in names, r


def test_library_exports_every_declared_symbol(hip_lib):
    for name in header_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in laenerf.h but not exported"


def test_python_binding_table_matches_header(hip_lib):
    from laenerf_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_no_compute_needed(hip_lib):
    assert hip_lib.lae_version().startswith(b"laenerf-hip gfx950")
    # the ABI tag: header, library and binding agree (a stale .so behind newer prototypes would misalign arguments silently;
    # _lib.load() compares before the first call and rebuilds or raises)
    from laenerf_amd import _lib
    hdr = re.search(r'#define\s+LAE_ABI_TAG\s+"(\w+)"', open(os.path.join(ROOT, "include", "laenerf.h")).read()).group(1)
    assert hip_lib.lae_version().split()[-1] == _lib.ABI_TAG == hdr.encode()
    # scratch-size helpers are pure host functions
    assert hip_lib.lae_march_rays_train_scratch_bytes(4096) >= 4 * 2 * 4096
    assert hip_lib.lae_compact_scratch_bytes(1000) >= 4000
    # binned grid backward: one region per (level, 1024 samples), worst case 8 items x 10 bytes per sample (fp16 gradients)
    assert hip_lib.lae_grid_backward_workspace_bytes(4096, 16, 1) >= 16 * 4096 * 8 * 10
    assert hip_lib.lae_workspace_bytes(0) == 0 and hip_lib.lae_workspace_bytes(1) == 0     # nothing allocated without a GPU call
    assert hip_lib.lae_free_workspaces() == 0


def test_invalid_arguments_are_rejected_before_any_launch(hip_lib):
    # N == 0 -> OK without touching pointers; NULL pointers -> LAE_ENULL; bad template params -> LAE_EINVAL
    assert hip_lib.lae_near_far_from_aabb(None, None, None, 0, 0.2, None, None, None) == 0
    assert hip_lib.lae_near_far_from_aabb(None, None, None, 8, 0.2, None, None, None) == -3
    one = ctypes.c_void_p(16)
    assert hip_lib.lae_sh_encode_forward(one, one, 4, 2, 4, None, None) == -1       # D must be 3
    assert hip_lib.lae_sh_encode_forward(one, one, 4, 3, 9, None, None) == -1       # degree <= 8
    assert hip_lib.lae_grid_encode_forward(one, one, one, one, 4, 3, 3, 16, 0.5, 16, None, 0, 0, 0, 0, None) == -1  # C=3
    assert hip_lib.lae_grid_encode_forward(one, one, one, one, 4, 6, 2, 16, 0.5, 16, None, 0, 0, 0, 0, None) == -1  # D=6
    assert hip_lib.lae_ffmlp_forward(one, one, 128, 32, 16, 48, 2, 0, 6, one, one, None) == -1   # hidden 48
    assert hip_lib.lae_ffmlp_forward(one, one, 128, 24, 16, 64, 2, 0, 6, one, one, None) == -1   # in % 16
