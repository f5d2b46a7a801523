"""-m gpu: the collectives of laenerf_amd/dist.py on real device tensors.

(1) ONE rank over RCCL: tools/dist_check.py in a fresh child process initialises the "nccl" (= RCCL) process group before any
other GPU call and, with the world-size-1 early returns switched off, runs gather_frame (all_gather_into_tensor),
broadcast_model_state (broadcast), the reduce_scatter_tensor + all_gather_into_tensor path of the table gradient and the
all_reduce of the MLP gradients, next to frame-loop launches that use the library's side stream (W = 1
moves no data but runs every RCCL entry point, dtype and stream hand-off the 8-GPU run uses; nerf/utils.py:380-382,
1555-1570).
(2) TWO ranks sharing cuda:0 over gloo: the same script under torch.distributed.run -- ranks start with different weights,
the broadcast must equalise parameters AND fp16 shadows (copies through `.data` left the shadows stale)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LAE_DIST_FORCE_COLLECTIVES", None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    return json.loads(lines[0])


def test_every_collective_executes_on_rccl_with_one_rank():
    j = _run([sys.executable, os.path.join(ROOT, "tools", "dist_check.py")])
    assert j["backend"] == "nccl" and j["world_size"] == 1 and j["ok"], j
    c = j["collectives"]
    assert any(x.startswith("float32:cuda") for x in c["all_gather_into_tensor"])            # the frame block
    assert any(x.startswith("float16:cuda") for x in c["all_gather_into_tensor"])            # second half of the table gradient
    assert any(x.startswith("float16:cuda") and int(x.split(":")[2]) >= (1 << 20) for x in c["reduce_scatter_tensor"])
    assert {x.split(":")[0] for x in c["broadcast"]} >= {"float32", "uint8"}                 # table / MLPs / grids, bitfield
    assert any(x.startswith("float16:cuda") for x in c["all_reduce"])                        # MLP weight gradients
    assert all(j["checks"].values()), j["checks"]


def test_two_ranks_over_gloo_share_weights_shadows_frames_and_gradients():
    port = 29500 + (os.getpid() + 4242) % 2000
    j = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), os.path.join(ROOT, "tools", "dist_check.py"), "--backend", "gloo"])
    assert j["backend"] == "gloo" and j["world_size"] == 2 and j["ok"], j
    for k in ("broadcast_shadows_follow_parameters", "broadcast_ranks_hold_rank0_table", "sharded_frame_equals_direct_bits",
              "ranks_hold_the_same_frame", "ranks_hold_the_same_gradient"):
        assert j["checks"][k] is True, k
