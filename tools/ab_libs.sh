#!/bin/bash
# same-box A/B of prebuilt libraries (laenerf_amd/lib/<tag>_liblaenerf_hip.so, selected with LAE_HIP_LIB): train step, style step, flower step
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for tag in "$@"; do
    echo "$tag: $(LAE_HIP_LIB=$PWD/laenerf_amd/lib/${tag}_liblaenerf_hip.so python3 bench.py --steps 20 --no-cpu-baseline --no-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['median'], 'style', d['style_step']['ms_per_step'], 'flower', d['flower_step']['ms_per_step'], 'bwd op', d['operator_ms_per_step']['grid_encode_backward'])")"
  done
done
