#!/bin/bash
# same-box A/B of the accumulate pass's workgroups per level (LAE_GRID_BWD_BK_TARGET; default: 32 below 400 k samples, 64 beyond)
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for bk in ${BKS:-"" 32 64}; do
    if [ -n "$bk" ]; then export LAE_GRID_BWD_BK_TARGET=$bk; else unset LAE_GRID_BWD_BK_TARGET; fi
    f=$(python3 bench.py --workload flower --no-cpu-baseline --no-frame --no-style 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['flower_step']['ms_per_step'])")
    echo "bk=${bk:-default}: flower $f"
  done
done
