#!/bin/bash
# same-box A/B of frame-loop settings: each argument is an "ENV=VALUE[,ENV=VALUE]" set ("-" = defaults); prints eval_frame / frame1080 / shard ms
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
one() { python3 tools/bench_frames.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['eval_frame']['ms_per_frame'], d['frame1080']['ms_per_frame'], d['frame1080']['shard_of_8_ms'], d['frame1080']['shard_ref_ms'], d['eval_frame']['max_abs_image_diff_vs_operator_loop'])"; }
for rep in 1 2 3; do
  for set in "$@"; do
    if [ "$set" = "-" ]; then echo "default: $(one)"; else echo "$set: $(env $(echo $set | tr ',' ' ') bash -c "$(declare -f one); one")"; fi
  done
done
