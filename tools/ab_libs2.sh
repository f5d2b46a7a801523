#!/bin/bash
# same-box A/B of two libraries on the bench's headline objects: tools/ab_libs2.sh libA.so libB.so
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    LAE_HIP_LIB=$PWD/$lib python3 bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', 'step', j['ms_per_step'], 'enc_us', j['roofline']['avg_launch_us'], 'frame800', j['eval_frame']['ms_per_frame'], 'f1080', j['frame1080']['ms_per_frame'], 'shard', j['frame1080']['shard_of_8']['ms'], 'style', j['style_step']['ms_per_step'], 'flower', j['flower_step']['ms_per_step'], 'gridupd', j['grid_update']['partial_sweep_ms'], 'ops', j['operator_ms_per_step'])"
  done
done
