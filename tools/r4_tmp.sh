export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for k in 0.92 0.5 0.2 0.0; do
  export LAE_GRID_FWD_FRAME_SLOPE=$k
  d=$(python3 tools/bench_frames.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['eval_frame']['ms_per_frame'], d['frame1080']['ms_per_frame'], d['frame1080']['shard_of_8_ms'])")
  d2=$(python3 tools/bench_frames.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['eval_frame']['ms_per_frame'], d['frame1080']['ms_per_frame'], d['frame1080']['shard_of_8_ms'])")
  echo "SLOPE $k: $d | $d2"
done
