export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() {
  a=$(python3 tools/frame_prof.py 5 2>/dev/null | tail -2 | awk '{print $3}' | tr '\n' ' ')
  b=$(python3 tools/frame1080_prof.py whole 4 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 6 2>/dev/null | tail -3 | awk '{print $4}' | tr '\n' ' ')
  d=$(python3 tools/bench_frames.py 2>/dev/null | tail -1 | cut -c1-48)
  echo "$1: 800 $a | whole $b | shard $c | $d"
}
LAE_FRAME_EMIT_LDS=2 run lds2
run lds3
LAE_FRAME_EMIT_LDS=4 run lds4
