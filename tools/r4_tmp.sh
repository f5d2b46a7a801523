export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for k in 3 5 7 8 0; do
  export LAE_FRAME_EMIT_LDS=$k
  a=$(python3 tools/frame_prof.py 6 2>/dev/null | tail -3 | awk '{print $3}' | tr '\n' ' ')
  b=$(python3 tools/frame1080_prof.py whole 5 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 6 2>/dev/null | tail -3 | awk '{print $4}' | tr '\n' ' ')
  echo "EMIT_LDS $k: 800 $a | whole $b | shard $c"
done
