export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for k in 0 4 6 8 12; do
  if [ $k != 0 ]; then export LAE_FRAME_MAX_ROUNDS=$k; fi
  a=$(python3 tools/frame_prof.py 5 2>/dev/null | tail -2 | awk '{print $3}' | tr '\n' ' ')
  b=$(python3 tools/frame1080_prof.py whole 4 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 6 2>/dev/null | tail -3 | awk '{print $4}' | tr '\n' ' ')
  echo "MAX_ROUNDS $k: 800 $a | whole $b | shard $c"
done
