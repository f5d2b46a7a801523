export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() {
  a=$(python3 tools/frame_prof.py 5 2>/dev/null | tail -2 | awk '{print $3}' | tr '\n' ' ')
  b=$(python3 tools/frame1080_prof.py whole 4 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 6 2>/dev/null | tail -3 | awk '{print $4}' | tr '\n' ' ')
  echo "$1: 800 $a | whole $b | shard $c"
}
LAE_FRAME_ADMIT_CAP=0 run cap0
LAE_FRAME_ADMIT_CAP=1024 run cap1024
run cap4096_r2
LAE_FRAME_ADMIT_CAP=16384 run cap16384
LAE_FRAME_ADMIT_ROUND=1 run cap4096_r1
LAE_FRAME_ADMIT_ROUND=4 run cap4096_r4
