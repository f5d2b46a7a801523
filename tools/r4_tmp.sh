export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
run() {
  b=$(python3 tools/frame1080_prof.py whole 4 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 5 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  a=$(python3 tools/bench_frames.py 2>/dev/null | tail -1 | cut -c1-60)
  echo "$1: whole $b | shard $c | $a"
}
run base
LAE_GRID_FWD_COSTS="1.5,1.5,1.5,1.5,1.5,1.41,1.58,1.75,1.93,2.10,2.41,2.90,3.39,3.88,4.37,4.87" run A_dense1.5
LAE_GRID_FWD_COSTS="0.6,0.6,0.6,0.6,0.6,1.41,1.58,1.75,1.93,2.10,2.41,2.90,3.39,3.88,4.37,4.87" run B_dense0.6
LAE_GRID_FWD_COSTS="1,1,1,1,1,1.41,1.58,1.75,1.93,2.1,2.3,2.6,2.9,3.2,3.5,3.8" run C_flat
LAE_GRID_FWD_COSTS="0.5,0.5,0.5,0.5,0.5,1.29,1.26,1.45,2.05,2.05,2.24,2.37,2.42,2.66,3.13,3.45" run D_meas_dense0.5
LAE_GRID_FWD_COSTS="0.3,0.3,0.3,0.3,0.3,1.0,1.2,1.4,1.7,2.0,2.3,2.6,2.9,3.2,3.5,3.8" run E_dense0.3
