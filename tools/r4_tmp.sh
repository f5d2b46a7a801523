export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for k in 0 1 2; do
  export LAE_GRID_FWD_FRAME_SCHED=$k
  a=$(python3 tools/frame_prof.py 6 2>/dev/null | tail -3 | awk '{print $3}' | tr '\n' ' ')
  b=$(python3 tools/frame1080_prof.py whole 4 2>/dev/null | tail -2 | awk '{print $4}' | tr '\n' ' ')
  c=$(python3 tools/frame1080_prof.py shard 6 2>/dev/null | tail -3 | awk '{print $4}' | tr '\n' ' ')
  echo "FRAME_SCHED $k: 800 $a | whole $b | shard $c"
done
LAE_BUILD_EXTRA_FLAGS=-DLAE_GRID_STAMPS python3 -m laenerf_amd.build --force > gpurun_out/r4/gs_build.log 2>&1 || exit 1
for k in 0 1 2; do
  echo "== LAE_GRID_FWD_FRAME_SCHED=$k"
  LAE_GRID_FWD_FRAME_SCHED=$k python3 tools/frame_grid_spans.py 10 30 2>&1 | grep -v amdgpu
done
