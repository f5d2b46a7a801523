#!/bin/bash
# rebuild on the box with FRAME_SPEC / FRAME_LANE_VISITS variants and time the three frames
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in "4 8" "8 8" "12 8" "16 16" "8 16"; do
  set -- $cfg
  sed -i "s/^constexpr int FRAME_SPEC = [0-9]*; /constexpr int FRAME_SPEC = $1; /; s/^constexpr uint32_t FRAME_LANE_VISITS = [0-9]*; /constexpr uint32_t FRAME_LANE_VISITS = $2; /" laenerf_amd/csrc/raymarching.hip
  python3 -m laenerf_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "SPEC $1 LANE_VISITS $2: $(python3 tools/frame_prof.py 5 | tail -1 | cut -d: -f2 | cut -d' ' -f2) / $(python3 tools/frame1080_prof.py whole 5 | tail -1 | cut -d: -f2 | cut -d' ' -f2) / $(python3 tools/frame1080_prof.py shard 5 | tail -1 | cut -d: -f2 | cut -d' ' -f2) ms"
done
