#!/bin/bash
# lookahead wave stamps on the GPU box (probe build with -DLAE_FRAME_STAMPS, in the box's copy only)
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r4; mkdir -p $out
LAE_BUILD_EXTRA_FLAGS=-DLAE_FRAME_STAMPS python3 -m laenerf_amd.build --force > $out/stamps_build.log 2>&1 || exit 1
LAE_STAMPS_SCENE=whole python3 tools/frame_look_stamps.py 3 30 100 > $out/stamps_whole.txt 2>&1
LAE_STAMPS_SCENE=shard python3 tools/frame_look_stamps.py 3 10 30 > $out/stamps_shard.txt 2>&1
tail -n 40 $out/stamps_whole.txt $out/stamps_shard.txt
