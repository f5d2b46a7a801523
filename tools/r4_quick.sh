#!/bin/bash
# quick A/B on the GPU box: frame tests + untraced frame times (800x800, 1080p whole, shard of 8); env passes through
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-q}; out=gpurun_out/r4; mkdir -p $out
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 500 python3 -m pytest tests/test_gpu_frame.py tests/test_gpu_frame1080.py -x -q -m gpu > $out/${tag}_tests.log 2>&1 || { tail -30 $out/${tag}_tests.log; exit 1; }
  tail -3 $out/${tag}_tests.log
fi
python3 tools/frame_prof.py 5 > $out/${tag}_f800.log 2>&1
python3 tools/frame1080_prof.py whole 5 > $out/${tag}_whole.log 2>&1
python3 tools/frame1080_prof.py shard 5 > $out/${tag}_shard.log 2>&1
tail -n 2 $out/${tag}_f800.log $out/${tag}_whole.log $out/${tag}_shard.log
