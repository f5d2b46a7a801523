#!/bin/bash
# round-4 frame-loop evidence: kernel traces of the 1080p frame (whole and one rank's shard of 8), the 800x800 frame and the
# style step.  usage: tools/r4_frame_profile.sh <tag>  ->  gpurun_out/r4/<tag>_*
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-base}
out=gpurun_out/r4
mkdir -p $out
for mode in whole shard; do
  rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_f1080_${mode} -o t -- python3 tools/frame1080_prof.py $mode 3 > $out/${tag}_f1080_${mode}.log 2>&1
  python3 tools/frame_trace_summary.py $(find $out/${tag}_f1080_${mode} -name "*kernel_trace.csv") 3 > $out/${tag}_f1080_${mode}_summary.txt 2>&1
  true
done
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_f800 -o t -- python3 tools/frame_prof.py 3 > $out/${tag}_f800.log 2>&1
python3 tools/frame_trace_summary.py $(find $out/${tag}_f800 -name "*kernel_trace.csv") 3 > $out/${tag}_f800_summary.txt 2>&1
true
python3 tools/frame1080_prof.py whole 5 > $out/${tag}_f1080_whole_notrace.log 2>&1
python3 tools/frame1080_prof.py shard 5 > $out/${tag}_f1080_shard_notrace.log 2>&1
python3 tools/frame_prof.py 5 > $out/${tag}_f800_notrace.log 2>&1
tail -n 3 $out/${tag}_f1080_whole_notrace.log $out/${tag}_f1080_shard_notrace.log $out/${tag}_f800_notrace.log
