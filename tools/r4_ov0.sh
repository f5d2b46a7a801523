export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT LAE_FRAME_OVERLAP=0
out=gpurun_out/r4; tag=ov0; mkdir -p $out
for mode in whole shard; do
  rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_f1080_${mode} -o t -- python3 tools/frame1080_prof.py $mode 3 > $out/${tag}_f1080_${mode}.log 2>&1
  python3 tools/frame_trace_summary.py $(find $out/${tag}_f1080_${mode} -name "*kernel_trace.csv") 3 > $out/${tag}_f1080_${mode}_summary.txt 2>&1
  find $out/${tag}_f1080_${mode} -name "*kernel_trace.csv" -delete
done
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_f800 -o t -- python3 tools/frame_prof.py 3 > $out/${tag}_f800.log 2>&1
python3 tools/frame_trace_summary.py $(find $out/${tag}_f800 -name "*kernel_trace.csv") 3 > $out/${tag}_f800_summary.txt 2>&1
find $out/${tag}_f800 -name "*kernel_trace.csv" -delete
tail -n 2 $out/${tag}_f1080_whole.log $out/${tag}_f1080_shard.log $out/${tag}_f800.log
