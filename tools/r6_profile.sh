#!/bin/bash
# round-6 evidence run on the GPU box (ONE refresh per session end): driver-shaped bench line, kernel trace of the replayed step,
# HBM counter passes and SQ counter passes (train step, style step, inference frame -- every --pmc pass is its own run with
# --kernel-trace only, the program directly after `--`), flower / style / grid-update kernel stats, frame summaries.
# usage: tools/r6_profile.sh <tag>   ->  gpurun_out/<tag>_*; then locally: tools/r6_profile_collect.sh <tag>
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-r6a}
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 300 gpurun_out/${tag}_bench_line.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o bench -- python3 bench.py --steps 200 --no-cpu-baseline --no-frame --no-style --no-dropin > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-dropin --no-graph > gpurun_out/${tag}_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-dropin --no-graph > gpurun_out/${tag}_pmcw.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d gpurun_out/${tag}_pmc_sq -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-dropin --no-graph > gpurun_out/${tag}_pmcsq.log 2>&1
# the frame loop under a counter-collecting profiler runs its lookahead in line (dispatches are serialised: lae_render_frame's probe)
rocprofv3 --kernel-trace --pmc $SQ -d gpurun_out/${tag}_pmc_sq_frame -o frame --output-format csv -- python3 tools/frame_prof.py 2 > gpurun_out/${tag}_pmcsq_frame.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d gpurun_out/${tag}_pmc_sq_style -o style --output-format csv -- python3 tools/style_prof.py > gpurun_out/${tag}_pmcsq_style.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_flower -o flower -- python3 bench.py --workload flower > gpurun_out/${tag}_flower.json 2>gpurun_out/${tag}_flower.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_style -o s -- python3 tools/style_prof.py > gpurun_out/${tag}_style.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_gridupd -o gu -- python3 tools/grid_update_bench.py > gpurun_out/${tag}_gridupd.log 2>&1
# the zero-edit drop-in step (tools/reference_chain.py): two traces, 10 and 30 steady steps -> per-step kernel table by difference
for k in 10 30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_dropin_k$k -o d -- python3 tools/dropin_profile.py --steps $k > gpurun_out/${tag}_dropin_k$k.log 2>&1
done
python3 tools/dropin_launches.py $(find gpurun_out/${tag}_dropin_k10 -name "*kernel_stats.csv" | head -1) 10 $(find gpurun_out/${tag}_dropin_k30 -name "*kernel_stats.csv" | head -1) 30 --json gpurun_out/${tag}_drop_in_launches.json > gpurun_out/${tag}_dropin_kernel_table.txt 2>&1
for mode in whole shard; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_f1080_${mode} -o t -- python3 tools/frame1080_prof.py $mode 3 > gpurun_out/${tag}_f1080_${mode}.log 2>&1
  python3 tools/frame_trace_summary.py $(find gpurun_out/${tag}_f1080_${mode} -name "*kernel_trace.csv") 3 > gpurun_out/${tag}_frame1080_${mode}_summary.txt 2>&1
  true
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_f800 -o t -- python3 tools/frame_prof.py 3 > gpurun_out/${tag}_f800.log 2>&1
python3 tools/frame_trace_summary.py $(find gpurun_out/${tag}_f800 -name "*kernel_trace.csv") 3 > gpurun_out/${tag}_frame800_summary.txt 2>&1
# the per-dispatch counter tables are large: keep only what the summaries need (kernel name, counter, value)
for d in pmc_fetch pmc_write pmc_sq pmc_sq_frame pmc_sq_style; do
  for f in $(find gpurun_out/${tag}_${d} -name "*counter_collection.csv"); do
    python3 - "$f" <<'PY'
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
with open(f, "w", newline="") as o:
    w = csv.writer(o); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
    for r in rows:
        if "at::native" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]:
            continue
        w.writerow([r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]])
PY
  done
done
find gpurun_out/${tag}_* -name "*kernel_trace.csv" -delete
du -sh gpurun_out/${tag}_* | tail -12
