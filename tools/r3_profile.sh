#!/bin/bash
# round-3 evidence run on the GPU box (gpurun): driver-shaped bench line, kernel trace, HBM counter passes and one SQ counter
# pass -- every --pmc pass is its own run, with --kernel-trace only (the pool refuses --pmc beside the sys / hip / hsa traces).
# usage: tools/r3_profile.sh <tag>   ->  gpurun_out/<tag>_*; then locally:
#   python tools/pmc_summarize.py gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write <tag>
#   python tools/pmc_sq_summary.py gpurun_out/<tag>_pmc_sq > profiles/<tag>_sq_stall_breakdown.txt
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-r3a}
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench_line.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o bench -- python3 bench.py --steps 200 --no-cpu-baseline --no-frame --no-style > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-graph > gpurun_out/${tag}_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-graph > gpurun_out/${tag}_pmcw.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d gpurun_out/${tag}_pmc_sq -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-graph > gpurun_out/${tag}_pmcsq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_flower -o flower -- python3 bench.py --workload flower > gpurun_out/${tag}_flower.json 2>gpurun_out/${tag}_flower.err
find gpurun_out/${tag}_flower -name "*kernel_trace.csv" -delete
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_gridupd -o gu -- python3 tools/grid_update_bench.py > gpurun_out/${tag}_gridupd.log 2>&1
find gpurun_out/${tag}_gridupd -name "*kernel_trace.csv" -delete
# the per-dispatch counter tables are large: keep only what the summaries need (kernel name, counter, value)
for d in pmc_fetch pmc_write pmc_sq; do
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
  find gpurun_out/${tag}_${d} -name "*kernel_trace.csv" -delete
done
find gpurun_out/${tag}_trace -name "*kernel_trace.csv" -delete
du -sh gpurun_out/${tag}_* | tail -8
