#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r4; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/style -o s -- python3 tools/style_prof.py > $out/style.log 2>&1
tail -2 $out/style.log
python3 - $(find $out/style -name "*kernel_stats.csv") <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
PY
find $out/style -name "*kernel_trace.csv" -delete
