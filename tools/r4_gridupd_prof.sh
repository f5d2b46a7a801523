#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r4; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/gu -o t -- python3 tools/grid_update_bench.py > $out/gu.log 2>&1
python3 - $(find $out/gu -name "*kernel_trace.csv") <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last partial sweep of the first model (bound 1): find the last k_packbits of the first 24 and print the kernels between the previous packbits and it
pk=[i for i,r in enumerate(rows) if "k_packbits" in r["Kernel_Name"]]
a,b=pk[22],pk[23]
t0=int(rows[a+1]["Start_Timestamp"])
for r in rows[a+1:b+1]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:8.1f} us +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}  {r["Kernel_Name"][:110]}')
PY
