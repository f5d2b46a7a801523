#!/bin/bash
# round-2 evidence run on the GPU box: driver-shaped bench line, kernel trace, HBM counter passes (separate --pmc runs)
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
tag=${1:-r2a}
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench_line.json; echo
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o bench -- python3 bench.py --steps 200 --no-cpu-baseline --no-frame --no-style > gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-graph > gpurun_out/${tag}_pmcf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o bench --output-format csv -- python3 bench.py --steps 40 --no-cpu-baseline --no-frame --no-style --no-graph > gpurun_out/${tag}_pmcw.log 2>&1
ls gpurun_out/${tag}_trace gpurun_out/${tag}_pmc_fetch | head
