export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoders.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace -d gpurun_out/prof_c1 -o bwd -- python3 tools/grid_bwd_bench.py > gpurun_out/r2_c1.log 2>&1
