#!/bin/bash
# The GPU parity suite run BESIDE a second process that keeps the matrix pipe busy (tools/ubench/bin/spinner mfma): every test that
# is bit-exact alone must stay bit-exact there (gfx950 packed-fp32 erratum, docs/GFX950_PACKED_FP32_ERRATUM.md -- and anything else of its kind).  Round 6: the fuzz tests and the drop-in step are in the list.
# Timing-sensitive tests (bench contract, frame-loop degrade paths, multi-rank children) are left out.
#   tools/suite_beside_mfma.sh [out-file]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=${1:-gpurun_out/r6_suite_beside_mfma.txt}
B=tools/ubench/bin
[ -x $B/spinner ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B/spinner tools/ubench/spinner.hip || exit 1
$B/spinner mfma 1100 > /dev/null &
sp=$!
sleep 2
timeout -k 10 1000 python3 -m pytest -q -m gpu tests/test_gpu_raymarching.py tests/test_gpu_encoders.py tests/test_gpu_ffmlp.py tests/test_gpu_e2e.py \
    tests/test_gpu_optim.py tests/test_gpu_style.py tests/test_gpu_density_grid.py tests/test_gpu_rays.py tests/test_gpu_editgrid.py \
    tests/test_gpu_edit_dataset.py tests/test_gpu_train_loop.py tests/test_gpu_frame.py tests/test_gpu_frame1080.py \
    tests/test_gpu_frame_fuzz.py tests/test_gpu_fuzz_operators.py tests/test_gpu_dropin.py -k "not degrades and not matrix_pipe and not two_processes and not variants and not two_ranks" > "$out" 2>&1
rc=$?
kill $sp 2>/dev/null; wait $sp 2>/dev/null
echo "rc=$rc (beside spinner mfma)" >> "$out"
tail -4 "$out"
