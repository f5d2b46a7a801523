#!/bin/bash
# Reproducer of round 4's "two-process fault" and of its cause (laenerf_amd/build.py "erratum").
# Builds three probe libraries into tools/ubench/bin/ unless they exist:
#   liblaenerf_loop1_raw.so   k_grid_fwd_lean in the failing loop form (LAE_GRID_FWD_LOOP_PROBE=1), compiler defaults (packed fp32 ON)
#   liblaenerf_loop1_fix.so   the same source built like the shipped library (packed fp32 OFF)
#   liblaenerf_loop2_raw.so   the loop around the __forceinline__ function (LAE_GRID_FWD_LOOP_PROBE=2), compiler defaults
# then renders the same frame 150 times per (library, neighbour) pair and counts frames that differ from the first
# (tools/grid_loop_fault.py; one JSON line each), and runs the isolated instruction test (tools/ubench/pk_opsel).
#   tools/grid_loop_fault.sh [frames] [out-file]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
frames=${1:-150}; out=${2:-gpurun_out/r5_grid_loop_fault.txt}
B=tools/ubench/bin
mkdir -p $B "$(dirname "$out")"
for f in $B/liblaenerf_loop1_raw.so $B/liblaenerf_loop1_fix.so $B/liblaenerf_loop2_raw.so; do     # probe builds of older sources lack newer entry points
  [ -f $f ] && [ $f -ot laenerf_amd/lib/liblaenerf_hip.so ] && rm -f $f
done
[ -f $B/liblaenerf_loop1_raw.so ] || python3 -m laenerf_amd.build --out $B/liblaenerf_loop1_raw.so -DLAE_GRID_FWD_LOOP_PROBE=1 --packed-fp32 > /dev/null 2>&1 || exit 1
[ -f $B/liblaenerf_loop1_fix.so ] || python3 -m laenerf_amd.build --out $B/liblaenerf_loop1_fix.so -DLAE_GRID_FWD_LOOP_PROBE=1 > /dev/null 2>&1 || exit 1
[ -f $B/liblaenerf_loop2_raw.so ] || python3 -m laenerf_amd.build --out $B/liblaenerf_loop2_raw.so -DLAE_GRID_FWD_LOOP_PROBE=2 --packed-fp32 > /dev/null 2>&1 || exit 1
[ -x $B/spinner ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B/spinner tools/ubench/spinner.hip || exit 1
[ -x $B/pk_opsel ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B/pk_opsel tools/ubench/pk_opsel.hip || exit 1
: > "$out"
run() { timeout -k 10 400 python3 tools/grid_loop_fault.py --frames "$frames" "$@" 2>/dev/null | grep "^{" | cut -c1-420 >> "$out"; }
run --neighbour same                                                     # shipped library beside itself
for nb in none same mfma alu stream lds; do run --lib $B/liblaenerf_loop1_raw.so --neighbour $nb; done
run --lib $B/liblaenerf_loop1_fix.so --neighbour same
run --lib $B/liblaenerf_loop1_fix.so --neighbour mfma
run --lib $B/liblaenerf_loop2_raw.so --neighbour same
run --lib $B/liblaenerf_loop2_raw.so --neighbour mfma
echo "--- isolated instruction test (tools/ubench/pk_opsel.hip)" >> "$out"
timeout -k 10 300 $B/pk_opsel | cut -c1-260 >> "$out"
cat "$out"
