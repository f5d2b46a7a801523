#!/bin/bash
# after tools/r6_profile.sh <tag> came back through gpurun: copy the judged summaries into profiles/<tag>_*
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r6a}
cp gpurun_out/${tag}_bench_line.json profiles/${tag}_bench_line.json
cp $(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1) profiles/${tag}_bench_kernel_stats.csv
cp $(find gpurun_out/${tag}_flower -name "*kernel_stats.csv" | head -1) profiles/${tag}_flower_kernel_stats.csv
cp $(find gpurun_out/${tag}_style -name "*kernel_stats.csv" | head -1) profiles/${tag}_style_kernel_stats.csv
cp $(find gpurun_out/${tag}_gridupd -name "*kernel_stats.csv" | head -1) profiles/${tag}_gridupd_kernel_stats.csv
python3 tools/pmc_summarize.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write ${tag}
{ echo "# rocprofv3 --kernel-trace --pmc <8 SQ counters>: medians per dispatch.  Train step (bench.py --no-graph), then the LAENeRF palette step,"
  echo "# then one 800x800 inference frame (under counter collection dispatches are serialised and the frame loop runs its lookahead in line)."
  echo "## train step"; python3 tools/pmc_sq_summary.py gpurun_out/${tag}_pmc_sq
  echo "## style step"; python3 tools/pmc_sq_summary.py gpurun_out/${tag}_pmc_sq_style k_bwd k_palette k_style k_mlp k_grid_fwd k_apply
  echo "## inference frame"; python3 tools/pmc_sq_summary.py gpurun_out/${tag}_pmc_sq_frame k_frame k_grid_fwd k_near_far; } > profiles/${tag}_sq_stall_breakdown.txt
for f in frame800 frame1080_whole frame1080_shard; do cp gpurun_out/${tag}_${f}_summary.txt profiles/${tag}_${f}_summary.txt; done
cp gpurun_out/${tag}_dropin_kernel_table.txt profiles/${tag}_dropin_kernel_table.txt
cp gpurun_out/${tag}_drop_in_launches.json profiles/drop_in_launches.json
ls -la profiles/${tag}_*
