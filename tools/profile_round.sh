#!/bin/bash
# The profile set of a round, on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh r03
# 1. rocprofv3 --kernel-trace --stats of the default bench  -> gpurun_out/<tag>_stats/
# 2. kernel trace of a few steps                            -> gpurun_out/<tag>_trace/  (tools/trace_csv.py)
# 3. PMC passes, one counter set per pass (never together with trace domains other than --kernel-trace)
tag=${1:-r04}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.json 2> $R/gpurun_out/${tag}_stats.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_trace.json 2> $R/gpurun_out/${tag}_trace.err || exit 1
for set in "pmc_fetch:FETCH_SIZE" "pmc_write:WRITE_SIZE" "pmc_clk:GRBM_GUI_ACTIVE" "pmc_sq:SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  name=${set%%:*}; ctrs=${set#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc/$name -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_$name.log 2>&1 || exit 1
done
cd $R
f=$(ls -t gpurun_out/${tag}_trace/*/*kernel_trace.csv | head -1)
python tools/trace_csv.py $f 8 > gpurun_out/${tag}_step_timeline.txt
python tools/trace_csv.py $f 8 --sum > gpurun_out/${tag}_step_anatomy.txt
STEP_TIMELINE=gpurun_out/${tag}_step_timeline.txt python tools/pmc_summary.py --dominant gpurun_out/${tag}_pmc_dominant.json gpurun_out/${tag}_pmc/pmc_fetch gpurun_out/${tag}_pmc/pmc_write gpurun_out/${tag}_pmc/pmc_clk gpurun_out/${tag}_pmc/pmc_sq > gpurun_out/${tag}_pmc_summary.txt
cp $(ls -t gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
