#!/bin/bash
# Kernel trace of a few steps on the R200k sweep:  tools/trace_r200k.sh <tag>  -> gpurun_out/<tag>_step_{timeline,anatomy}_r200k.txt
tag=$1
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace_r200k -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --cloud r200k > $R/gpurun_out/${tag}_trace_r200k.json 2> $R/gpurun_out/${tag}_trace_r200k.err || exit 1
cd $R
f=$(ls -t gpurun_out/${tag}_trace_r200k/*/*kernel_trace.csv | head -1)
python tools/trace_csv.py $f 8 > gpurun_out/${tag}_step_timeline_r200k.txt
python tools/trace_csv.py $f 8 --sum > gpurun_out/${tag}_step_anatomy_r200k.txt
head -40 gpurun_out/${tag}_step_anatomy_r200k.txt
