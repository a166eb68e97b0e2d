#!/bin/bash
# Kernel trace of a few steps of the default bench under one LISEC_TUNING setting (run on the GPU box through gpurun):
#   tools/trace_one.sh <tag> ["key=value,..."]   -> gpurun_out/<tag>_step_timeline.txt, <tag>_step_anatomy.txt
tag=$1; tuning=$2
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export LISEC_TUNING="$tuning"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_trace.json 2> $R/gpurun_out/${tag}_trace.err || exit 1
cd $R
f=$(ls -t gpurun_out/${tag}_trace/*/*kernel_trace.csv | head -1)
python tools/trace_csv.py $f 8 > gpurun_out/${tag}_step_timeline.txt
python tools/trace_csv.py $f 8 --sum > gpurun_out/${tag}_step_anatomy.txt
head -1 gpurun_out/${tag}_step_anatomy.txt
