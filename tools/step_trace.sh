#!/bin/bash
# kernel trace of a few steps -> step timeline + anatomy (no PMC passes):  tools/step_trace.sh tag
tag=${1:-x}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_trace.json 2> $R/gpurun_out/${tag}_trace.err || exit 1
cd $R
f=$(ls -t gpurun_out/${tag}_trace/*/*kernel_trace.csv | head -1)
python tools/trace_csv.py $f 8 > gpurun_out/${tag}_step_timeline.txt
python tools/trace_csv.py $f 8 --sum > gpurun_out/${tag}_step_anatomy.txt
rm -rf gpurun_out/${tag}_trace
