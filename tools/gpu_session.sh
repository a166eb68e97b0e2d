#!/bin/bash
# One gpurun session: steps run in order; a step that times out or is killed (rc 124 / 137 / >= 128) ends the
# session (no further GPU work after a hang), an ordinary failure (assertion, exception) does not.
# usage: tools/gpu_session.sh <tag> "<cmd1>" "<cmd2>" ...   (each cmd gets its own log gpurun_out/<tag>_<i>.log)
tag=$1; shift
mkdir -p gpurun_out
i=0
for cmd in "$@"; do
  i=$((i+1))
  log=gpurun_out/${tag}_${i}.log
  echo "== step $i: $cmd" | tee $log
  start=$(date +%s)
  timeout -k 10 ${STEP_TIMEOUT:-900} bash -c "$cmd" >> $log 2>&1
  rc=$?
  echo "== step $i rc=$rc in $(( $(date +%s) - start )) s" | tee -a $log
  tail -n 6 $log
  if [ $rc -ge 124 ]; then echo "step $i timed out or was killed: stopping the session"; exit $rc; fi
done
exit 0
