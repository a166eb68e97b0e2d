#!/bin/bash
# same-box A/B of LISEC_TUNING settings: tools/knob_sweep.sh "a=1" "b=2,c=3" ...   (each: python bench.py --steps 60 --no-cpu-baseline)
for w in "$@"; do
  LISEC_TUNING=$w timeout -k 10 200 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %7.1f samples/s  %.3f ms  R200k %.1f' % (sys.argv[1], d['value'], d['ms_per_step'], d['r200k']['value']))" "$w" || exit 1
done
