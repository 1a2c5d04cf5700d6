#!/bin/bash
# Same-box A/B of one environment knob: scripts/ab_env.sh NAME A B [landmarks] -> steps/s of bench.py for NAME=A and NAME=B, interleaved 3x
NAME=$1; A=$2; B=$3; N=${4:-256}
for rep in 1 2 3; do
  for v in $A $B; do
    env $NAME=$v python bench.py --steps 200 --warmup 20 --landmarks $N --no-cpu-baseline --no-full-loop 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('$NAME=$v', 'steps/s %.0f' % r['value'], 'us/step %.2f' % (1e3*r['ms_per_step']), 'gemm us %.2f frac %.3f' % (r['roofline']['avg_launch_us'], r['roofline']['frac']), {k: round(x,1) for k,x in r['stage_us_per_step'].items()})"
  done
done
