#!/bin/bash
# Same-box A/B of several environment settings: scripts/ab_env2.sh N "A=1 B=2" "A=0" ... -> steps/s of bench.py per setting, interleaved 3x
N=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --steps 200 --warmup 20 --landmarks $N --no-cpu-baseline --no-full-loop 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('%-40s' % '$v', 'steps/s %.0f' % r['value'], 'us/step %.2f' % (1e3*r['ms_per_step']), {k: round(x,1) for k,x in r['stage_us_per_step'].items()})"
  done
done
