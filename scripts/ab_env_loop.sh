#!/bin/bash
# Same-box A/B of one environment knob on the node's frame loop: scripts/ab_env_loop.sh NAME A B
NAME=$1; A=$2; B=$3
for rep in 1 2 3; do
  for v in $A $B; do
    echo "== $NAME=$v"
    env $NAME=$v python scripts/node_outputs_ab.py 2>/dev/null
  done
done
