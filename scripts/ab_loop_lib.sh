#!/bin/bash
# image-loop rate (bench.full_loop) of the library as built: bash scripts/ab_loop_lib.sh [reps]
for rep in $(seq 1 ${1:-3}); do python - <<'PY'
import bench
for N, kw in ((256, {}), (256, {"outputs": True}), (100, {"node_defaults": True, "outputs": True})):
    r = bench.full_loop(N, 0, frames=120, warm=6, **kw)
    print(N, kw, "frames/s %.0f" % r["frames_per_s"], {k: round(v, 1) for k, v in r["stage_us_per_frame"].items()})
PY
done
