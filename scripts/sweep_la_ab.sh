# N = 1024 bench under the two forms of the split sweep (EKFVIO_SWEEP_LA=0: panel launch + tile launch per block step)
for LA in 0 1; do
EKFVIO_SWEEP_LA=$LA timeout -k 10 400 python bench.py --landmarks 1024 --steps 40 --warmup 5 --no-cpu-baseline --no-full-loop > gpurun_out/bench_la_$LA.json 2> gpurun_out/bench_la_$LA.err
python3 -c "
import json; r = json.load(open('gpurun_out/bench_la_$LA.json')); print('LA $LA', r['value'], r['ms_per_step'], {a: round(b,1) for a,b in r['stage_us_per_step'].items()})"
done
