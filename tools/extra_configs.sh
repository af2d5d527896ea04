#!/bin/bash
# BASELINE configs that are not the default bench line: C3 (synthetic 10k-node lattice, HBM-roofline run) and C5 (high-res
# lattice, single-tick latency with the follow-mode profile on every tick). Writes gpurun_out/{c3_bench.json,c5_latency.json}.
python bench.py --workload c3 --batch 8192 --steps 50 --warmup 5 --latency-ticks 500 > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
tail -c 400 gpurun_out/c3_bench.err
python tools/c5_latency.py 300 > gpurun_out/c5_latency.json 2> gpurun_out/c5_latency.err
tail -c 400 gpurun_out/c5_latency.err
python tools/c5_latency.py 100 > gpurun_out/c5_latency_100m.json 2> gpurun_out/c5_latency_100m.err
cut -c1-300 gpurun_out/c3_bench.json; cat gpurun_out/c5_latency.json gpurun_out/c5_latency_100m.json
