#!/bin/bash
# first-level sampling workgroups with padded LDS (one per CU): pipelined bench, alternating settings
out=gpurun_out/r3a; mkdir -p $out
for rep in 1 2; do
for pad in 0 52000 100000; do
  DPM_FPS_LDS_PAD=$pad python bench.py --steps 60 --warmup 5 --cpu-frames 0 --no-extras > $out/bench_pad${pad}_$rep.json 2> $out/bench_pad${pad}_$rep.err
  echo "pad=$pad rep=$rep $(python -c "import json,sys; j=json.load(open('$out/bench_pad${pad}_$rep.json')); print(j['value'], j['ms_per_step'], j['roofline']['us_per_round'], j['roofline_mfma']['achieved'])")"
done
done
