#!/bin/bash
# round 5, GPU call 4: dc2 on the direct tap-pair kernel instead of the specialised Winograd form; the cohort runner's parts; fixed test
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_fp16_range_gpu.py -q -k "calibration" 2>&1 | tail -5 > $O/r05_t_cal.log
for i in 1 2; do
  TAG=a$i OPTIONS= python3 scripts/seg_time.py 2>&1 | tail -1
  TAG=b$i OPTIONS=winograd_layers=229375 python3 scripts/seg_time.py 2>&1 | tail -1
  TAG=c$i OPTIONS=winograd=51 python3 scripts/seg_time.py 2>&1 | tail -1
done > $O/r05_ab_dc2.log 2>&1
python scripts/bench_cohort.py > $O/r05_cohort.log 2>&1
