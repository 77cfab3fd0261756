#!/bin/bash
# round 5, GPU call 2: the whole GPU suite with the tap-pair direct kernels + the two-level fp32 accumulation, then A/B timing
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/r05_t_all.log
for i in 1 2; do
  TAG=a$i OPTIONS=m16=1 python3 scripts/seg_time.py 2>&1 | tail -1
  TAG=b$i OPTIONS=m16=0 python3 scripts/seg_time.py 2>&1 | tail -1
done > $O/r05_ab_m16.log 2>&1
bash scripts/layers_ab.sh "m16=0" "m16=1" > $O/r05_layers_m16.log 2>&1
