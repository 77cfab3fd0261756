#!/bin/bash
# round 5, first GPU call: correctness of the new kernels, then A/B timing
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_unet_gpu.py -x -q 2>&1 | tail -15 > $O/r05_t_unet.log
for i in 1 2; do
  TAG=a$i OPTIONS=m16=1 python3 scripts/seg_time.py 2>&1 | tail -1
  TAG=b$i OPTIONS=m16=0 python3 scripts/seg_time.py 2>&1 | tail -1
done > $O/r05_ab_m16.log 2>&1
python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "golden" 2>&1 | grep -v "^$" | tail -60 > $O/r05_t_fullsize.log
bash scripts/layers_ab.sh "m16=0" "m16=1" > $O/r05_layers_m16.log 2>&1
