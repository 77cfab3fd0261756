#!/bin/bash
# round 5, GPU call 6: two-group Winograd form with the fragments of step 0 requested at the chunk's top + pieces in steps 0..3 (this tree) against HEAD's library; brick cap 14 KB
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_unet_gpu.py -q -x 2>&1 | tail -3 > $O/r05_t_unet2.log
for i in 1 2 3; do
  TAG=new$i python3 scripts/seg_time.py 2>&1 | tail -1
  TAG=old$i OAI_LIB_PATH=build/exp/liboai_hip_rev.so python3 scripts/seg_time.py 2>&1 | tail -1
done > $O/r05_ab_chunktop.log 2>&1
BRICK=1 python3 scripts/bench_warp.py 2>&1 | grep -v "^{" > $O/r05_brick14.log
bash scripts/layers_ab.sh "" > $O/r05_layers_chunktop.log 2>&1
