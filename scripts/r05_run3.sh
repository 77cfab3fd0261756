#!/bin/bash
# round 5, GPU call 3: the whole GPU suite (not -x), then a bench line
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/r05_t_all.log
python bench.py --steps 5 --warmup 2 > $O/r05_bench1.json 2> $O/r05_bench1.err
