#!/bin/bash
# builds and runs the conv3_wino_f32 ablation probe (every kernel under `timeout`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino_f32_ablate; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Xclang -target-feature -Xclang -packed-fp32-ops -I$R/include $R/scripts/micro/wino_f32_ablate.hip -o /tmp/wino_f32_ablate 2> $O/build.err || { tail -5 $O/build.err; exit 1; }
timeout 240 /tmp/wino_f32_ablate > $O/run.log 2>&1; cat $O/run.log
