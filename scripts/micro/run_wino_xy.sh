#!/bin/bash
# builds and runs the x + y Winograd tap-stream probe, then an SQ counter pass of it (MFMA-busy, clock)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino_xy; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 $R/scripts/micro/wino_xy_taps.hip -o /tmp/wino_xy_taps 2> $O/build.err || { tail -5 $O/build.err; exit 1; }
timeout 180 /tmp/wino_xy_taps > $O/run.log 2>&1; cat $O/run.log; grep -q "fault" $O/run.log && exit 1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o s --output-format csv -- /tmp/wino_xy_taps > $O/sq.log 2>&1
ls $O/sq 2>/dev/null | head
