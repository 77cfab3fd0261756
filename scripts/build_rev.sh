#!/bin/bash
# The library of another revision, for same-box A/Bs with scripts/seg_time.py:  bash scripts/build_rev.sh [rev, default HEAD]  ->  build/exp/liboai_hip_rev.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); REV=${1:-HEAD}; W=/tmp/oai_rev_wt
git -C "$R" worktree remove --force $W >/dev/null 2>&1 || true
git -C "$R" worktree add -f --detach $W "$REV" >/dev/null
(cd $W && python3 -c "from oai_analysis_2_amd import build; build.build_library(force=True)" 2>&1 | tail -1)
mkdir -p "$R/build/exp" && cp $W/oai_analysis_2_amd/liboai_hip.so "$R/build/exp/liboai_hip_rev.so"
git -C "$R" worktree remove --force $W; git -C "$R" worktree prune
echo "$R/build/exp/liboai_hip_rev.so = $REV"
