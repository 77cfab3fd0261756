cd $GRAFT_REPO_ROOT
for v in base cur base cur; do
  if [ $v = base ]; then L=$GRAFT_REPO_ROOT/build/ab/lib_base.so; else L=$GRAFT_REPO_ROOT/oai_analysis_2_amd/liboai_hip.so; fi
  echo "== $v: $(OAI_LIB_PATH=$L python scripts/bench_icon.py 2>&1 | tail -1)"
done
