cd $GRAFT_REPO_ROOT
bash tools/r4_validate.sh
bash profiles/run_all.sh r04 > gpurun_out/r4_runall.log 2>&1
python __graft_entry__.py smoke 2>&1 | tail -2
