cd $GRAFT_REPO_ROOT
./tools/micro/bin/opbench
bash tools/ablate_prof.sh r02c
