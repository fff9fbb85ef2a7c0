cd $GRAFT_REPO_ROOT
bash tools/ktrace.sh c2new --config c2 --modes normal --iters 20
bash tools/ktrace.sh c5new --config c5 --batch 8 --modes normal --iters 8
cp gendr_amd/libgendr_hip_exact.so /tmp/keep.so
