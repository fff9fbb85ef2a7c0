cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do for f in /tmp/base.so v_dg.so; do cp $f gendr_amd/libgendr_hip.so; echo "== $f"; python tools/kbench.py --config c5 --batch 16 --modes normal --iters 6 | grep normal; done; done
cp v_dg.so gendr_amd/libgendr_hip.so
timeout 600 python -m pytest tests/test_gpu_c5.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
cp /tmp/base.so gendr_amd/libgendr_hip.so
