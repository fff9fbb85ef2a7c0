cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/keep.so
cp x_c3tab.so gendr_amd/libgendr_hip.so
python -m pytest tests/test_gpu_exact_math.py -q -s -k normal 2>&1 | grep -E "norm_cdf|passed|failed|Error" 
cp /tmp/keep.so gendr_amd/libgendr_hip.so
bash tools/ab_cfg.sh "--config c3 --iters 20 --modes normal" x_c3poly.so x_c3tab.so 2>&1 | grep -v amdgpu
