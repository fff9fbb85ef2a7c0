cd $GRAFT_REPO_ROOT
python tools/kbench.py --iters 30 2>&1 | grep -E "normal"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
python tools/kbench.py --iters 30 2>&1 | grep -E "normal"
