cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py -q -x 2>&1 | tail -3
python tools/kbench.py --config c4 --batch 32 --modes normal --iters 5 | grep normal
python tools/kbench.py --config c2 --modes normal --iters 30 | grep normal
python tools/kbench.py --config c3 --modes normal --iters 30 | grep normal
python tools/kbench.py --config c5 --batch 8 --modes normal --iters 8 | grep normal
python tools/shapebench.py 64 24 dist_func=logistic aggr_rgb_func=hard dist_eps=100
