cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_c5.py tests/test_gpu_graph.py -q -x 2>&1 | tail -4
for b in 64 8; do python tools/kbench.py --config c2 --batch $b --modes normal --iters 30 | grep normal; done
python tools/kbench.py --config c5 --batch 8 --modes normal --iters 8 | grep normal
python tools/kbench.py --config c5 --batch 32 --modes normal --iters 5 | grep normal
python tools/kbench.py --config c4 --batch 32 --modes normal --iters 5 | grep normal
bash tools/ktrace.sh c5new --config c5 --batch 8 --modes normal --iters 8
