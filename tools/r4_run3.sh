cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_silhouette.py tests/test_gpu_graph.py tests/test_gpu_api.py tests/test_gpu_fill.py -q -x 2>&1 | tail -4
for b in 64 32 16 8 4 2 1; do python tools/kbench.py --config c2 --batch $b --modes normal --iters 30 | grep normal; done
bash tools/ktrace.sh c2b8 --config c2 --batch 8 --modes normal --iters 20
bash tools/ktrace.sh c2b1 --config c2 --batch 1 --modes normal --iters 20
