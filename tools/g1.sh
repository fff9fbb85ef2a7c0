cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r02a/bench_c2.json 2> gpurun_out/r02a/bench_c2.err; tail -c 600 gpurun_out/r02a/bench_c2.json
timeout 300 python bench.py --steps 30 --warmup 5 --launch graph --no-cpu-baseline > gpurun_out/r02a/bench_c2_graph.json 2> gpurun_out/r02a/bench_c2_graph.err; head -c 400 gpurun_out/r02a/bench_c2_graph.json
for b in 8 16 32; do for l in eager graph; do timeout 300 python bench.py --steps 50 --warmup 5 --batch $b --launch $l --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('batch',$b,'$l',j['value'],j['ms_per_step'],j['config']['launch'])"; done; done
GENDR_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r02a/bench_2ranks.json 2> gpurun_out/r02a/bench_2ranks.err; head -c 1500 gpurun_out/r02a/bench_2ranks.json; tail -3 gpurun_out/r02a/bench_2ranks.err
GENDR_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --config c4 --batch 32 --steps 5 --warmup 2 > gpurun_out/r02a/bench_c4_2ranks.json 2> gpurun_out/r02a/bench_c4_2ranks.err; head -c 1500 gpurun_out/r02a/bench_c4_2ranks.json; tail -3 gpurun_out/r02a/bench_c4_2ranks.err
