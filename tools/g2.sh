cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02b/pytest.log
tail -8 gpurun_out/r02b/pytest.log
timeout 300 python tools/kbench.py --config c2 2>&1 | tail -4
timeout 300 python tools/kbench.py --config c2 --batch 8 2>&1 | tail -4
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02b/bench_c2.json 2> gpurun_out/r02b/bench_c2.err; head -c 300 gpurun_out/r02b/bench_c2.json; echo
bash profiles/run_profile.sh r02b c2 > gpurun_out/r02b/profile.log 2>&1; head -12 gpurun_out/prof_r02b_c2/c2_kernel_stats.csv
