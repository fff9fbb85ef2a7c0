cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_exact_math.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_projection.py -x -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02d/pytest.log
tail -8 gpurun_out/r02d/pytest.log
bash profiles/run_profile.sh r02d c2 > gpurun_out/r02d/profile.log 2>&1; head -8 gpurun_out/prof_r02d_c2/c2_kernel_stats.csv | cut -c1-150
cat gpurun_out/prof_r02d_c2/bench.json | head -c 400; echo
for b in 8 16; do timeout 300 python bench.py --steps 50 --warmup 5 --batch $b --launch graph --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('batch',$b,j['value'],j['ms_per_step'],j['config']['launch'], j['kernel_ms'])"; done
