cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r02b.log 2>&1; tail -3 gpurun_out/pytest_r02b.log
bash profiles/run_all.sh r02b > gpurun_out/run_all_r02b.log 2>&1
for c in c2 c3 c4 c5; do echo "== $c"; python -c "
import json; j=json.load(open('gpurun_out/all_r02b/r02b_${c}_bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['kernel_ms'])"; head -7 gpurun_out/all_r02b/r02b_${c}_kernel_stats.csv | cut -c1-120; done
./tools/micro/bin/opbench > gpurun_out/all_r02b/r02_opbench.txt; cat gpurun_out/all_r02b/r02_opbench.txt
timeout 1500 python tests/gpu_report.py r02 > gpurun_out/parity_r02.log 2>&1; tail -1 gpurun_out/parity_r02.log
python tools/silbench.py 2>&1 | grep frames
