cd $GRAFT_REPO_ROOT
bash profiles/run_all.sh r02a > gpurun_out/run_all_r02a.log 2>&1
tail -25 gpurun_out/run_all_r02a.log
for c in c2 c3 c4 c5; do echo "== $c"; head -c 420 gpurun_out/all_r02a/r02a_${c}_bench.json; echo; head -8 gpurun_out/all_r02a/r02a_${c}_kernel_stats.csv | cut -c1-110; done
timeout 1500 python tests/gpu_report.py r02 > gpurun_out/parity_r02.log 2>&1; tail -3 gpurun_out/parity_r02.log
