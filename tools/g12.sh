cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; tail -3 gpurun_out/pytest_final.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline_oracle']['value'])"
