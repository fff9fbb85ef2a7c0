#!/bin/bash
# The whole -m gpu suite and the default bench line on one box:  gpurun --timeout 3000 -- 'bash tools/r4_suite.sh'
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_suite.txt
cat gpurun_out/r4_suite.txt
timeout 900 python bench.py 2> gpurun_out/r4_bench.err | grep '^{' > gpurun_out/r4_c2_bench.json
tail -3 gpurun_out/r4_bench.err
python -c "
import json; j=json.load(open('gpurun_out/r4_c2_bench.json'))
print(j['value'], j['ms_per_step'], j['config']['launch'], j['roofline']['frac'])
print(json.dumps(j['extra'], indent=1)[:6000])"
