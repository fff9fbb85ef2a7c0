#!/bin/bash
# bench.py (autograd path, C2) with several library builds, interleaved: bash tools/bench_ab.sh a.so b.so
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/full.so
for rep in 1 2 3; do for f in "$@"; do cp $f gendr_amd/libgendr_hip.so; touch gendr_amd/libgendr_hip.so
  echo -n "$f: "; python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step; bwd kernel', round(d['roofline']['kernel_ms']*1e3,1) if 'kernel_ms' in d.get('roofline',{}) else '')"
done; done
cp /tmp/full.so gendr_amd/libgendr_hip.so
