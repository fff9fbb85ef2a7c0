#!/bin/bash
# per-kernel average durations (rocprofv3 kernel trace) of kbench for each library given: bash tools/ktrace.sh "<kbench args>" a.so b.so
ARGS=$1; shift
cd /tmp; export TMPDIR=/tmp
cp $GRAFT_REPO_ROOT/gendr_amd/libgendr_hip.so /tmp/base.so
for f in "$@"; do
  cp $GRAFT_REPO_ROOT/$f $GRAFT_REPO_ROOT/gendr_amd/libgendr_hip.so
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py $ARGS > /tmp/kt.log 2>&1
  echo "== $f"
  python - "$(find /tmp/kt -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('  %-62s calls %5s avg %8.1f us  min %8.1f' % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
cp /tmp/base.so $GRAFT_REPO_ROOT/gendr_amd/libgendr_hip.so
