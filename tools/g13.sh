#!/bin/bash
# per-kernel durations at small per-rank batches (strong-scaling floor)
cd /tmp; export TMPDIR=/tmp
for b in 8 16 64; do
  rm -rf /tmp/p$b
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$b -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --batch $b --no-cpu-baseline --launch eager > /dev/null 2>&1
  echo "== batch $b"
  f=$(find /tmp/p$b -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print('%-60s calls %5s avg %8.1f us  min %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
