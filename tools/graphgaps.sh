#!/bin/bash
# kernel-to-kernel gaps of the bench step under eager launches and under graph replay (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp
for l in eager graph; do
  rm -rf /tmp/gg
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gg -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --launch $l > /tmp/gg.log 2>&1
  python - "$(find /tmp/gg -name '*kernel_trace.csv' | head -1)" $l <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
# steps: from a face_setup kernel to the next face_setup kernel
idx = [i for i, r in enumerate(rows) if 'face_setup' in r[2]]
spans = []
for a, b in zip(idx[:-1], idx[1:]):
    ks = rows[a:b]
    busy = sum(e - s for s, e, _ in ks)
    spans.append((ks[-1][1] - ks[0][0], busy, rows[b][0] - ks[0][0], len(ks), [k[2].split('(')[0][-28:] for k in ks]))
tail = spans[-12:]
print(sys.argv[2], 'kernels per step', tail[0][3], tail[0][4])
print('   median: first start -> last end %.1f us, kernel time %.1f us, step period %.1f us' % (
    sorted(t[0] for t in tail)[len(tail)//2] / 1e3, sorted(t[1] for t in tail)[len(tail)//2] / 1e3, sorted(t[2] for t in tail)[len(tail)//2] / 1e3))
PY
done
