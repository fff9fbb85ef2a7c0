#!/bin/bash
# Per-kernel durations of one option set at one shape (rocprofv3 kernel trace around tools/shapebench.py):
#   bash tools/shapetrace.sh <tag> <shapebench args...>      -> gpurun_out/shapetrace_<tag>.txt
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/shapetrace_$TAG
rocprofv3 --kernel-trace -d gpurun_out/shapetrace_$TAG -o t -- python tools/shapebench.py "$@" > gpurun_out/shapetrace_$TAG.log 2>&1
python - "$TAG" <<'PY'
import sqlite3, glob, sys
tag = sys.argv[1]
db = glob.glob('gpurun_out/shapetrace_%s/**/*.db' % tag, recursive=True)[0]
c = sqlite3.connect(db)
q = """select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d
       join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like '%gendr%' group by s.kernel_name, d.grid_size_x order by 3 desc"""
out = open('gpurun_out/shapetrace_%s.txt' % tag, 'w')
for r in c.execute(q):
    line = '%-100s n=%4d avg=%8.1f us min=%8.1f grid=%d wg=%d' % (r[0][9:109], r[1], r[2] / 1e3, r[3] / 1e3, r[4], r[5])
    print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/shapetrace_$TAG
