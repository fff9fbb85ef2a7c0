#!/bin/bash
# backward kernel with 256 extra VALU instructions per batch vs the plain build, C4 and C5 (how VALU-bound are they?)
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for f in base v_pad256.so; do
  if [ $f = base ]; then cp /tmp/base.so gendr_amd/libgendr_hip.so; else cp $f gendr_amd/libgendr_hip.so; fi
  echo "== $f"
  python tools/kbench.py --config c4 --batch 16 --iters 5 2>&1 | grep normal
  python tools/kbench.py --config c5 --batch 4 --iters 5 2>&1 | grep normal
  python tools/kbench.py --config c3 --iters 20 2>&1 | grep normal
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
python tools/entrystats.py --config c4 --batch 2 | head -1
python tools/entrystats.py --config c5 --batch 2 | head -1
python tools/entrystats.py --config c3 --batch 8 | head -1
