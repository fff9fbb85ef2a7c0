cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for f in x_tcover_off.so x_tcover.so; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"
  for sh in "64 24" "64 8" "128 8" "256 1" "256 4"; do
    for sg in 0.0001 0.001 0.003 0.01 0.03; do
      python tools/shapebench.py $sh dist_func=logistic aggr_rgb_func=hard dist_eps=100 dist_scale=$sg 2>&1 | tail -1
    done
  done
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
