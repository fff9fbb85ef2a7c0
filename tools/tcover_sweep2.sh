#!/bin/bash
# The tile limit of the coverage kernel's team form (team_cover(), kTeamCoverMaxTiles): shapes between 8192 and 65536 tiles whose tiles
# list faces by the dozen, with the shipped limit and with a library built -DGENDR_TEAM_COVER_MAX_TILES=65536 (x_tc64k.so at the repo root).
cd $GRAFT_REPO_ROOT
cp gendr_amd/libgendr_hip.so /tmp/base.so
for f in /tmp/base.so x_tc64k.so; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"
  python tools/shapebench.py 64 256 dist_func=uniform dist_scale=0.0316 aggr_rgb_func=hard dist_eps=300 2>&1 | tail -1
  python tools/shapebench.py 64 128 dist_func=logistic dist_scale=0.01 aggr_rgb_func=hard dist_eps=100 2>&1 | tail -1
  python tools/shapebench.py 64 256 dist_func=logistic dist_scale=0.01 aggr_rgb_func=hard dist_eps=100 2>&1 | tail -1
  python tools/shapebench.py 128 32 dist_func=logistic dist_scale=0.01 aggr_rgb_func=hard 2>&1 | tail -1
  python tools/shapebench.py 256 16 dist_func=logistic dist_scale=0.01 aggr_rgb_func=softmax 2>&1 | tail -1
  python tools/shapebench.py 256 64 dist_func=logistic dist_scale=0.01 aggr_rgb_func=softmax 2>&1 | tail -1
done
cp /tmp/base.so gendr_amd/libgendr_hip.so
