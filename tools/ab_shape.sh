#!/bin/bash
# A/B of library builds at one shape: bash tools/ab_shape.sh "<shapebench args>" name1.so name2.so ...   (files at the repo root)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
cp gendr_amd/libgendr_hip.so /tmp/base.so
for rep in 1 2; do
for f in "$@"; do
  cp $f gendr_amd/libgendr_hip.so
  echo "== $f"; python tools/shapebench.py $ARGS 2>&1 | tail -1
done; done
cp /tmp/base.so gendr_amd/libgendr_hip.so
