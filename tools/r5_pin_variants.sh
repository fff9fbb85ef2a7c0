#!/bin/bash
# round 5, item 1: which of the default build's short cuts keep gaussian / gamma option sets from a flat 1e-5 against the
# reference's kernels?  t1 = exact CDF forms only, t2 = exact CDF + exact densities; and what each costs at C3 / C5.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp t1.so gendr_amd/libgendr_hip_t1.so; cp t2.so gendr_amd/libgendr_hip_t2.so
PIN_VARIANTS=default,t1,t2,exact python tests/golden/make_pin_table.py gamma gauss C3 C5 > gpurun_out/r5_pin_variants.log 2>&1
cp gpurun_out/pin_table.json gpurun_out/r5_pin_variants_table.json
for cfg in c3 c5; do
  for rep in 1 2; do
  for v in default t1 t2 exact; do
    echo "== $cfg $v" >> gpurun_out/r5_pin_variants.log
    GENDR_VARIANT=$v python tools/kbench.py --config $cfg --iters 20 --modes normal 2>&1 | grep normal >> gpurun_out/r5_pin_variants.log
  done; done
done
tail -60 gpurun_out/r5_pin_variants.log
