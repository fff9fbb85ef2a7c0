cd $GRAFT_REPO_ROOT
python tools/fuzzdbg.py 500 6 147 376 418 > gpurun_out/r06_fuzz500_seed6_arbitration.log 2>&1
grep -E "^[0-9]|ref kernels|oracle f32  *vs|ref kernels  *vs" gpurun_out/r06_fuzz500_seed6_arbitration.log | cut -c1-220
bash tools/ab_batches.sh "1 8 16 64" x_base.so x_app2.so x_pfb.so x_pfb4.so x_both.so 2>&1 | grep -v amdgpu
