cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
python tools/silbench.py 2>&1 | grep frames
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02f/pytest.log
tail -6 gpurun_out/r02f/pytest.log
