#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_silhouette.py tests/test_gpu_api.py tests/test_native_abi.py -x -q 2>&1 | tail -12
python tools/kbench.py --config c2 --modes normal --iters 30 2>&1 | grep normal
GENDR_DETERMINISTIC=1 python tools/kbench.py --config c2 --modes normal --iters 10 2>&1 | grep normal
GENDR_DETERMINISTIC=1 python tools/kbench.py --config c3 --modes normal --iters 10 2>&1 | grep normal
GENDR_DETERMINISTIC=1 python tools/kbench.py --config c4 --modes normal --iters 3 --batch 32 2>&1 | grep normal
GENDR_DETERMINISTIC=1 python tools/kbench.py --config c5 --modes normal --iters 3 --batch 8 2>&1 | grep normal
