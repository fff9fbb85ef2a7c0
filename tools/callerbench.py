"""The shapes the reference's scripts render (bench.py caller_shapes_extra), alone -- for a kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/caller -- python tools/callerbench.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
print(json.dumps(bench.caller_shapes_extra(torch.device('cuda:0'), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 30), indent=1))
