"""Voxelization (row f-2) timing: B=64 icosphere(3) meshes -> 32^3 (the evaluation setting of train_reconstruction.py:238)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gendr_amd import functional as Fn
from gendr_amd.synthetic import icosphere
from oracle import voxel_ref as V

v0, f0 = icosphere(3)
for B, vs in ((64, 32), (64, 64), (8, 128)):
    fv = (v0 * 0.8)[f0][None].repeat(B, axis=0).astype(np.float32) + 0.5
    t = torch.from_numpy(fv).cuda()
    for _ in range(5):
        Fn.voxelization(t, vs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        out = Fn.voxelization(t, vs)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 50
    line = 'B=%d nf=%d vs=%d: HIP %.3f ms per call (%.0f meshes/s)' % (B, f0.shape[0], vs, ms, B / ms * 1e3)
    if vs == 32:
        t0 = time.time()
        ref = V.voxelization(fv[:2], vs)
        dt = (time.time() - t0) / 2
        assert (out[:2].cpu().numpy() == ref).all()
        line += '; numpy oracle %.2f s per mesh (1 thread)' % dt
    print(line)
