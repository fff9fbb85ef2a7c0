"""Host cost of one eager step of the headline bench (enqueue time of N steps before any synchronisation) against the GPU time.
    python tools/hostcost.py [--steps 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B

ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=200)
a = ap.parse_args()
args = B.parse_args(['--no-cpu-baseline'])
cfg = dict(B.CONFIGS['c2'])
dev = torch.device('cuda', 0)
wl = B.Workload(args, cfg, 0, 1, dev, 'weak')
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.1f us per step, with the GPU drained %.1f us per step' % ((t1 - t0) / a.steps * 1e6, (t2 - t0) / a.steps * 1e6), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    wl.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
