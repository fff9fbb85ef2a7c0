#!/usr/bin/env python
"""Headline benchmark: generalized soft rasterizer forward + backward, frames/s.

Workload (BASELINE.json configs[1], "C2"): synthetic 1280-face mesh, 256x256, GLOBAL batch 64,
dist_func=uniform, aggr_alpha_func=probabilistic, aggr_rgb_func=softmax, tau=1e-2, library defaults
otherwise.  A step = one forward and one backward of the autograd Function (`gendr_amd.functional.render`)
on inputs already resident in HBM.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), the batch axis is
sharded and the data path has no collective.  Headline (round 4, as up to round 2): STRONG scaling -- BASELINE.json's
metric is "batch 64; 1/2/4/8 GPU" and SURVEY 8(d) says "batch sharded evenly": 64 frames in total, 64 / N per GPU.
The WEAK figure (every GPU renders the config's batch: 64 N frames in all -- flat by construction, the data path has
no collective) is measured in the same run and reported under "extra" -> "weak" (`--scaling weak` swaps the two); on
one GPU the two are the same measurement, and "extra" -> "strong_projection" carries what the single-GPU batch sweep
predicts for the strong figure.
`python bench.py --gpus N` starts the N ranks itself when it is not already running under torchrun
(WORLD_SIZE unset); under the driver's `python -m torch.distributed.run ... bench.py --gpus N` it reads
RANK / LOCAL_RANK / WORLD_SIZE from the environment.

`--config c4` is BASELINE config 4: 256 views of 512^2 sharded over the ranks, and the step is
render -> all-gather of the views (`gendr_amd.dist.gather_views`) -> a loss that couples every view ->
backward (reduce-scatter of the view gradients, then the rasterizer's backward); the collective's share of the
step is reported.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     algorithmic HBM bytes of the dominant kernel / its measured average duration vs 8 TB/s
  "cpu_baseline": the pure-PyTorch evaluation of the same per-pixel math on all host cores (bounded sample);
                  "cpu_baseline_oracle" is the C/OpenMP oracle on the same cores;
  "reference_kernels_baseline": the reference's own kernels (oracle/_ref, built from the reference's .cu file by
                  oracle/build_ref.py) timed on the same GPU on the same workload, when that code object exists.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md); ~6290 GB/s measured achievable

CONFIGS = {
    # name: (GLOBAL batch, subdivisions, image_size, render options, texture)
    'c2': dict(batch=64, subdiv=3, image_size=256, texture='surface',
               opts=dict(dist_func='uniform', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax')),
    'c3': dict(batch=64, subdiv=3, image_size=256, texture='surface',
               opts=dict(dist_func='gaussian', dist_scale=1e-4, dist_squared=True, aggr_alpha_func='einstein', aggr_rgb_func='softmax')),
    'c4': dict(batch=256, subdiv=3, image_size=512, texture='surface', gather=True,
               opts=dict(dist_func='logistic', dist_scale=1e-2, aggr_alpha_func='probabilistic', aggr_rgb_func='softmax')),
    'c5': dict(batch=32, subdiv=3, image_size=2048, texture='vertex',
               opts=dict(dist_func='gamma', dist_shape=2.0, dist_scale=1e-2, aggr_alpha_func='yager', aggr_alpha_t_conorm_p=2.0,
                         aggr_rgb_func='softmax', texture_type='vertex')),
}


def _active_variant():
    from gendr_amd import _native
    return _native._active


def algorithmic_bytes(P, nf, T):
    """every tensor crossing the Function boundary touched once.
    forward: read faces 36 nf + textures 12 T nf, write RGBA 16 P;
    backward: read faces + textures again, RGBA 16 P, grad RGBA 16 P, write grad_faces 36 nf + grad_textures 12 T nf."""
    per_face = 36 + 12 * T
    fwd = 16 * P + per_face * nf
    bwd = 32 * P + 2 * per_face * nf
    return fwd, bwd


# ------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; bounded samples)
# ------------------------------------------------------------------------------------------------------------
def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that sees
    128 CPUs but is throttled to a quota runs a 128-thread PyTorch pool far slower than a right-sized one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            a, b = f.read().split()[:2]
            if a != 'max':
                quota = float(a) / float(b)
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline_oracle(cfg, fv, tex, target_seconds=12.0):
    """Times the CPU oracle (oracle/, test infrastructure used here only as a reported baseline)
    on a bounded sample of the same workload with all host cores."""
    import numpy as np
    import oracle
    oracle.build()
    isz = cfg['image_size']
    opts = dict(cfg['opts'])
    opts.setdefault('double_side', False)
    cores = min(oracle.max_threads(), usable_cores())          # OpenMP threads = cores the cgroup quota really grants
    oo = oracle.make_opts(image_size=isz, num_threads=cores, **opts)
    fvn = fv.cpu().numpy()
    texn = tex.cpu().numpy()
    rs = np.random.RandomState(1)

    def run(n):
        g = rs.randn(n, 4, isz, isz).astype(np.float32)
        t0 = time.perf_counter()
        fwd = oracle.forward(fvn[:n], texn[:n], oo)
        oracle.backward(fwd, g, oo)
        return time.perf_counter() - t0

    t1 = run(1)
    n = int(max(1, min(fvn.shape[0], target_seconds / max(t1, 1e-3))))
    tn = run(n) if n > 1 else t1
    return dict(value=n / tn, unit='frames/s', cores=cores, kind='port',
                sample='%d frame(s) of the same workload (%dx%d, %d faces), forward+backward, C oracle with OpenMP on %d threads, %.1f s'
                       % (n, isz, isz, fvn.shape[1], cores, tn))


def gpu_baseline_reference_kernels(cfg, fv, tex, steps=3):
    """The REFERENCE's own kernels (oracle/_ref: the device half of the reference's generalized_renderer_cuda_kernel.cu
    compiled for gfx950 by oracle/build_ref.py; test infrastructure, used here only as a reported baseline like the CPU
    legs) timed on this GPU on the same workload: thread per pixel, every face visited for every pixel, per-pair float
    atomics -- the design this repository replaces, on the hardware it is replaced on.  None when oracle/_ref was not
    built.  The build with the compiler's default contraction is timed (what installing the upstream package gives)."""
    try:
        from oracle import ref_gpu
        if not ref_gpu.available():
            return None
        from gendr_amd.functional import renderer as R
        isz = cfg['image_size']
        o = dict(R_DEFAULTS)
        o.update(cfg['opts'])
        o.setdefault('double_side', False)
        p = R.make_params(isz, [0., 0., 0.], *[o[k] for k in R_KEYS])
        ms = ref_gpu.time_step(fv.cpu().numpy(), tex.cpu().numpy(), isz, p, steps=steps, warmup=1)
        B = fv.shape[0]
        return dict(value=B / ms * 1e3, unit='frames/s', ms_per_step=ms, kind='reference device code on this GPU',
                    sample='%d steps of the same workload (batch %d, %dx%d, %d faces), forward+backward; the reference\'s '
                           'kernels from oracle/_ref (clang default contraction), launch shapes of kernel.cu:1099-1222, '
                           'output buffers re-initialised per step as functional/renderer.py does' % (steps, B, isz, isz, fv.shape[1]))
    except Exception as e:
        return dict(error='%s: %s' % (type(e).__name__, e))


R_KEYS = ('dist_func', 'dist_scale', 'dist_squared', 'dist_shape', 'dist_shift', 'dist_eps', 'aggr_alpha_func',
          'aggr_alpha_t_conorm_p', 'aggr_rgb_func', 'aggr_rgb_eps', 'aggr_rgb_gamma', 'near', 'far', 'double_side', 'texture_type')
R_DEFAULTS = dict(dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None, dist_eps=1e4,
                  aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None, aggr_rgb_func='softmax', aggr_rgb_eps=1e-3,
                  aggr_rgb_gamma=1e-3, near=1, far=100, double_side=True, texture_type='surface')


def _torch_baseline_worker(q, cfg, fv, tex, stride, threads, budget):
    """Child process: times 1 frame on every `stride`-th face; starts from a coarse sample and refines while the
    projected time fits the budget.  Puts (seconds, stride) or ('error', text)."""
    try:
        import torch
        from oracle import torch_ref
        torch.set_num_threads(threads)
        opts = dict(cfg['opts'])
        opts.setdefault('double_side', False)
        isz = cfg['image_size']
        g = torch.randn(1, 4, isz, isz, generator=torch.Generator().manual_seed(1))

        def run(st):
            t0 = time.perf_counter()
            torch_ref.render(fv[:1, ::st].contiguous(), tex[:1, ::st].contiguous(), isz, grad=g, **opts)
            return time.perf_counter() - t0

        st = max(stride, 64)
        dt = run(st)
        dt = run(st)                                  # the first call pays for thread-pool start-up and page faults
        while st > stride and dt * 2.2 < budget:
            st //= 2
            dt = run(st)
        q.put((dt, st))
    except Exception as e:      # reported in the JSON line instead of vanishing with the child
        q.put(('error', '%s: %s' % (type(e).__name__, e)))


def cpu_baseline_torch(cfg, fv, tex, stride=4, timeout=200, budget=25.0):
    """The pure-PyTorch evaluation of the same per-pixel math (oracle/torch_ref.py) that BASELINE.json's
    north_star asks for next to the GPU number, on ALL host cores.  It evaluates every (pixel, face) pair, so its
    cost is linear in the face count: a BOUNDED sample (1 frame, every k-th face, k chosen so that the sample takes
    about `budget` seconds at most) is timed in a child process with a hard timeout and scaled by k.  Only for
    option sets the restatement covers.  Returns a dict; on failure a dict with 'error'."""
    import multiprocessing as mp
    from oracle import torch_ref
    opts = cfg['opts']
    if opts.get('dist_func') not in torch_ref.DIST or opts.get('aggr_alpha_func') not in torch_ref.ALPHA or 'dist_shape' in opts:
        return None
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    threads = usable_cores()
    p = ctx.Process(target=_torch_baseline_worker, args=(q, cfg, fv[:1].cpu(), tex[:1].cpu(), stride, threads, budget))
    p.start()
    p.join(timeout)
    if p.is_alive():
        p.kill()
        p.join()
        return dict(error='pure-PyTorch baseline exceeded its %d s limit' % timeout)
    if q.empty():
        return dict(error='pure-PyTorch baseline child exited with code %s' % p.exitcode)
    dt, st = q.get()
    if dt == 'error':
        return dict(error=st)
    nf = fv.shape[1]
    return dict(value=1.0 / (dt * st), unit='frames/s', cores=threads, kind='port',
                sample='1 frame, every %dth of the %d faces (all-pairs evaluation, linear in faces), forward+backward, '
                       'vectorised pure PyTorch (oracle/torch_ref.py), torch.set_num_threads(%d) = all usable host cores '
                       '(os.cpu_count() = %d, affinity / cgroup quota applied); %.1f s scaled x%d'
                       % (st, nf, threads, os.cpu_count() or 0, dt, st))


# ------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` outside torchrun starts its own N ranks
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """Re-executes this script as n ranks of one node through torch.distributed.run (rendezvous on 127.0.0.1).
    Rank 0's JSON line is the only thing the children print on stdout."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='GLOBAL batch (default: the config\'s)')
    ap.add_argument('--scaling', default='strong', choices=('strong', 'weak'),
                    help='strong (headline, SURVEY 8(d): "batch sharded evenly"): the config\'s batch in all, 1/N of it per GPU; weak: the '
                         'config\'s batch per GPU (reported under extra otherwise)')
    ap.add_argument('--launch', default='auto', choices=('auto', 'eager', 'graph'),
                    help='graph: the step is captured once in a HIP graph and replayed (configs without a collective); '
                         'auto: graph when a rank holds at most 32 frames (the step is then about as short as the host\'s '
                         'launch work, 0.18-0.24 ms per step depending on the box: batch 8 eager 0.25 ms, replayed '
                         '0.13 ms), eager otherwise')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL; gloo with --stub)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the second measurement (the other scaling mode) at N > 1')
    ap.add_argument('--no-cull', action='store_true', help='visit every (pixel, face) pair (diagnostic)')
    ap.add_argument('--stub', action='store_true',
                    help='no GPU: the step is a small CPU tensor op (exercises launcher, sharding, timing and JSON on gloo)')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------
# one measurement: W warm-up steps, K timed steps between fences, max over ranks
# ------------------------------------------------------------------------------------------------------------
class Workload:
    """The per-rank shard of a config and its step function."""

    def __init__(self, args, cfg, rank, world, dev, scaling):
        import torch
        from gendr_amd.dist import shard_range
        from gendr_amd.synthetic import benchmark_scene
        self.torch = torch
        self.cfg, self.rank, self.world, self.dev = cfg, rank, world, dev
        self.isz = cfg['image_size']
        self.opts = dict(cfg['opts'])
        self.opts.setdefault('double_side', False)          # gendr.GenDR() default (gendr/renderer.py:34)
        base = args.batch or cfg['batch']
        self.global_batch = base if scaling == 'strong' else base * world
        a, b = shard_range(self.global_batch, rank, world)
        if cfg.get('gather') and self.global_batch % world:
            raise SystemExit('config %s needs a global batch divisible by the number of ranks' % args.config)
        self.B = b - a
        # every rank renders its own shard of views: the cameras differ per view
        fv_all, tex_all = benchmark_scene(self.global_batch, subdivisions=cfg['subdiv'], texture=cfg['texture'], seed=0)
        self.fv_cpu, self.tex_cpu = fv_all, tex_all
        if args.stub:
            self.fv = fv_all[a:b].clone().requires_grad_(True)
            self.tex = tex_all[a:b].clone().requires_grad_(True)
            self.nf, self.T = self.fv.shape[1], self.tex.shape[2]
            return
        self.fv = fv_all[a:b].to(dev).requires_grad_(True)
        self.tex = tex_all[a:b].to(dev).requires_grad_(True)
        self.nf, self.T = self.fv.shape[1], self.tex.shape[2]
        g = torch.Generator(device='cpu').manual_seed(1 + a)
        self.grad = torch.randn(max(self.B, 1), 4, self.isz, self.isz, generator=g)[:self.B].to(dev)
        if cfg.get('gather'):
            # weights of the coupling loss: every rank holds the weights of ALL views (different per rank)
            self.w_all = torch.rand(self.global_batch, 1, 1, generator=g).to(dev)

    def step_stub(self):
        (self.fv * 2.0).sum().backward()

    def step(self, events=None):
        """One forward + backward.  `events`: four torch.cuda.Event (fwd start/end, bwd start/end) recorded on the
        launch stream around the native calls, or None."""
        from gendr_amd.functional import renderer as R
        self.fv.grad = None
        self.tex.grad = None
        R.PROFILE_EVENTS = events
        try:
            img = R.render(self.fv, self.tex, image_size=self.isz, **self.opts)
            if self.cfg.get('gather'):
                from gendr_amd.dist import gather_views
                views = gather_views(img, assume_equal_blocks=True)                                # [global batch, 4, is, is] on every rank
                sil = views[:, 3]
                # couples every view: weighted silhouettes against the mean silhouette over ALL views
                loss = ((sil - sil.mean(0, keepdim=True)) ** 2 * self.w_all).mean() + (views[:, :3] * self.w_all[:, None]).mean()
                loss.backward()
            else:
                img.backward(self.grad)
        finally:
            R.PROFILE_EVENTS = None


def measure(args, wl, dist, dev):
    """Returns dict(elapsed, fwd_ms, bwd_ms, n_events, coll_ms, launch)."""
    torch = wl.torch
    stub = args.stub

    def fence():
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    if stub:
        for _ in range(args.warmup):
            wl.step_stub()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step_stub()
        fence()
        elapsed = time.perf_counter() - t0
        launch = 'stub'
        events, coll = [], []
    else:
        import gendr_amd.dist as gdist
        launch = args.launch
        calibrate = False
        if launch == 'auto':
            # Up to 32 frames per rank a captured graph wins clearly.  Above, the eager step's host work (~165 us on a fast
            # host, more on others) is close to the GPU's step (~225 us at C2): eager launches are ~4 % faster than a replayed
            # graph where the host keeps ahead, ~6 % slower where it does not -- so both are timed on a few untimed steps after
            # the warm-up and the faster one is used for the timed region ("launch" in the output says which).
            launch = 'graph'
            calibrate = wl.B > 32
        if launch == 'graph' and wl.cfg.get('gather'):
            launch = 'eager'                          # a collective inside the step: not captured
            calibrate = False
        graph = None
        for _ in range(args.warmup):
            wl.step()
        if launch == 'graph' and wl.B > 0:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        wl.step()
                torch.cuda.current_stream().wait_stream(s)
                wl.fv.grad = None
                wl.tex.grad = None
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):   # RCCL's watchdog thread may query events meanwhile
                    wl.step()
                for _ in range(2):
                    graph.replay()
            except Exception as e:                    # capture refused on this stack: measure eagerly and say so
                sys.stderr.write('bench: graph capture failed (%s); eager launches\n' % (e,))
                graph = None
                launch = 'eager'
            if graph is not None and calibrate:
                def _timed(fn, n=8):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t) / n
                t_graph = min(_timed(graph.replay), _timed(graph.replay))
                t_eager = min(_timed(wl.step), _timed(wl.step))
                # (eager has to win clearly: in the timed region it also pays for a second event-sampled step and for the host's
                # forward path ahead of the very first launch -- about 6 % of a 20-step window)
                if t_eager * 1.06 < t_graph:
                    graph = None
                    launch = 'eager'
        # Kernel durations come from HIP events around the native calls, recorded live inside the timed region -- but
        # only on TWO of the steps, the first and the last: a timed event serialises the launches around it (about 10 us each on
        # this stack; four per step cost 11 % of the throughput when every step carried them), and in the middle of the loop it
        # also lets the queue run dry, which the host -- only ~15 us per step faster than the GPU at C2 -- takes steps to refill.
        # Sampled steps are always launched eagerly (an event cannot be recorded inside a replayed graph).
        events, coll = [], []
        # (a replayed graph: the last step only -- a sampled step is launched eagerly, and at 8 frames per rank an eager step costs
        # the host three replayed ones)
        sampled = {args.steps - 1} if graph is not None else {0, args.steps - 1}
        # (the event objects exist before the clock starts: creating one costs the host ~10 us)
        ready = {i: [torch.cuda.Event(enable_timing=True) if (k >= 2 or i == min(sampled)) else None for k in range(4)] for i in sampled}
        import gc
        gc.collect()
        gc.disable()                                  # a collection in the middle of 20 steps of 0.24 ms is a 5 % outlier
        try:
            fence()
            t0 = time.perf_counter()
            for i in range(args.steps):
                if i in sampled and wl.B > 0:
                    e = ready[i]
                    if wl.cfg.get('gather'):
                        gdist.PROFILE_EVENTS = c = []
                    wl.step(e)
                    gdist.PROFILE_EVENTS = None
                    events.append(e)
                    if wl.cfg.get('gather'):
                        coll.append(c)
                elif graph is not None:
                    graph.replay()
                else:
                    wl.step()
            fence()
            elapsed = time.perf_counter() - t0
        finally:
            gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if stub else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = dict(elapsed=elapsed, launch=launch, fwd_ms=None, bwd_ms=None, n_events=len(events), coll_ms=None)
    if events:
        fwd_samples = [e for e in events if e[0] is not None]
        out['fwd_ms'] = sum(e[0].elapsed_time(e[1]) for e in fwd_samples) / len(fwd_samples)
        out['bwd_ms'] = sum(e[2].elapsed_time(e[3]) for e in events) / len(events)
    if coll and all(coll):
        out['coll_ms'] = sum(sum(a.elapsed_time(b) for a, b in c) for c in coll) / len(coll)
    return out



# ------------------------------------------------------------------------------------------------------------
# extras of the single-GPU line: the flagged fast build variant, eager vs replayed launches, and the
# shapes the reference's own scripts render
# ------------------------------------------------------------------------------------------------------------
def fast_variant_extra(args, cfg, rank, world, dev, dist):
    """The same measurement through libgendr_hip_fast.so (gendr_amd/build.py VARIANTS['fast']: the reference's formulas,
    order, skip tests and culling with the per-pair arithmetic at hardware accuracy and contraction on).  Never the
    headline: it does not reproduce the reference's rounding; its parity standing is quoted from the committed table."""
    from gendr_amd import _native, build
    if not os.path.exists(build.lib_path('fast')) or build.needs_build('fast'):
        return {'skipped': 'libgendr_hip_fast.so is built on request only since round 5 (python -m gendr_amd.build fast); round 4 measured '
                           '+7.7 % at this config on the driver\'s box (BENCH_r04.json)'}
    with _native.use_variant('fast'):
        wl = Workload(args, cfg, rank, world, dev, args.scaling)
        a = argparse.Namespace(**dict(vars(args), steps=max(10, args.steps), warmup=3))
        m = measure(a, wl, dist, dev)
    out = {'value': wl.global_batch * a.steps / m['elapsed'], 'unit': 'frames/s', 'ms_per_step': m['elapsed'] / a.steps * 1e3,
           'launch': m['launch'], 'steps': a.steps,
           'kernel_ms': {'forward_phase': m['fwd_ms'], 'backward_phase': m['bwd_ms']},
           'what': '-DGENDR_FAST_MATH=1 -ffp-contract=on: float reciprocals (<= 1 ulp) instead of exactly rounded quotients, v_sqrt_f32, '
                   '2^x-based exp, float instead of double sub-expressions, contraction on; formulas, operation order, skip tests and '
                   'culling unchanged'}
    try:
        t = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference', 'pin_table.json')))
        name = args.config.upper()
        br = t.get('fast_bracket', {}).get(name, {})
        out['parity'] = {
            'gate': 'tests/test_gpu_fast_variant.py: culled == all-pairs bit for bit; error quantiles (p50..p99.9) against the reference\'s '
                    'own kernels within 4x the same quantiles of the spread of the reference\'s two builds (contraction off / on)',
            'inside_reference_spread': name not in t.get('fast_outside_spread', {}),
            'elements_outside_elementwise_bracket': {k: '%d of %d' % (v['violations'], v['n']) for k, v in br.items()},
            'deviation_from_reference_kernels': t.get('fast', {}).get(name),
            'note': 'does NOT reproduce the reference\'s rounding (the default build does: rgba bit-identical to the reference\'s kernels at '
                    'C2 / C4); the reference\'s own contracted build deviates from its uncontracted one as much as this variant does',
            'table_kernel_sha': t.get('meta', {}).get('kernel_sha')}
    except Exception as e:
        out['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def other_launch_extra(args, wl, dist, dev, used):
    """The launch mode the headline did NOT use, measured on the same workload: the reference's callers are eager Python
    loops, the headline usually replays a captured HIP graph."""
    other = 'eager' if used == 'graph' else 'graph'
    a = argparse.Namespace(**dict(vars(args), launch=other, steps=max(10, args.steps), warmup=3))
    m = measure(a, wl, dist, dev)
    return other, {'value': wl.global_batch * a.steps / m['elapsed'], 'unit': 'frames/s', 'ms_per_step': m['elapsed'] / a.steps * 1e3,
                   'launch': m['launch'], 'steps': a.steps}


def caller_shapes_extra(dev, steps=30):
    """What the reference's own scripts render, through GenDR.forward, EAGER (they are Python loops), on the 1280-face
    icosphere at 64^2:
      opt_shape       experiments/opt_shape.py:134-159,289-303: 24 views; soft renderer (logistic, probabilistic, hard RGB,
                      dist_eps 100, sigma 1e-2) forward + backward of the silhouette, then the hard renderer (dist_func 0,
                      aggr_alpha_func 0, dist_eps 1) forward under no_grad
      reconstruction  experiments/train_reconstruction.py:181-196,226-231,506,518,557: 4 x 64 = 256 views, uniform,
                      tau = 10^-1.5, probabilistic, hard RGB, dist_eps 300, forward + backward of the silhouettes
    ms per step with eager launches and with the same step replayed from a HIP graph; their ratio is the share of the eager
    step the host (Python, autograd, 6-7 launches per render) accounts for."""
    import torch
    import gendr_amd
    from gendr_amd.synthetic import benchmark_scene
    out = {}

    def timed(fn, n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    def both(step, params):
        for _ in range(5):
            step()
        eager = min(timed(step, steps), timed(step, steps))
        replay = None
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            for p in params:
                p.grad = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                step()
            g.replay()
            replay = min(timed(g.replay, steps), timed(g.replay, steps))
        except Exception as e:
            sys.stderr.write('bench: caller-shape graph capture failed (%s)\n' % (e,))
        return eager, replay

    class M(object):
        pass

    # opt_shape.py
    fv, tex = benchmark_scene(24, device=dev)
    fv.requires_grad_(True)
    soft = gendr_amd.GenDR(image_size=64, dist_func='logistic', dist_scale=1e-2, dist_squared=False, dist_shape=0., dist_shift=0.,
                           dist_eps=100, aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=0., aggr_rgb_func='hard')
    hard = gendr_amd.GenDR(image_size=64, dist_func=0, dist_scale=1e-4, dist_squared=True, dist_shape=0., dist_shift=0., dist_eps=1,
                           aggr_alpha_func=0, aggr_alpha_t_conorm_p=0., aggr_rgb_func='hard')
    mesh = M()
    mesh.face_vertices, mesh.face_textures = fv, tex
    target = torch.rand(24, 64, 64, device=dev)

    def step_opt():
        fv.grad = None
        sil = soft(mesh)[:, 3]
        ((sil - target) ** 2).mean().backward()
        with torch.no_grad():
            hard(mesh)[:, 3]

    e, r = both(step_opt, [fv])
    out['opt_shape_64x64_b24_soft_fwd_bwd_plus_hard_fwd'] = {
        'eager_ms_per_step': e, 'graph_replay_ms_per_step': r, 'frames_per_s_eager': 24 / e * 1e3,
        'host_bound_share_of_eager_step': (None if r is None else max(0.0, 1.0 - r / e))}

    # train_reconstruction.py
    fv2, tex2 = benchmark_scene(256, device=dev)
    fv2.requires_grad_(True)
    rec = gendr_amd.GenDR(image_size=64, dist_func='uniform', dist_scale=10 ** -1.5, dist_squared=False, dist_shape=0, dist_shift=0,
                          dist_eps=300., aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=0, aggr_rgb_func='hard')
    mesh2 = M()
    mesh2.face_vertices, mesh2.face_textures = fv2, tex2
    target2 = torch.rand(256, 64, 64, device=dev)

    def step_rec():
        fv2.grad = None
        sil = rec(mesh2)[:, 3]
        ((sil - target2) ** 2).mean().backward()

    e, r = both(step_rec, [fv2])
    out['train_reconstruction_64x64_b256_dist_eps_300_fwd_bwd'] = {
        'eager_ms_per_step': e, 'graph_replay_ms_per_step': r, 'frames_per_s_eager': 256 / e * 1e3,
        'host_bound_share_of_eager_step': (None if r is None else max(0.0, 1.0 - r / e))}
    out['note'] = ('through gendr_amd.GenDR.forward with eager launches, as the reference\'s scripts run; the step includes the loss '
                   '(two tensor ops) and autograd; graph_replay = the same step captured once and replayed: what the GPU alone needs')
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(1, args.gpus) and 'WORLD_SIZE' in os.environ:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.no_cull:
        os.environ['GENDR_CULL'] = '0'

    dev = None
    oversub = os.environ.get('GENDR_BENCH_OVERSUBSCRIBE') == '1'      # several ranks on one GPU (1-GPU boxes, tests)
    if not args.stub:
        ndev = torch.cuda.device_count()
        if local_rank >= ndev and not oversub:
            raise SystemExit('bench.py: rank %d has no GPU (%d visible); GENDR_BENCH_OVERSUBSCRIBE=1 shares GPUs' % (local_rank, ndev))
        torch.cuda.set_device(local_rank % max(ndev, 1))
        dev = torch.device('cuda', local_rank % max(ndev, 1))
    dist = None
    backend = args.backend or ('gloo' if (args.stub or oversub) else 'nccl')   # RCCL refuses two ranks on one device
    if world > 1 or os.environ.get('GENDR_BENCH_FORCE_DIST') == '1':          # the env var exercises the N>1 calls on one GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world, (dist.get_world_size(), world)

    if not args.stub:
        from gendr_amd import build
        build.build()

    cfg = dict(CONFIGS[args.config])
    if cfg.get('gather'):
        args.scaling = 'strong'          # BASELINE config 4 is DEFINED by its global batch: 256 views over the ranks
    wl = Workload(args, cfg, rank, world, dev, args.scaling)
    m = measure(args, wl, dist, dev)
    extra = {}
    if world > 1 and not args.no_extra and not cfg.get('gather'):
        other = 'weak' if args.scaling == 'strong' else 'strong'
        wl_o = Workload(args, cfg, rank, world, dev, other)
        mo = measure(args, wl_o, dist, dev)
        extra[other] = {'value': wl_o.global_batch * args.steps / mo['elapsed'], 'unit': 'frames/s',
                        'global_batch': wl_o.global_batch, 'ms_per_step': mo['elapsed'] / args.steps * 1e3,
                        'scaling': other, 'launch': mo['launch']}
        del wl_o

    if rank == 0:
        isz, nf, T, B = wl.isz, wl.nf, wl.T, wl.B
        P = isz * isz
        fwd_b, bwd_b = algorithmic_bytes(P, nf, T)
        opts = wl.opts
        elapsed = m['elapsed']
        out = {
            'metric': 'soft_rasterize fwd+bwd frames/s @%d^2, %d faces, batch %d%s' % (isz, nf, (args.batch or cfg['batch']), ' per GPU' if (args.scaling == 'weak' and world > 1) else ''),
            'value': wl.global_batch * args.steps / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s: %d-face icosphere variant, %dx%d, %s, global batch %d (%d on rank 0), T=%d, double_side=%s, dist_eps=1e4%s'
                                   % (args.config.upper(), nf, isz, isz,
                                      '/'.join(str(opts[k]) for k in ('dist_func', 'aggr_alpha_func', 'aggr_rgb_func')),
                                      wl.global_batch, B, T, opts['double_side'],
                                      '; step = render -> all-gather of views -> coupled loss -> backward (reduce-scatter)' if cfg.get('gather') else ''),
                       'global_batch': wl.global_batch,
                       'parallelism': ('batch-sharded x%d, ' % world) + ('RCCL all-gather / reduce-scatter of views' if cfg.get('gather') and world > 1
                                                                          else 'no data-path collective'),
                       'backend': (backend if dist is not None else None),
                       'launch': m['launch'],
                       'cull': os.environ.get('GENDR_CULL', '1') != '0',
                       # the build variant that ran (gendr_amd/build.py): 'default' is the shipped library -- since round 5 within a
                       # flat 1e-5 of the reference's own kernels on every BASELINE configuration (tests/test_gpu_reference_pin.py,
                       # tests/golden/reference/pin_table.json: no exception rows)
                       'variant': _active_variant()},
        }
        if m['bwd_ms'] is not None:
            # The longest single kernel is the backward render kernel; the events around the backward native call
            # bracket exactly that launch (plus, since round 2, nothing else: the gradient zero fill happens before).
            dom = 'render_backward_kernel'
            dom_bytes = bwd_b * B
            dom_ms = m['bwd_ms']
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
            # Profiler figures of the same kernels: HBM traffic (separate rocprofv3 --pmc passes of FETCH_SIZE / WRITE_SIZE) and
            # the VALU occupation (SQ_ACTIVE_INST_VALU), read from profiles/pmc_<config>.json -- but only if that file was
            # collected on THESE kernel sources (it carries their hash): a stale file yields null, not a number.
            traffic, valu_busy, source, lane_frac, valu_ratio = None, None, None, None, None
            pmc_path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % args.config)
            if os.path.exists(pmc_path):
                try:
                    from gendr_amd import build as _b
                    pmc = json.load(open(pmc_path))
                    if pmc.get('kernel_sha') == _b.source_sha():
                        traffic = pmc.get('hbm_bytes_per_launch', {}).get(dom)
                        # (calibrated against a pure-VALU kernel of known occupation collected in the same profile run, when the file
                        # has it: tools/micro/valucal.hip)
                        # the figure relative to an all-FMA kernel EXCEEDS 1 for kernels with transcendental / f64 / cross-lane
                        # instructions (the counter sums per-wave execution cycles, which overlap across pipes) -- it is an issue-rate
                        # ratio, not a saturating fraction: reported under its own name, and `valu_busy` is that ratio clamped to 1
                        valu_ratio = pmc.get('valu_issue_vs_fma_kernel', pmc.get('valu_busy_calibrated', {})).get(dom)
                        valu_busy = min(1.0, valu_ratio) if valu_ratio is not None else pmc.get('valu_busy', {}).get(dom)
                        lane_frac = pmc.get('useful_lane_frac', {}).get(dom)
                        # traffic is quoted PER LAUNCH of this line's batch: the counter passes record the batch they ran at
                        # (C4 / C5 counters of one batch were set beside the algorithmic bytes of another); a pass at
                        # another batch is scaled linearly (the traffic is per frame to a few percent) and the line says so
                        tb = pmc.get('traffic_batch')
                        scaled = ''
                        if traffic is not None and tb and tb != B:
                            traffic = traffic * B / tb
                            scaled = '; traffic measured at batch %d and scaled to this line\'s %d frames on rank 0' % (tb, B)
                        source = ('profiles/pmc_%s.json (kernel sources %s): rocprofv3 --pmc passes (profiles/run_all.sh): traffic = '
                                  '2*FETCH_SIZE+WRITE_SIZE of bench.py at batch %s%s; valu_issue_vs_fma_kernel = (SQ_ACTIVE_INST_VALU / '
                                  'SQ_BUSY_CYCLES) relative to tools/micro/valucal.hip in the same profile run -- a ratio that exceeds 1 for f64 / transcendental '
                                  'mixes, valu_busy = that clamped to 1 -- tools/kbench.py at batch %s; not re-measured in this run'
                                  % (args.config, pmc['kernel_sha'], tb, scaled, pmc.get('sq_batch')))
                    else:
                        source = ('profiles/pmc_%s.json was collected on other kernel sources (%s, now %s): traffic and valu_busy withheld'
                                  % (args.config, pmc.get('kernel_sha'), _b.source_sha()))
                except Exception:
                    traffic = None
            out['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'valu_busy': valu_busy,
                               'valu_issue_vs_fma_kernel': valu_ratio,
                               # share of the issued vector lane-slots that did work: SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)
                               'useful_lane_frac': lane_frac, 'traffic_source': source,
                               'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_ms': dom_ms,
                               'note': 'VALU-bound path (SURVEY.md H2: report the VALU occupation beside the HBM fraction); '
                                       'whole-op fraction on rank 0 = %.4f'
                                       % ((fwd_b + bwd_b) * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS)}
            out['kernel_ms'] = {'forward_phase': m['fwd_ms'], 'backward_phase': m['bwd_ms'], 'event_samples': m['n_events']}
        if m['coll_ms'] is not None:
            out['collective'] = {'ms_per_step': m['coll_ms'], 'share_of_step': m['coll_ms'] / (elapsed / args.steps * 1e3),
                                 'what': 'all_gather_into_tensor of [%d,4,%d,%d] f32 views + reduce_scatter_tensor of their gradients'
                                         % (wl.global_batch, isz, isz)}
        if extra:
            out['extra'] = extra
        if world == 1 and not args.stub and not args.no_extra and not cfg.get('gather'):
            # "also report dist_eps = 300" (the training default, train_reconstruction.py:518) -- same
            # workload, a short second measurement
            try:
                cfg300 = dict(cfg, opts=dict(cfg['opts'], dist_eps=300.0))
                wl300 = Workload(args, cfg300, rank, world, dev, args.scaling)
                a300 = argparse.Namespace(**dict(vars(args), steps=max(5, args.steps // 3), warmup=2))
                m300 = measure(a300, wl300, dist, dev)
                extra['dist_eps_300'] = {'value': wl300.global_batch * a300.steps / m300['elapsed'], 'unit': 'frames/s',
                                         'ms_per_step': m300['elapsed'] / a300.steps * 1e3, 'steps': a300.steps}
                del wl300
            except Exception as e:
                extra['dist_eps_300'] = {'error': '%s: %s' % (type(e).__name__, e)}
            try:
                other, res = other_launch_extra(args, wl, dist, dev, m['launch'])
                extra[other] = res
            except Exception as e:
                extra['other_launch'] = {'error': '%s: %s' % (type(e).__name__, e)}
            try:
                extra['fast_variant'] = fast_variant_extra(args, cfg, rank, world, dev, dist)
            except Exception as e:
                extra['fast_variant'] = {'error': '%s: %s' % (type(e).__name__, e)}
            if args.config == 'c2':
                try:
                    extra['caller_shapes'] = caller_shapes_extra(dev)
                except Exception as e:
                    extra['caller_shapes'] = {'error': '%s: %s' % (type(e).__name__, e)}
            sweep = next((q for q in (os.path.join(ROOT, 'profiles', 'r%02d_%s_batch_sweep.json' % (r, args.config)) for r in (6, 5, 4, 3))
                          if os.path.exists(q)), '')
            if os.path.exists(sweep):
                try:
                    from gendr_amd import build as _b
                    sw = json.load(open(sweep))
                    extra['strong_projection'] = {
                        'speedup_at_n_gpus': sw['strong_projection'], 'ms_per_step_at_batch': sw['batches'],
                        'current_kernels': sw.get('kernel_sha') == _b.source_sha(),
                        'note': 'PROJECTION, not a measurement: t(batch 64) / t(batch 64 / N) of this op on ONE MI355X '
                                '(profiles/%s, tools/batch_sweep.py); no multi-GPU run is behind it' % os.path.basename(sweep)}
                except Exception:
                    pass
            if extra:
                out['extra'] = extra
        if world == 1 and not args.no_cpu_baseline and not args.stub:
            nb = min(B, 64)
            tb = cpu_baseline_torch(cfg, wl.fv_cpu[:nb], wl.tex_cpu[:nb])
            oc = cpu_baseline_oracle(cfg, wl.fv_cpu[:nb], wl.tex_cpu[:nb])
            if tb is not None and 'error' not in tb:
                out['cpu_baseline'] = tb
                out['cpu_baseline_oracle'] = oc
            else:
                oc['note'] = ('pure-PyTorch baseline failed (%s); C/OpenMP oracle instead' % tb['error']) if tb else \
                             'oracle/torch_ref.py does not cover this option set; C/OpenMP oracle instead'
                out['cpu_baseline'] = oc
            rk = gpu_baseline_reference_kernels(cfg, wl.fv_cpu[:nb], wl.tex_cpu[:nb])
            if rk is not None:
                out['reference_kernels_baseline'] = rk
                if 'value' in rk:
                    rk['speedup_of_this_repo'] = out['value'] / rk['value']
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
