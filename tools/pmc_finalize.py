"""Stamps a traffic summary (tools/traffic_summary.py) with the kernel sources' hash and adds the VALU occupation of every
kernel:   python tools/pmc_finalize.py <pmc_cfg.json> <dir of the SQ counter pass with SQ_ACTIVE_INST_VALU> <kernel_stats.csv> [sq batch] [dir of the calibration pass]
valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles, MI355X_MICROARCH.md) * 4 / (1024 SIMDs * average kernel duration * 2.4 GHz);
valu_issue_vs_fma_kernel (round 5: valu_busy_calibrated) = that divided by what the same expression reads for tools/micro/valucal.hip -- a kernel whose vector ALUs are
occupied 100 % by construction -- under the same counters in the same gpurun call (the raw figure read 1.17 at
saturation); useful_lane_frac = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU): the share of issued vector lane-slots that worked."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gendr_amd import build

def short(name):
    name = name.split('(')[0].replace('void ', '').split('<')[0].replace('gendr::', '')
    return re.sub(r'_w[0-9a-z]$', '', name)

path, sq_dir, stats = sys.argv[1:4]
d = json.load(open(path))
acc = collections.defaultdict(list)
gui = collections.defaultdict(list)
thr = collections.defaultdict(list)
busy = collections.defaultdict(list)
for f in glob.glob(os.path.join(sq_dir, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'SQ_ACTIVE_INST_VALU' and 'gendr' in r['Kernel_Name']:
            acc[short(r['Kernel_Name'])].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
        if r['Counter_Name'] == 'SQ_BUSY_CYCLES' and 'gendr' in r['Kernel_Name']:
            busy[short(r['Kernel_Name'])].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
        if r['Counter_Name'] == 'SQ_THREAD_CYCLES_VALU' and 'gendr' in r['Kernel_Name']:
            thr[short(r['Kernel_Name'])].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and 'gendr' in r['Kernel_Name']:
            gui[short(r['Kernel_Name'])].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
quad = {}
for k, v in acc.items():
    g = max(x[0] for x in v)
    vals = [x[1] for x in v if x[0] == g]
    quad[k] = sum(vals) / len(vals)
# kernel durations of the SAME run the counters come from (its kernel trace: same batch, same launches), largest grid only;
# the bench's kernel_stats.csv is the fallback
dur = {}
tr = collections.defaultdict(list)
for f in glob.glob(os.path.join(sq_dir, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gendr' in r['Kernel_Name']:
            g = int(r.get('Grid_Size', 0) or r.get('Grid_Size_X', 0) or 0)
            tr[short(r['Kernel_Name'])].append((g, (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3))
for k, v in tr.items():
    g = max(x[0] for x in v)
    vals = sorted(x[1] for x in v if x[0] == g)
    dur[k] = vals[len(vals) // 2]
if not dur:
    for r in csv.DictReader(open(stats)):
        if 'gendr' in r['Name']:
            dur[short(r['Name'])] = float(r['AverageNs']) * 1e-3
d['kernel_sha'] = build.source_sha()
d['valu_active_quad_cycles'] = quad
d['avg_kernel_us'] = dur
# GPU cycles of the launch from the SAME pass (GRBM_GUI_ACTIVE: no assumption about the clock the run happened at -- round 3
# divided by duration * 2.4 GHz and printed occupations above 1 for runs at a lower clock); fallback: the nominal clock
cycles = {}
for k, v in gui.items():
    g = max(x[0] for x in v)
    vals = [x[1] for x in v if x[0] == g]
    cycles[k] = sum(vals) / len(vals)
d['gpu_cycles'] = cycles
d['valu_busy'] = {k: quad[k] * 4 / (1024 * dur[k] * 1e-6 * 2.4e9) for k in quad if k in dur and dur[k] > 0}
# GRBM_GUI_ACTIVE is summed over the 8 XCDs and covers the whole dispatch (start-up, drain, overlap with its neighbours): for
# the long render kernels it gives a clock-independent second opinion, for the short geometry kernels it over-counts
d['valu_busy_by_gpu_cycles'] = {k: quad[k] * 4 / (1024 * cycles[k] / 8.0) for k in quad if cycles.get(k)}
d['valu_busy_note'] = ('valu_busy = SQ_ACTIVE_INST_VALU * 4 cycles / (1024 SIMDs * median kernel duration IN THE SAME counter pass * 2.4 GHz '
                       'nominal clock): +-10 % (the clock of the pass is not recorded; values a few percent above 1 mean "fully occupied"); '
                       'valu_busy_by_gpu_cycles = the same over GRBM_GUI_ACTIVE / 8 XCDs of the same launch (no clock assumption, but the '
                       'counter spans the whole dispatch); the counter pass runs tools/kbench.py at the batch recorded in sq_batch, the traffic '
                       'passes run bench.py at the batch recorded in traffic_batch')
lanes = {}
for k, v in thr.items():
    g = max(x[0] for x in v)
    vals = [x[1] for x in v if x[0] == g]
    if quad.get(k):
        lanes[k] = (sum(vals) / len(vals)) / (64.0 * quad[k])
d['useful_lane_frac'] = lanes
if len(sys.argv) > 4:
    d['sq_batch'] = int(sys.argv[4])
# calibration pass (tools/micro/valucal.hip under the same counters): what the occupation expression reads at 100 %
if len(sys.argv) > 5:
    cq, cg, ct, cd, cb = [], [], [], [], []
    for f in glob.glob(os.path.join(sys.argv[5], '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'valu_calibration_kernel' not in r['Kernel_Name']:
                continue
            {'SQ_ACTIVE_INST_VALU': cq, 'GRBM_GUI_ACTIVE': cg, 'SQ_THREAD_CYCLES_VALU': ct, 'SQ_BUSY_CYCLES': cb}.get(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(sys.argv[5], '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'valu_calibration_kernel' in r['Kernel_Name']:
                cd.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    if cq and cd:
        med = lambda v: sorted(v)[len(v) // 2]
        f_nom = med(cq) * 4 / (1024 * med(cd) * 1e-6 * 2.4e9)
        f_gui = med(cq) * 4 / (1024 * med(cg) / 8.0) if cg else None
        d['valu_calibration'] = {'kernel': 'tools/micro/valucal.hip (8 waves per SIMD of independent v_fma_f32: occupation 1.00 by construction)',
                                 'reads_by_nominal_clock': f_nom, 'reads_by_gpu_cycles': f_gui,
                                 'lane_frac_reads': (med(ct) / (64.0 * med(cq))) if ct else None, 'duration_us': med(cd)}
        d['valu_calibration']['effective_clock_ghz_of_the_calibration_pass'] = 2.4 * f_nom
        # The occupation as a FRACTION: vector-instruction quad-cycles over the SQ's busy quad-cycles of the SAME
        # dispatch -- both in the shader-clock domain, so the clock of the pass (1.9 GHz for the calibration kernel under counters, up
        # to 2.4 GHz for sparser kernels: MI355X_MICROARCH.md) drops out -- normalised by the same ratio of the calibration kernel,
        # whose vector pipes are occupied 100 % by construction (that fixes the counters' aggregation over XCDs / SEs).
        if cb:
            r_cal = med(cq) / med(cb)
            d['valu_calibration']['active_over_busy_reads'] = r_cal
            vb = {}
            for k, v in busy.items():
                g = max(x[0] for x in v)
                vals = [x[1] for x in v if x[0] == g]
                if quad.get(k) and vals:
                    vb[k] = (quad[k] / (sum(vals) / len(vals))) / r_cal
            d['valu_issue_vs_fma_kernel'] = vb       # (round 5 called this valu_busy_calibrated; it exceeds 1 and is a ratio, not a fraction: ADVICE r5)
        d['valu_busy_note'] += ('; valu_issue_vs_fma_kernel = (SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of the same dispatch) / (the same ratio of '
                                'tools/micro/valucal.hip, occupied 100 %% by construction, collected in the same gpurun call): a RATIO that exceeds 1 for kernels with '
                                'transcendental / f64 / cross-lane instructions (the counter sums per-wave execution cycles, which overlap across pipes), '
                                'independent of the clock the pass ran at -- the raw valu_busy assumes 2.4 GHz, and the calibration kernel '
                                'itself reads %.2f by that assumption because profiled dense-VALU passes clock at %.2f GHz' % (f_nom, 2.4 * f_nom))
json.dump(d, open(path, 'w'), indent=1)
print(json.dumps({k: round(v, 3) for k, v in d['valu_busy'].items()}), d['kernel_sha'])
