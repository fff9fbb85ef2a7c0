"""Summarises a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch)."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name']
    short = name.split('(')[0].replace('void ', '')[:60]
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    key = (short, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key)
        dur[short].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        acc[short]['_vgpr'] = [float(r['VGPR_Count'])]; acc[short]['_lds'] = [float(r['LDS_Block_Size'])]; acc[short]['_scratch'] = [float(r['Scratch_Size'])]
for k, v in acc.items():
    if 'gendr' not in k: continue
    d = dur[k]
    print('%s  dispatches=%d  dur_us(min/med)=%.1f/%.1f  vgpr=%d lds=%d scratch=%d' % (k, len(d), min(d), sorted(d)[len(d)//2], v['_vgpr'][0], v['_lds'][0], v['_scratch'][0]))
    for c, vals in sorted(v.items()):
        if c.startswith('_'): continue
        print('     %-24s mean %.4g  (per dispatch list: %s)' % (c, sum(vals)/len(vals), ' '.join('%.3g' % x for x in vals[:8])))
