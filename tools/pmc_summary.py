"""Summary of a rocprofv3 --pmc run (counter_collection csv): per kernel and counter the mean over the dispatches of the
LARGEST grid (the bench kernels; warm-up launches of other sizes are dropped).   python tools/pmc_summary.py <dir> [substr ...]"""
import collections, csv, glob, os, sys

def main():
    d = sys.argv[1]; want = sys.argv[2:]
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name') or r.get('Kernel-Name')
            if want and not any(w in k for w in want):
                continue
            acc[k][r['Counter_Name']].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
    for k in sorted(acc):
        print(k[:110])
        for c in sorted(acc[k]):
            v = acc[k][c]
            g = max(x[0] for x in v)
            vals = [x[1] for x in v if x[0] == g]
            print('    %-24s mean %.4g  (n=%d, grid %d)' % (c, sum(vals) / len(vals), len(vals), g))

main()
