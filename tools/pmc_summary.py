"""Summarise rocprofv3 --pmc runs (one counter per run) per kernel: python tools/pmc_summary.py DIR [DIR...]
Counter values of FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE under-reports
wide coalesced reads by 2x -> a corrected column is printed next to the raw one."""
import csv, glob, os, sys
from collections import defaultdict


def main(dirs):
    for d in dirs:
        files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        for f in files:
            acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get('Kernel_Name', row.get('Kernel Name', '?')).split('(')[0]
                    c = row.get('Counter_Name', '?'); v = float(row.get('Counter_Value', 0))
                    a = acc[c][name]; a[0] += v; a[1].add(row.get('Dispatch_Id', row.get('Dispatch Id', '')))
            for c, per in acc.items():
                print(c)
                for name, (tot, ids) in sorted(per.items(), key=lambda kv: -kv[1][0]):
                    n = max(len(ids), 1); gb = tot / n * 1024 / 1e9
                    extra = '  (x2 corrected: %.3f GB)' % (2 * gb) if c == 'FETCH_SIZE' else ''
                    print('  %-60s calls %3d  per launch %.3f GB%s' % (name[:60], n, gb, extra))


if __name__ == '__main__':
    main(sys.argv[1:])
