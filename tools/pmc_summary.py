"""Summarise rocprofv3 --pmc runs (one counter per run) per kernel and refresh profiles/pmc_traffic.json:
    python tools/pmc_summary.py --tag r02 --workload cfg3 DIR_FETCH DIR_WRITE [--stats DIR_STATS]
Counter values of FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read -> a x2-corrected column is printed next to the raw one and used for roofline.traffic."""
import argparse, csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
DOMINANT = {'cfg3': ('ihgp_adf8_kernel', 'ihgp_adf_kernel', 'ihgp_filter_kernel'), 'cfg2': ('gf_filter_kernel<1, 0, 3, 256',), 'cfg5': ('gf_filter_kernel<3, 0, 6, 256',),
            'cfg4': ('gf_filter_kernel<1, 1, 0',), 'cfg3_sqrt': ('ihgp_adf8sq_kernel',)}
SHAPES = {'cfg2': (84010, 1), 'cfg3': (200000, 1), 'cfg4': (88200, 1), 'cfg5': (100000, 8), 'cfg3_sqrt': (200000, 1)}


def per_kernel(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', row.get('Kernel Name', '?'))
                c = row.get('Counter_Name', '?'); v = float(row.get('Counter_Value', 0))
                a = acc[c][name]; a[0] += v; a[1].add(row.get('Dispatch_Id', row.get('Dispatch Id', '')))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dirs', nargs='+'); ap.add_argument('--tag', default='r02'); ap.add_argument('--workload', default=None); ap.add_argument('--stats', default=None)
    a = ap.parse_args()
    lines = ['rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv, no trace flags): python3 bench.py --workload %s --steps 1 --warmup 0 '
             '--no-cpu-baseline --extras none' % a.workload,
             'counter unit = KB; totals over ALL dispatches of a kernel in the one execute, and per dispatch; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM)']
    tot = {}
    for d in a.dirs:
        for c, per in per_kernel(d).items():
            lines.append(c)
            for name, (v, ids) in sorted(per.items(), key=lambda kv: -kv[1][0]):
                n = max(len(ids), 1); gb = v * 1024 / 1e9
                k = 2.0 if c == 'FETCH_SIZE' else 1.0
                lines.append('  %-70s dispatches %3d  total %.4f GB%s  per dispatch %.4f GB' % (name[:70], n, gb, ('  (x2: %.4f GB)' % (2 * gb)) if k == 2 else '', gb / n))
                tot[(c, name)] = (k * v * 1024, n)
    text = '\n'.join(lines)
    print(text)
    if not a.workload:
        return
    out = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.txt' % (a.tag, a.workload))
    with open(out, 'w') as fh:
        fh.write(text + '\n')
    if a.stats:
        for f in glob.glob(os.path.join(a.stats, '**', '*kernel_stats.csv'), recursive=True):
            shutil.copy(f, os.path.join(ROOT, 'profiles', '%s_kernel_stats_%s.csv' % (a.tag, a.workload)))
    from nagp import _lib as L
    keys = DOMINANT.get(a.workload, ())
    fetch = sum(v for (c, nm), (v, n) in tot.items() if c == 'FETCH_SIZE' and any(k in nm for k in keys))
    write = sum(v for (c, nm), (v, n) in tot.items() if c == 'WRITE_SIZE' and any(k in nm for k in keys))
    names = sorted({nm for (c, nm) in tot if any(k in nm for k in keys)})
    jp = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(jp) as fh:
            js = json.load(fh)
    except (OSError, ValueError):
        js = {}
    js['_comment'] = ('HBM bytes of the dominant kernel, summed over its dispatches of ONE execute (bench.py divides by its launches per execute), from rocprofv3 --pmc '
                      'FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_run.sh); FETCH_SIZE x2 (gfx950); bench.py reports roofline.traffic only when source_hash matches the loaded library')
    js[a.workload] = dict(kernel=', '.join(names), fetch_bytes_per_execute=fetch, write_bytes_per_execute=write, source=os.path.relpath(out, ROOT),
                          source_hash=L.source_hash(), T=SHAPES[a.workload][0], segments=SHAPES[a.workload][1] if a.workload == 'cfg5' else 1)
    with open(jp, 'w') as fh:
        json.dump(js, fh, indent=1)
    print('profiles/pmc_traffic.json <-', a.workload, 'fetch %.4f GB write %.4f GB' % (fetch / 1e9, write / 1e9))


if __name__ == '__main__':
    main()
