"""Summarise rocprofv3 --pmc runs (one counter per run) per kernel and refresh profiles/pmc_traffic.json:
    python tools/pmc_summary.py --tag r02 --workload cfg3 DIR_FETCH DIR_WRITE [--stats DIR_STATS]
Counter values of FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read -> a x2-corrected column is printed next to the raw one and used for roofline.traffic."""
import argparse, csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
SHAPES = {'cfg2': (84010, 1), 'cfg3': (200000, 1), 'cfg4': (88200, 1), 'cfg5': (100000, 8), 'cfg3_sqrt': (200000, 1),
          'cfg5_fill': (12500, 32), 'cfg2_batch': (4000, 128), 'cfg3_batch': (4000, 256)}


def category(name):
    """The timing category (nagp_plan_timings: filter, filter_lin, gain, scan, epsite, reduce) a kernel belongs to -- bench.py looks the traffic
    of its dominant CATEGORY up, whatever instantiations happened to serve it (round 4: a hard-coded instantiation name went stale when the
    default kernel changed)."""
    import re
    if 'rts_gain' in name:
        return 'gain'
    if any(k in name for k in ('rts_compose', 'rts_boundary', 'rts_apply', 'rts_big', 'ihgp_aff', 'ihgp_scan')):
        return 'scan'
    if 'ep_site' in name:
        return 'epsite'
    if 'sum_kernel' in name:
        return 'reduce'
    if 'gf_filter_lin' in name:
        return 'filter_lin'
    m = re.search(r'gf_filter_kernel<\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)', name)
    if m:
        return 'filter_lin' if (int(m.group(2)) == 0 and int(m.group(3)) < 0) else 'filter'       # MEAS == 0, MV = -1: the fixed-site launches
    if any(k in name for k in ('gf_adf8', 'ihgp_adf', 'ihgp_filter', 'ekf_grad', 'iekf')):
        return 'filter'
    return 'other'


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
    cats = defaultdict(lambda: dict(fetch=0.0, write=0.0, kernels=set()))
    for (c, nm), (v, n) in tot.items():
        e = cats[category(nm)]
        e['fetch' if c == 'FETCH_SIZE' else 'write'] += v
        e['kernels'].add(nm.split('(')[0][:90])
    jp = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(jp) as fh:
            js = json.load(fh)
    except (OSError, ValueError):
        js = {}
    js['_comment'] = ('HBM bytes per timing category (filter, filter_lin, gain, scan, epsite: tools/pmc_summary.py category()), summed over the dispatches of ONE execute (bench.py divides by its launches per execute), from rocprofv3 --pmc '
                      'FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_run.sh); FETCH_SIZE x2 (gfx950); bench.py reports roofline.traffic only when source_hash matches the loaded library')
    js[a.workload] = dict(by_category={k: dict(fetch_bytes_per_execute=e['fetch'], write_bytes_per_execute=e['write'], kernels=sorted(e['kernels'])) for k, e in sorted(cats.items())},
                          source=os.path.relpath(out, ROOT), source_hash=L.source_hash(), T=SHAPES[a.workload][0], segments=SHAPES[a.workload][1])
    with open(jp, 'w') as fh:
        json.dump(js, fh, indent=1)
    print('profiles/pmc_traffic.json <-', a.workload, {k: 'fetch %.3f GB write %.3f GB' % (e['fetch'] / 1e9, e['write'] / 1e9) for k, e in cats.items()})


if __name__ == '__main__':
    main()
