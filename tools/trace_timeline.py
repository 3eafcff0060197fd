"""Timeline summary of one `rocprofv3 --kernel-trace` run of bench.py: which kernels ran when, on which queue, and how much of the
parallel smoother kernels' time lay beside the sequential filter launches (the chunk-pipelined schedule of nagp_api.hip).
    python tools/trace_timeline.py <rocprofv3 output dir> [--bins N]  > profiles/<tag>_pipeline_timeline_<workload>.txt"""
import csv, glob, os, sys


def short(name):
    for key, lab in (('gf_filter_kernel', 'filter'), ('gf_adf8_kernel', 'filter'), ('gf_filter_lin_mfma_kernel', 'filter'), ('rts_gain_kernel', 'gain'), ('rts_gain_mfma_kernel', 'gain'), ('rts_big_phi_kernel', 'compose'), ('rts_compose', 'compose'),
                     ('rts_boundary', 'boundary'), ('rts_apply', 'apply'), ('ep_site_', 'ep_site'), ('sum_kernel', 'reduce')):
        if key in name:
            if key == 'gf_filter_kernel':
                return 'filter'
            return lab
    if 'rts_big_kernel' in name:
        # template arguments <N, MODE>: 0 = compose (C chain), 1 = boundary, 2 = apply
        try:
            mode = int(name.split('rts_big_kernel<')[1].split('>')[0].split(',')[1])
        except Exception:
            mode = -1
        return {0: 'compose', 1: 'boundary', 2: 'apply'}.get(mode, 'big')
    return 'other'


def main():
    d = sys.argv[1]
    bins = int(sys.argv[sys.argv.index('--bins') + 1]) if '--bins' in sys.argv else 40
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')))
    rows.sort()
    if not rows:
        print('no kernel trace found under', d); return
    # the execute = from the first filter launch to the last kernel
    t0 = min(r[0] for r in rows if r[2] == 'filter'); t1 = max(r[1] for r in rows)
    rows = [r for r in rows if r[1] >= t0]
    tot = (t1 - t0) / 1e6
    print('one execute: %.1f ms from the first filter launch to the last kernel; %d kernel dispatches, queues %s' % (tot, len(rows), sorted({r[3] for r in rows})))
    filt = [(a, b) for a, b, n, q in rows if n == 'filter']

    def overlap(a, b):
        return sum(max(0, min(b, fb) - max(a, fa)) for fa, fb in filt)
    print('\n%-10s %8s %12s %14s %16s' % ('kernel', 'launches', 'sum of ms', 'beside filter', 'queues'))
    for lab in ('filter', 'gain', 'compose', 'boundary', 'apply', 'ep_site', 'reduce', 'other'):
        sel = [r for r in rows if r[2] == lab]
        if not sel:
            continue
        sm = sum(b - a for a, b, _, _ in sel) / 1e6
        ov = sum(overlap(a, b) for a, b, _, _ in sel) / 1e6 if lab != 'filter' else 0.0
        print('%-10s %8d %12.1f %13.1f%% %16s' % (lab, len(sel), sm, 100.0 * ov / sm if sm else 0.0, ','.join(sorted({r[3] for r in sel}))))
    # wall-clock union of the non-filter kernels outside the filter launches = what the schedule leaves exposed
    ev = []
    for a, b, n, q in rows:
        if n != 'filter':
            ev.append((a, 1)); ev.append((b, -1))
    fe = []
    for a, b in filt:
        fe.append((a, 1)); fe.append((b, -1))
    pts = sorted(set([t for t, _ in ev] + [t for t, _ in fe]))
    ev.sort(); fe.sort()
    exposed = both = only_f = 0
    ie = jf = 0; ne = nf = 0
    for i in range(len(pts) - 1):
        while ie < len(ev) and ev[ie][0] <= pts[i]:
            ne += ev[ie][1]; ie += 1
        while jf < len(fe) and fe[jf][0] <= pts[i]:
            nf += fe[jf][1]; jf += 1
        dt = pts[i + 1] - pts[i]
        if nf > 0 and ne > 0: both += dt
        elif nf > 0: only_f += dt
        elif ne > 0: exposed += dt
    print('\nwall clock: filter alone %.1f ms | filter with smoother kernels beside it %.1f ms | smoother / EP kernels with no filter running (exposed) %.1f ms | idle %.1f ms'
          % (only_f / 1e6, both / 1e6, exposed / 1e6, tot - (only_f + both + exposed) / 1e6))
    if '--gap' in sys.argv:
        # every dispatch between the end of the first sweep's last filter launch and the start of the next filter launch
        fl = sorted(filt)
        gaps = [(fl[i][1], fl[i + 1][0]) for i in range(len(fl) - 1) if fl[i + 1][0] - fl[i][1] > 2e6]
        if gaps:
            ga, gb = gaps[0]
            print('\nfirst gap between filter launches: %.1f ms; dispatches in it (start after the filter ended, duration, queue):' % ((gb - ga) / 1e6))
            for a, b, n, q in rows:
                if b > ga and a < gb and n != 'filter':
                    print('   %-9s +%8.2f ms  %8.2f ms  q%s' % (n, (a - ga) / 1e6, (b - a) / 1e6, q))
    print('\ntimeline (%d bins of %.1f ms): F = a filter launch is running, letters = kernels that ran in the bin (g gain, c compose, b boundary, a apply, e ep_site)' % (bins, tot / bins))
    w = (t1 - t0) / bins
    for i in range(bins):
        lo, hi = t0 + i * w, t0 + (i + 1) * w
        labs = set()
        for a, b, n, q in rows:
            if a < hi and b > lo:
                labs.add(n)
        line = ('F' if 'filter' in labs else '.') + ' ' + ''.join(ch if lab in labs else ' ' for lab, ch in (('gain', 'g'), ('compose', 'c'), ('boundary', 'b'), ('apply', 'a'), ('ep_site', 'e')))
        print('%8.1f ms  %s' % (i * w / 1e6, line))


if __name__ == '__main__':
    main()
