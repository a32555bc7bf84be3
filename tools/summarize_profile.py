#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01 r01

Writes (next to the raw output, copy them to profiles/):
  <tag>_kernel_stats.csv   per-kernel calls / total / average ns   (from the --stats pass)
  <tag>_pmc_summary.csv    per-kernel average counter values per launch (FETCH_SIZE, WRITE_SIZE, SQ_*)
  <tag>_traffic.json       HBM bytes per launch for each kernel: 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes), the gfx950
                           correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at half their bytes)
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    """conv_gemm_kernel<2,2,1,1,...> style names as bench.py prints them."""
    m = re.match(r'(?:void )?(?:dfl::)?(?:\(anonymous namespace\)::)?(\w+)<([^>]*)>', name)
    if not m:
        return re.sub(r'\(.*', '', name).replace('void ', '').replace('dfl::', '')[:80]
    args = [a.strip() for a in m.group(2).split(',')]
    keep = {'conv_gemm_kernel': 4, 'conv_rows_kernel': 4, 'wgrad_kernel': 5, 'convp_kernel': 4, 'wgradp_kernel': 2}.get(m.group(1), len(args))
    tail = ',k2' if (m.group(1) == 'convp_kernel' and len(args) >= 7 and args[6] == '2') else ''     # two k-groups (512 threads)
    if m.group(1) == 'convq_kernel':          # (anonymous namespace: the demangled name may carry it) -> the wave layout (+ ',p' persistent), as bench.py names it
        return 'convq_kernel<%s%s>' % (','.join(args[1:4]), ',p' if len(args) > 5 and args[5] != '0' else '')     # <CK, WM, WN, KS, AFF, PERS>
    if m.group(1) == 'convn_kernel':          # <CK, NCT, WX, R, AFF, MB, PERS> -> column tiles, waves side by side, rows per wave
        return 'convn_kernel<%s%s>' % (','.join(args[1:4]), ',p' if len(args) > 6 and args[6] in ('true', '1') else '')
    return '%s<%s%s>' % (m.group(1), ','.join(args[:keep]), tail)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(d, 'trace', '**', '*kernel_stats.csv'), recursive=True)
    rows = []
    if stats:
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(stats[0])):
            k = short(r['Name'])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += int(r['Calls'])
            a[1] += float(r['TotalDurationNs'])
        tot = sum(v[1] for v in agg.values())
        with open(os.path.join(d, tag + '_kernel_stats.csv'), 'w') as f:
            f.write('kernel,calls,total_ns,average_ns,percent\n')
            for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s",%d,%.0f,%.1f,%.2f\n' % (k, c, t, t / c, 100 * t / tot))
    pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            a = pmc[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    names = sorted({c for v in pmc.values() for c in v})
    with open(os.path.join(d, tag + '_pmc_summary.csv'), 'w') as f:
        f.write('kernel,launches,' + ','.join(names) + '\n')
        for k, v in sorted(pmc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', [0, 1])[0]):
            n = max(a[1] for a in v.values())
            f.write('"%s",%d,%s\n' % (k, n, ','.join('%.1f' % (v[c][0] / v[c][1]) if c in v else '' for c in names)))
    traffic = {}
    for k, v in pmc.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            fe, wr = v['FETCH_SIZE'][0] / v['FETCH_SIZE'][1], v['WRITE_SIZE'][0] / v['WRITE_SIZE'][1]
            traffic[k] = {'fetch_kb_raw': round(fe, 1), 'write_kb_raw': round(wr, 1),
                          'hbm_bytes_per_launch': round((2 * fe + wr) * 1024)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as _bench                                 # (the hashes bench.py compares with: per kernel, its own source files)
    for k, rec in traffic.items():
        rec['src_sha16'] = _bench.csrc_sha16(k)
    json.dump({'note': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of each kernel in '
                       'bench.py --steps 6 --warmup 2; see tools/summarize_profile.py', 'round': tag.split('_')[0],
               'csrc_sha16': _bench.csrc_sha16(), 'kernels': traffic},
              open(os.path.join(d, tag + '_traffic.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
