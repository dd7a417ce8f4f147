#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per kernel and counter, print the kernel-trace stats, and write
OUTDIR/TAG_counters.json: the per-read figures bench.py reports with their provenance (HBM bytes with the
gfx950 FETCH_SIZE correction of /opt/skills/guides/MI355X_MICROARCH.md, vector instructions, VALU busy share).
usage: pmc_sum.py OUTDIR TAG"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or '')


def main():
    out, tag = sys.argv[1], sys.argv[2]
    stats = {}
    for path in sorted(glob.glob(os.path.join(out, tag + '_kt', '**', '*kernel_stats.csv'), recursive=True)):
        print('# ' + path)
        for i, row in enumerate(csv.reader(open(path))):
            if i < 9:
                print(','.join(c[:76] for c in row))
            if i and short(row[0]):
                stats[short(row[0])] = (int(row[1]), float(row[3]))     # calls, average ns
    sums = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    meta = {}
    for path in sorted(glob.glob(os.path.join(out, tag + '_pmc*', '**', '*counter_collection.csv'), recursive=True)):
        for row in csv.DictReader(open(path)):
            k = short(row['Kernel_Name'])
            if not k or not re.match(r'(align|ell|plan|consensus|posterior|expected|lane3)', k):
                continue
            sums[(k, row['Counter_Name'])] += float(row['Counter_Value'])
            launches[(k, row['Counter_Name'])] += 1
            meta[k] = (row['Grid_Size'], row['Workgroup_Size'], row['LDS_Block_Size'], row['VGPR_Count'], row['SGPR_Count'])
    for k, m in sorted(meta.items()):
        print('# %s grid=%s wg=%s lds=%s vgpr=%s sgpr=%s' % ((k,) + m))
    for (k, c), v in sorted(sums.items()):
        print('%-44s %-24s launches=%d sum=%.6g per_launch=%.6g' % (k, c, launches[(k, c)], v, v / launches[(k, c)]))

    # per-read figures of the dominant kernels (one bench step = one batch; the pmc passes ran --steps 1)
    bench = {}
    try:
        bench = json.loads(open(os.path.join(out, tag + '_bench_under_rocprof.json')).read().strip().splitlines()[-1])
    except Exception:
        pass
    cfg = bench.get('config', {})
    n = cfg.get('reads_per_gpu_per_step')
    if not n:
        return
    dom = [k for k in meta if re.match(r'(align3_kernel|ell_kernel)', k)]
    per = lambda c: sum(sums[(k, c)] / max(launches[(k, c)], 1) for k in dom if (k, c) in sums)
    fetch_kb, write_kb = per('FETCH_SIZE'), per('WRITE_SIZE')
    valu, gui = per('SQ_INSTS_VALU'), max([sums[(k, 'GRBM_GUI_ACTIVE')] / max(launches[(k, 'GRBM_GUI_ACTIVE')], 1)
                                            for k in dom if (k, 'GRBM_GUI_ACTIVE') in sums] or [0])
    act = per('SQ_ACTIVE_INST_VALU')
    gui_all = per('GRBM_GUI_ACTIVE')
    j = {'comment': 'per-read figures of %s from rocprofv3 --pmc passes over one step of `bench.py %s` (tools/profile_round.sh %s); '
                    'FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950, '
                    'WRITE_SIZE exact; valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)'
                    % (' + '.join(sorted(dom)), '--workload ' + cfg.get('workload', '?'), tag),
         'workload': cfg.get('workload'), 'reads_per_launch': n, 'kernels': sorted(dom),
         'kernel_avg_ms': {k: stats[k][1] / 1e6 for k in dom if k in stats},
         'FETCH_SIZE_KB_raw': fetch_kb, 'WRITE_SIZE_KB': write_kb, 'fetch_correction': 2.0,
         'hbm_bytes_per_read': (2.0 * fetch_kb + write_kb) * 1024.0 / n if (fetch_kb or write_kb) else None,
         'valu_insts_per_read': valu / n if valu else None,
         'salu_insts_per_read': per('SQ_INSTS_SALU') / n if per('SQ_INSTS_SALU') else None,
         'valu_busy_frac': (act * 4.0) / (gui_all / 8.0 * 1024.0) if act and gui_all else None}
    j['kernel_source_sha1'] = kernel_source_hash(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    json.dump(j, open(os.path.join(out, tag + '_counters.json'), 'w'), indent=1)
    print('# wrote %s_counters.json: %s' % (tag, {k: j[k] for k in ('hbm_bytes_per_read', 'valu_insts_per_read', 'valu_busy_frac')}))



def kernel_source_hash(root):
    """sha1 over the HIP sources and headers of the library with comments and white space removed (the code the
    counters describe; editing a comment does not make a profile stale)."""
    import glob, hashlib, re
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(root, 'nadavca_amd', 'csrc', '*.hip')) +
                       glob.glob(os.path.join(root, 'nadavca_amd', 'csrc', '*.h')) +
                       glob.glob(os.path.join(root, 'include', '*.h'))):
        text = open(path, 'r', encoding='utf-8', errors='replace').read()
        text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
        text = re.sub(r'//[^\n]*', ' ', text)
        text = re.sub(r'\s+', ' ', text)
        h.update(os.path.basename(path).encode())
        h.update(text.encode())
    return h.hexdigest()


if __name__ == '__main__':
    main()
