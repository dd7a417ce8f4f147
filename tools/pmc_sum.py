#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per kernel and counter, and print the kernel-trace stats.
usage: pmc_sum.py OUTDIR TAG"""
import csv, glob, os, re, sys, collections

def main():
    out, tag = sys.argv[1], sys.argv[2]
    for path in sorted(glob.glob(os.path.join(out, tag + '_kt', '**', '*kernel_stats.csv'), recursive=True)):
        print('# ' + path)
        for i, row in enumerate(csv.reader(open(path))):
            if i < 8:
                print(','.join(c[:70] for c in row))
    sums = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    meta = {}
    for path in sorted(glob.glob(os.path.join(out, tag + '_pmc*', '**', '*counter_collection.csv'), recursive=True)):
        for row in csv.DictReader(open(path)):
            m = re.search(r'(\w+_kernel(<\d+>)?)', row['Kernel_Name'])
            if not m or not re.match(r'(align|ell|plan|consensus|posterior|expected)', m.group(1)):
                continue
            k = m.group(1)
            sums[(k, row['Counter_Name'])] += float(row['Counter_Value'])
            launches[(k, row['Counter_Name'])] += 1
            meta[k] = (row['Grid_Size'], row['Workgroup_Size'], row['LDS_Block_Size'], row['VGPR_Count'], row['SGPR_Count'])
    for k, m in sorted(meta.items()):
        print('# %s grid=%s wg=%s lds=%s vgpr=%s sgpr=%s' % ((k,) + m))
    for (k, c), v in sorted(sums.items()):
        print('%-50s %-24s launches=%d sum=%.6g per_launch=%.6g' % (k, c, launches[(k, c)], v, v / launches[(k, c)]))

if __name__ == '__main__':
    main()
