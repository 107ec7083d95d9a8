import csv, glob, collections, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'mmvq_kernel'
for d in sorted(glob.glob('gpurun_out/pmc_*')):
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            if pat in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        kt = f.replace('counter_collection', 'kernel_trace')
        d_us = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt)) if pat in r['Kernel_Name']]
        print(d, 'dur_us mean %.2f min %.2f n=%d' % (sum(d_us) / len(d_us), min(d_us), len(d_us)))
        for k, v in sorted(agg.items()):
            print(f"   {k:24s} {sum(v)/len(v):.4g}")
