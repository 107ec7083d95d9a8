#!/bin/bash
# round 6, call N: PMC pass over the matrix-core batched launches (where does a wave's time go?)
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc1 -o r -- python scripts/bench_dec.py --b 8 --mm --reps 1 > $O/pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -o r -- python scripts/bench_dec.py --b 8 --mm --reps 1 > $O/pmc2.log 2>&1
for d in pmc1 pmc2; do python - $O/$d <<'PY'
import csv, glob, collections, sys, re
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])
        if 'dec_mm' in k: agg[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in sorted(agg.items()):
        print(k[0][-40:], k[1], len(next(iter(cs.values()))), ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
done
