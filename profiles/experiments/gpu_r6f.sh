#!/bin/bash
# round 6, call F: instruction-cache cold-fetch probe, the I-cache counters of the decode step, split-router tests
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
timeout 120 profiles/experiments/icache_cold_probe > $O/icache_cold_probe.log 2>&1; cat $O/icache_cold_probe.log
(rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_|INST_LEVEL|SQ_INSTS_SALU|SQ_WAIT_INST|SQ_IFETCH" | cut -c1-160 | sort -u | head -60) > $O/avail.log; cat $O/avail.log
timeout 900 python -m pytest tests/test_dec_engine.py tests/test_moe.py tests/test_bf16_shadow.py -q -m gpu -rf -k "router or shadow or moe" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log | cut -c1-300
timeout 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $O/pmc_ic -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/pmc_ic.log 2>&1; tail -3 $O/pmc_ic.log | cut -c1-300
python - $O/pmc_ic <<'PY' 2>&1 | head -60
import csv, glob, collections, sys, re
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])
        if 'mrs::' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        print(k[:70], len(next(iter(cs.values()))), ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
