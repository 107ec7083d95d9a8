#!/bin/bash
# round-5 evidence refresh after the relaxed attention ticket: default bench line, in-graph kernel trace, batched lines, and a kernel trace of the batch-8 step
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench_default.log; cut -c1-200 $O/bench_default.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
for b in 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 --batch $b > $O/bench_b$b.log 2>&1; tail -1 $O/bench_b$b.log | cut -c1-200; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 --batch 8 > $O/kt8.log 2>&1
f=$(find $O/kt8 -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py $f --top 16 --match dec 2>&1 | cut -c1-200
