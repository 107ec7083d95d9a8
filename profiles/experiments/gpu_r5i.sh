#!/bin/bash
# relaxed attention ticket: the hand-off stress test, the decode engine / model parity files, the bench line and the in-graph kernel times
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 600 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -m gpu -x -q > $O/t1.log 2>&1; tail -2 $O/t1.log
timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 256 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py $f --top 14 --match dec 2>&1 | cut -c1-200
