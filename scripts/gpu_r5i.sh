#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 300 python scripts/bench_dec.py --reps 8 --hot --phases qkv,o,gate_up,down4,down6 > $O/dec_hot.log 2>&1; grep phase $O/dec_hot.log | cut -c1-130
MRS_DEC_PREFETCH=1 timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 128 > $O/bench_pf1.log 2>&1; tail -1 $O/bench_pf1.log | cut -c1-200
MRS_DEC_PREFETCH=0 timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 128 > $O/bench_pf0.log 2>&1; tail -1 $O/bench_pf0.log | cut -c1-200
MRS_DEC_PREFETCH=1 timeout 600 python -m pytest tests/test_dec_model.py tests/test_llama_runner.py -m gpu -x -q > $O/t_pf.log 2>&1; tail -2 $O/t_pf.log
MRS_DEC_PREFETCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py $f --top 14 --match "dec\|prefetch" 2>&1 | cut -c1-200
