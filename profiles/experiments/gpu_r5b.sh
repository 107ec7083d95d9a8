#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_dec2_core.py tests/test_dec_engine.py -m gpu -x -q > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 300 python scripts/bench_dec.py --reps 8 > $O/dec.log 2>&1; cat $O/dec.log | cut -c1-130
for v in ns4 ns6 ns12; do MRS_EXT_LIB=libmrs_hip_ext_$v.so timeout 300 python scripts/bench_dec.py --reps 8 --phases qkv,o,gate_up,down4,down6 > $O/dec_$v.log 2>&1; echo $v; cut -c1-110 $O/dec_$v.log; done
timeout 300 python scripts/bench_dec.py --timeline > $O/tl.log 2>&1; cat $O/tl.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); python scripts/rocprof_summary.py $f --top 12 --match dec 2>&1 | cut -c1-200
timeout 900 python -m pytest tests/test_dec_model.py tests/test_llama_runner.py tests/test_moe.py -m gpu -x -q > $O/t2.log 2>&1; tail -3 $O/t2.log
