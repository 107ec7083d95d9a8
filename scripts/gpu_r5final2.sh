#!/bin/bash
# round-5 evidence, second pass (after the last kernel change): dec tests, per-phase times, default bench line, batch lines, the three rocprofv3 passes
export TMPDIR=/tmp
O=gpurun_out/r5final2; mkdir -p $O
timeout 900 python -m pytest tests/test_dec2_core.py tests/test_dec_engine.py tests/test_dec_model.py tests/test_llama_runner.py tests/test_kernel_resources.py -m gpu -q > $O/t1.log 2>&1; tail -2 $O/t1.log
timeout 300 python scripts/bench_dec.py --reps 8 > $O/dec.log 2>&1; grep phase $O/dec.log | cut -c1-130
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench_default.log; cut -c1-300 $O/bench_default.log
for b in 2 4 8; do (timeout 300 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 --batch $b 2>&1 | tail -1) > $O/bench_b$b.log; cut -c1-160 $O/bench_b$b.log; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 > $O/write.log 2>&1
find $O -name "*.csv" | head
