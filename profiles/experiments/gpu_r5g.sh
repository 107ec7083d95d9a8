#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_dec2_core.py tests/test_dec_engine.py tests/test_dec_model.py tests/test_llama_runner.py tests/test_moe.py tests/test_sampling.py tests/test_prefill_exact.py tests/test_gemm_qi.py -m gpu -x -q > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 300 python scripts/bench_dec.py --reps 8 > $O/dec.log 2>&1; grep phase $O/dec.log | cut -c1-130
timeout 300 python scripts/bench_dec.py --timeline --phases qkv,gate_up,down4 > $O/tl.log 2>&1; grep median $O/tl.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-600
for b in 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 --batch $b > $O/bench_b$b.log 2>&1; tail -1 $O/bench_b$b.log | cut -c1-200; done
