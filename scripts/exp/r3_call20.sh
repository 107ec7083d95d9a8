#!/bin/bash
O=gpurun_out/r3c20; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_mmq_abi.py tests/test_zz_fast_mmq.py tests/test_zz_gguf_matmul.py tests/test_zz_moe_prefill.py -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
for b in 384 0 100000; do echo "== SMALL_TILES_BELOW=$b" | tee -a $O/mmq_bench.log; MRS_MMQ_SMALL_TILES_BELOW=$b timeout 300 python scripts/bench_gemm.py --mmq --types q4_k,q6_k,q8_0 --t 512 2>&1 | grep TFLOPs | cut -c1-120 | tee -a $O/mmq_bench.log; done
