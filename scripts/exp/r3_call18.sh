#!/bin/bash
O=gpurun_out/r3c18; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_mmq_abi.py tests/test_zz_fast_mmq.py tests/test_zz_gguf_matmul.py -q -m gpu > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-300
for m in 1 0; do echo "== MRS_MMQ_MFMA=$m" | tee -a $O/mmq_bench.log; MRS_MMQ_MFMA=$m timeout 300 python scripts/bench_gemm.py --mmq --types q8_0,q4_0,q5_1 --t 512 2>&1 | grep TFLOPs | tee -a $O/mmq_bench.log; done
