#!/bin/bash
O=gpurun_out/r3c17; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_mmq_abi.py tests/test_imatrix.py tests/test_zz_isq_kquants.py tests/test_isq.py -q -m gpu > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-300
for m in 1 0; do echo "== MRS_MMQ_MFMA=$m" | tee -a $O/mmq_bench.log; MRS_MMQ_MFMA=$m timeout 300 python scripts/bench_gemm.py --mmq --types q6_k --t 512 2>&1 | grep TFLOPs | tee -a $O/mmq_bench.log; done
MRS_MMQ_MFMA=1 timeout 300 python scripts/bench_gemm.py --mmq --types q6_k --t 2048 2>&1 | grep TFLOPs | tee -a $O/mmq_bench.log
