#!/bin/bash
O=gpurun_out/r3c16; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_mmq_abi.py tests/test_zz_fast_mmq.py -q -m gpu -x > $O/mmq_tests.log 2>&1; tail -4 $O/mmq_tests.log | cut -c1-300
for m in 1 0; do for t in 512 2048; do echo "== MRS_MMQ_MFMA=$m T=$t" | tee -a $O/mmq_bench.log; MRS_MMQ_MFMA=$m timeout 300 python scripts/bench_gemm.py --mmq --types q4_k,q5_k --t $t 2>&1 | grep TFLOPs | tee -a $O/mmq_bench.log; done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python scripts/bench_gemm.py --mmq --types q4_k --t 512 > $O/kt.log 2>&1
python scripts/rocprof_summary.py $(find $O/kt -name "*kernel_trace.csv" | head -1) --top 6 --match mrs:: 2>/dev/null | cut -c1-200 | tee $O/kt_stats.log
timeout 1500 python -m pytest tests -q -m gpu -rf > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|FAILED|Fatal|fault" $O/pytest_gpu_full.log | tail -12 | cut -c1-300
