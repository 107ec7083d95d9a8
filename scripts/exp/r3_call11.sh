#!/bin/bash
O=gpurun_out/r3c11; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
for g in 1 0; do
MRS_PREFILL_GEMM2=$g timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$g -o r -- python bench.py --no-cpu-baseline --no-dropin --steps 8 > $O/kt$g.log 2>&1
echo "== gemm2=$g" | tee -a $O/stats.log
python scripts/rocprof_summary.py $(find $O/kt$g -name "*kernel_trace.csv" | head -1) --top 14 --match mrs:: 2>/dev/null | cut -c1-170 | tee -a $O/stats.log
done
for f in tests/test_distributed.py tests/test_dec_engine.py tests/test_gemm.py tests/test_moe.py tests/test_llama_runner.py tests/test_dec_model.py tests/test_scheduler.py; do
  echo "== $f" | tee -a $O/pytest.log
  (timeout 1500 python -m pytest $f -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension modules" | tail -12) | tee -a $O/pytest.log
done
