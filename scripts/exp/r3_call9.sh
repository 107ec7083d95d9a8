#!/bin/bash
O=gpurun_out/r3c9; mkdir -p $O
cd /root/repo
for t in 512 2048; do
  echo "== old T=$t" | tee -a $O/gemm.log; timeout 300 python scripts/bench_gemm.py --big --t $t 2>&1 | grep type | tee -a $O/gemm.log
  echo "== g2 T=$t" | tee -a $O/gemm.log; timeout 300 python scripts/bench_gemm.py --g2 --t $t 2>&1 | grep type | tee -a $O/gemm.log
done
(timeout 900 python -m pytest tests/test_gemm.py -x -q -m gpu -k "gemm2" 2>&1 | tail -5) | tee $O/pytest_gemm2.log
for g in 1 0; do echo "== bench gemm2=$g" | tee -a $O/bench.log; MRS_PREFILL_GEMM2=$g timeout 600 python bench.py --no-cpu-baseline --steps 64 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ttft', j['ttft_ms'], 'prefill tok/s', j['prefill_tokens_per_sec'], 'prefill frac', j['prefill_roofline']['frac'])" | tee -a $O/bench.log; done
