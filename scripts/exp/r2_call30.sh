#!/bin/bash
O=gpurun_out/r2c30; mkdir -p $O
cd /root/repo
for pl in 512 2048; do for v in 1 0; do
  echo "== prompt $pl MRS_MOE_PREFILL_MFMA=$v"
  MRS_MOE_PREFILL_MFMA=$v timeout 900 python scripts/bench_mixtral.py --steps 16 --prompt-len $pl 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('prompt tok/s', j['prompt_tokens_per_sec'], 'decode tok/s', j['value'])"
done; done 2>&1 | tee $O/mixtral_prompt.log
