#!/bin/bash
O=gpurun_out/r3c22; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_paged_attn.py tests/test_llama_runner.py tests/test_fp8_kv.py -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-300
run() { echo "== $*" | tee -a $O/dropin.log; (env "$@" timeout 400 python bench.py --no-cpu-baseline --steps 64 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s', j['value'], 'dropin', j.get('dropin_tokens_per_sec'), j.get('dropin_step_roofline_frac'))") | tee -a $O/dropin.log; }
run MRS_X=0
run MRS_PA_WIDE=8
run MRS_PA_WIDE=0
run MRS_PA_MAX_G=2
