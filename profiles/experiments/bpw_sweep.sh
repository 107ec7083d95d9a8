#!/bin/bash
cd /root/repo
run() { echo "== $*"; (env "$@" timeout 400 python bench.py --no-cpu-baseline --no-dropin --steps 128 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s', j['value'], j['ms_per_step'])"); }
run MRS_X=0
run MRS_DEC_ATTN_BPW=2
run MRS_DEC_ATTN_BPW=3
run MRS_DEC_ATTN_BPW=4
run MRS_X=1
