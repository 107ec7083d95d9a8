#!/bin/bash
O=gpurun_out/r3c6; mkdir -p $O
cd /root/repo
export MRS_EXT_LIB=libmrs_hip_ext_b1.so
run() { local name=$1; shift
  echo "== $name" | tee -a $O/bench.log
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ms/step', j['ms_per_step'], 'gate_up us', j['roofline']['us_per_launch'])" | tee -a $O/bench.log
}
run all_on X=1
run small_off MRS_DEC_SMALL=0
run qkvbal_off MRS_DEC_QKV_BALANCE=0
run attn2_off MRS_DEC_ATTN2=0
run staged_off MRS_DEC_STAGED=0
timeout 300 python scripts/bench_dec.py 2>&1 | grep phase | cut -c1-110 | tee -a $O/bench.log
