#!/bin/bash
O=gpurun_out/r3c7; mkdir -p $O
cd /root/repo
run() { local name=$1; shift
  echo "== $name" | tee -a $O/bench.log
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 128 --weights blocks 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ms/step', j['ms_per_step'], 'gate_up us', j['roofline']['us_per_launch'])" | tee -a $O/bench.log
}
run b1_two_launch MRS_EXT_LIB=libmrs_hip_ext_b1.so
run b1_ticket MRS_EXT_LIB=libmrs_hip_ext_b1.so MRS_DEC_ATTN_TICKET=1
run b1d_diet MRS_EXT_LIB=libmrs_hip_ext_b1d.so
echo "== b1d gaussian weights" | tee -a $O/bench.log; MRS_EXT_LIB=libmrs_hip_ext_b1d.so timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | cut -c1-160 | tee -a $O/bench.log
MRS_EXT_LIB=libmrs_hip_ext_b1d.so timeout 300 python scripts/bench_dec.py 2>&1 | grep phase | cut -c1-110 | tee -a $O/bench.log
