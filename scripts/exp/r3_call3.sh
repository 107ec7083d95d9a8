#!/bin/bash
O=gpurun_out/r3c3; mkdir -p $O
cd /root/repo
for v in b1 d12 d16; do
  for s in 1 0; do
    echo "== $v staged=$s" | tee -a $O/bench.log
    MRS_EXT_LIB=libmrs_hip_ext_$v.so MRS_DEC_STAGED=$s timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ms/step', j['ms_per_step'], 'gate_up us', j['roofline']['us_per_launch'])" | tee -a $O/bench.log
  done
done
