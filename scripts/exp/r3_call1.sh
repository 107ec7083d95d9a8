#!/bin/bash
# round 3, call 1: baseline on this box + zero-code knobs (kernarg placement, HW queues), clocks
O=gpurun_out/r3c1; mkdir -p $O
cd /root/repo
rocm-smi --showclocks --showpower 2>&1 | head -30 > $O/smi.log
run() { # name, env...
  local name=$1; shift
  echo "== $name" | tee -a $O/bench.log
  env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ms/step', j['ms_per_step'], 'step_frac', j['step_roofline_frac'], 'gate_up us', j['roofline']['us_per_launch'], 'ttft', j['ttft_ms'], 'prefill frac', j['prefill_roofline']['frac'])" | tee -a $O/bench.log
}
run default X=1
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq1 GPU_MAX_HW_QUEUES=1
run default_again X=1
echo "== bench_dec" | tee -a $O/bench.log
timeout 300 python scripts/bench_dec.py 2>&1 | tail -12 | tee -a $O/bench.log
