#!/bin/bash
O=gpurun_out/r3c4; mkdir -p $O
cd /root/repo
(timeout 1200 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -x -q -m gpu 2>&1 | tail -15) | tee $O/pytest.log
for a in 1 0; do
  echo "== attn2=$a" | tee -a $O/bench.log
  MRS_DEC_ATTN2=$a timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('tok/s', j['value'], 'ms/step', j['ms_per_step'], 'step_frac', j['step_roofline_frac'], 'gate_up us', j['roofline']['us_per_launch'])" | tee -a $O/bench.log
done
