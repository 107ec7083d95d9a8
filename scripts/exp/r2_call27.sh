#!/bin/bash
O=gpurun_out/r2c27; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_dec_engine.py tests/test_dec_model.py -m gpu -x -q 2>&1 | tail -3
for cfg in "MRS_DEC_FUSED_ATTN=0" "MRS_DEC_FUSED_ATTN=1 MRS_DEC_ATTN_WAVES=12" "MRS_DEC_FUSED_ATTN=1 MRS_DEC_ATTN_WAVES=8"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], 'tok/s', j['ms_per_step'], 'ms', 'greedy', j.get('greedy_match'), j['greedy_tokens_head'][:4])"
done 2>&1 | tee $O/bench.log
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --steps 64 > $O/kt.log 2>&1
python scripts/rocprof_summary.py $O/kt 2>/dev/null | head -14
