mkdir -p gpurun_out/r2c9
timeout 300 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -x -k "persistent" 2>&1 | tail -5 > gpurun_out/r2c9/pytest.log
cat gpurun_out/r2c9/pytest.log
for mode in 1 2 0; do echo "PERSIST=$mode"; MRS_DEC_PERSIST=$mode timeout 300 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'],'tok/s',j['ms_per_step'],'ms step_frac',j['step_roofline_frac'],'prefill',j['prefill_tokens_per_sec'],j['greedy_tokens_head'][:5])"; done 2>&1 | tee gpurun_out/r2c9/bench.log
