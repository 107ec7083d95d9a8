mkdir -p gpurun_out/r2c12
timeout 300 python -m pytest tests/test_dec_model.py -m gpu -q -p no:cacheprovider -x -k "persistent" 2>&1 | tail -3
for mode in 1 2 0; do echo "PERSIST=$mode"; MRS_DEC_PERSIST=$mode timeout 300 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'],'tok/s',j['ms_per_step'],'ms step_frac',j['step_roofline_frac'],'prefill',j['prefill_tokens_per_sec'],j['greedy_tokens_head'][:5])"; done 2>&1 | tee gpurun_out/r2c12/bench.log
export TMPDIR=/tmp
MRS_DEC_PERSIST=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2c12/prof -- python bench.py --no-cpu-baseline --steps 12 --warmup 2 > gpurun_out/r2c12/prof.log 2>&1
python scripts/exp/phase_times.py gpurun_out/r2c12/prof 32 | tee gpurun_out/r2c12/phases.log
rm -rf gpurun_out/r2c12/prof
