mkdir -p gpurun_out/r2c16
timeout 600 python -m pytest tests/test_distributed.py tests/test_dec_model.py tests/test_hqq.py -m gpu -q -p no:cacheprovider -k "p2p or short_prompt or 3bit_default" 2>&1 | tail -5
export MRS_EXT_LIB=libmrs_hip_ext_ab.so
(echo "== full"; timeout 200 python scripts/bench_dec.py --reps 8
echo "== no prologue arithmetic (ABLATE=1)"; MRS_DEC_ABLATE=1 timeout 200 python scripts/bench_dec.py --reps 8
echo "== loads only (ABLATE=2)"; MRS_DEC_ABLATE=2 timeout 200 python scripts/bench_dec.py --reps 8
echo "== neither (ABLATE=3)"; MRS_DEC_ABLATE=3 timeout 200 python scripts/bench_dec.py --reps 8) 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('  %-8s %6.2f us  %5.3f TB/s' % (j['phase'], j['us'], j['TBps']))
    else: print(l)
" | tee gpurun_out/r2c16/ablate.log
