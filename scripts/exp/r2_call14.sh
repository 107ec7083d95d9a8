mkdir -p gpurun_out/r2c14
(echo "== base"; timeout 200 python scripts/bench_dec.py --reps 8
echo "== LDS pad 82000 (1 WG/CU forced)"; MRS_DEC_LDS_PAD=82000 timeout 200 python scripts/bench_dec.py --reps 8
echo "== depth 12/4"; MRS_EXT_LIB=libmrs_hip_ext_d12.so timeout 200 python scripts/bench_dec.py --reps 8
echo "== depth 16/6"; MRS_EXT_LIB=libmrs_hip_ext_d16.so timeout 200 python scripts/bench_dec.py --reps 8
echo "== depth 16/6 + pad"; MRS_DEC_LDS_PAD=82000 MRS_EXT_LIB=libmrs_hip_ext_d16.so timeout 200 python scripts/bench_dec.py --reps 8) 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('  %-8s %6.2f us  %5.3f TB/s' % (j['phase'], j['us'], j['TBps']))
    else: print(l)
" | tee gpurun_out/r2c14/bench.log
