mkdir -p gpurun_out/r2c21
(echo "== NT 512 (shipped)"; timeout 200 python scripts/bench_dec.py --reps 8
echo "== NT 1024"; MRS_EXT_LIB=libmrs_hip_ext_nt1024.so timeout 200 python scripts/bench_dec.py --reps 8
echo "== NT 1024, 2 waves' worth fewer units (UPW x2 -> 128 WGs)"; MRS_EXT_LIB=libmrs_hip_ext_nt1024.so MRS_DEC_UPW=0 timeout 200 python scripts/bench_dec.py --reps 8 --phases gate_up) 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('  %-8s %6.2f us  %5.3f TB/s' % (j['phase'], j['us'], j['TBps']))
    else: print(l)
" | tee gpurun_out/r2c21/nt.log
MRS_EXT_LIB=libmrs_hip_ext_nt1024.so timeout 300 python -m pytest tests/test_dec_engine.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
MRS_EXT_LIB=libmrs_hip_ext_nt1024.so timeout 300 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | cut -c1-200
