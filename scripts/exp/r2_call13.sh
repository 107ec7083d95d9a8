mkdir -p gpurun_out/r2c13
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'],'tok/s',j['ms_per_step'],'ms step_frac',j['step_roofline_frac'],j['greedy_tokens_head'][:5])"; }
(run persist_fence MRS_DEC_PERSIST=1
run persist_nofence MRS_DEC_PERSIST=1 MRS_EXT_LIB=libmrs_hip_ext_nofence.so
run phases_nofence MRS_DEC_PERSIST=2 MRS_EXT_LIB=libmrs_hip_ext_nofence.so
run launches MRS_DEC_PERSIST=0) 2>&1 | tee gpurun_out/r2c13/bench.log
