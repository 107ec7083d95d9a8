#!/bin/bash
# Round-2 A/B on the GPU box (one gpurun call, ~4 min): the shipped decode kernels (1 workgroup of 8 waves per CU, 133-139 VGPRs) against the
# 128-VGPR build (profiles/experiments/build_occ4.sh) at 1x and 2x workgroups per CU.  Each line is a full `bench.py` decode measurement; parity of the
# variant is checked first (same greedy tokens as the default build: greedy_tokens_head in the JSON line).
set -u
OUT=gpurun_out/occ4
mkdir -p $OUT
run() { tag=$1; shift; (env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 128 2>&1 | tail -1) > $OUT/$tag.json; python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], j["value"], "tok/s", j["ms_per_step"], "ms", "roofline", j["roofline"]["frac"], j["greedy_tokens_head"][:4])
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-300:])
PY
}
run base A=1
run occ4_1x MRS_EXT_LIB=libmrs_hip_ext_occ4.so
run occ4_2x MRS_EXT_LIB=libmrs_hip_ext_occ4.so MRS_PROJ_WGS=512 MRS_GLU_PER=32
run occ4_2x_qkv MRS_EXT_LIB=libmrs_hip_ext_occ4.so MRS_PROJ_WGS=512 MRS_GLU_PER=32 MRS_QKV_PPW=1
run base_2x MRS_PROJ_WGS=512 MRS_GLU_PER=32
