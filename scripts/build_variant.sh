#!/bin/bash
# A/B kernel experiments: build lib/libmrs_hip_ext_<name>.so = the default objects with the batch-1 decode GEMV unit (ext_dec_gemv, NC = 1) recompiled with extra
# defines; select it on the GPU box with MRS_EXT_LIB=libmrs_hip_ext_<name>.so (mistral.rs_amd/_lib.py).  Usage: build_variant.sh <name> -DMRS_DEC2_NS_Q4K=4 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../mistral.rs_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -Icsrc -I../include"
mkdir -p csrc/build/var_$name
/opt/rocm/bin/hipcc $F -DMRS_DEC_NC=1 "$@" -c csrc/ext_dec_gemv.hip -o csrc/build/var_$name/ext_dec_gemv_nc1.o 2>/dev/null
objs=$(python - <<PY
import importlib.util, os
spec = importlib.util.spec_from_file_location("b", "build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(" ".join(("csrc/build/var_$name/" + tu[1]) if tu[1] == "ext_dec_gemv_nc1.o" else os.path.join("csrc/build", tu[1]) for tu in b.libraries()["libmrs_hip_ext.so"]))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmrs_hip_ext_$name.so $objs -Llib -lmistralrsquant -lmistralrspagedattention -lmistralrscuda '-Wl,-rpath,$ORIGIN' -ldl
echo "built lib/libmrs_hip_ext_$name.so"
