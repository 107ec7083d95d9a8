#!/bin/bash
# Rebuild only the translation units that include dec_core2.cuh / dec_gemv.cuh (the decode engine and the exact prompt path) and relink libmrs_hip_ext.so;
# build.py rebuilds EVERY unit when any header changes.  Run from anywhere; then `touch` keeps build.py from redoing the rest.
# Usage: rebuild_dec.sh [--only "ext_p2p ext_decode ..."]   (--only: just these extra units + runtime.cpp, then relink)
set -e
ONLY=""
if [ "$1" = "--only" ]; then ONLY="$2"; fi
cd "$(dirname "$0")/../mistral.rs_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=off -Icsrc -I../include"
pids=""
if [ -z "$ONLY" ]; then
for nc in 1 2 3 4 5 6 7 8; do /opt/rocm/bin/hipcc $F -DMRS_DEC_NC=$nc -c csrc/ext_dec_gemv.hip -o csrc/build/ext_dec_gemv_nc$nc.o & pids="$pids $!"; done
for f in ext_dec ext_gemm_qi ext_prefetch; do /opt/rocm/bin/hipcc $F -c csrc/$f.hip -o csrc/build/$f.o & pids="$pids $!"; done
else
for f in $ONLY; do /opt/rocm/bin/hipcc $F -c csrc/$f.hip -o csrc/build/$f.o & pids="$pids $!"; done
fi
/opt/rocm/bin/hipcc -x hip $F -c csrc/host/runtime.cpp -o csrc/build/runtime.o & pids="$pids $!"
for p in $pids; do wait $p; done
objs=$(python - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("b", "build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(" ".join(os.path.join("csrc/build", tu[1]) for tu in b.libraries()["libmrs_hip_ext.so"]))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libmrs_hip_ext.so $objs -Llib -lmistralrsquant -lmistralrspagedattention -lmistralrscuda '-Wl,-rpath,$ORIGIN' -ldl
touch csrc/build/*.o lib/*.so
echo "relinked lib/libmrs_hip_ext.so"
