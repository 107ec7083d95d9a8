#!/bin/sh
# TEST INFRASTRUCTURE: rebuild single units of oracle/_hiphost/libhiphost.so (after a full oracle/build_hip_host.sh): scripts/rebuild_hiphost_one.sh ext_gemm_qi [runtime] ...
# Headers are refreshed every time; `runtime` = csrc/host/runtime.cpp.
set -e
HERE="$(cd "$(dirname "$0")/../oracle" && pwd)"
CSRC="$HERE/../mistral.rs_amd/csrc"
OUT="$HERE/_hiphost"
CXX="/opt/rocm/lib/llvm/bin/clang++"
FLAGS="-x c++ -std=c++17 -O1 -fPIC -march=native -fno-fast-math -ffp-contract=off -w -I$HERE/hip_host -I$OUT/src -I$HERE/../include"
for f in "$CSRC"/*.cuh; do
  sed -E 's/extern __shared__( __attribute__\(\(aligned\(16\)\)\))? ([A-Za-z_0-9]+) ([A-Za-z_0-9]+)\[\];/\2 *\3 = (\2 *)hiphost::dyn_lds;/' "$f" > "$OUT/src/$(basename "$f")"
done
pids=""
for u in "$@"; do
  if [ "$u" = runtime ]; then
    cp "$CSRC/host/runtime.cpp" "$OUT/src/host/"
    $CXX $FLAGS -I"$HERE/../include" -c "$OUT/src/host/runtime.cpp" -o "$OUT/obj/runtime.o" & pids="$pids $!"
  else
    sed -E 's/extern __shared__( __attribute__\(\(aligned\(16\)\)\))? ([A-Za-z_0-9]+) ([A-Za-z_0-9]+)\[\];/\2 *\3 = (\2 *)hiphost::dyn_lds;/' "$CSRC/$u.hip" > "$OUT/src/$u.hip"
    $CXX $FLAGS -c "$OUT/src/$u.hip" -o "$OUT/obj/$u.o" & pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
$CXX -shared -Wl,-Bsymbolic -o "$OUT/libhiphost.so" "$OUT"/obj/*.o
echo "relinked oracle/_hiphost/libhiphost.so ($*)"
