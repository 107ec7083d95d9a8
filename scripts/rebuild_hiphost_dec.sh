#!/bin/sh
# TEST INFRASTRUCTURE: rebuild only the decode-engine units of oracle/_hiphost/libhiphost.so (after a full oracle/build_hip_host.sh) -- the edit / check loop of
# dec_core2.cuh, dec_gemv.cuh, dec_attn.cuh, ext_dec.hip, ext_gemm_qi.hip.
set -e
HERE="$(cd "$(dirname "$0")/../oracle" && pwd)"
CSRC="$HERE/../mistral.rs_amd/csrc"
OUT="$HERE/_hiphost"
CXX="/opt/rocm/lib/llvm/bin/clang++"
for f in "$CSRC"/dec_core2.cuh "$CSRC"/dec_gemv.cuh "$CSRC"/dec_attn.cuh "$CSRC"/ext_dec.hip "$CSRC"/ext_dec_gemv.hip "$CSRC"/ext_gemm_qi.hip "$CSRC"/common.cuh "$CSRC"/gguf_blocks.cuh; do
  sed -E 's/extern __shared__( __attribute__\(\(aligned\(16\)\)\))? ([A-Za-z_0-9]+) ([A-Za-z_0-9]+)\[\];/\2 *\3 = (\2 *)hiphost::dyn_lds;/' "$f" > "$OUT/src/$(basename "$f")"
done
FLAGS="-x c++ -std=c++17 -O1 -fPIC -march=native -fno-fast-math -ffp-contract=off -w -I$HERE/hip_host -I$OUT/src -I$HERE/../include"
pids=""
for nc in ${NCS:-1 2 3 4 5 6 7 8}; do $CXX $FLAGS -DMRS_DEC_NC=$nc -c "$OUT/src/ext_dec_gemv.hip" -o "$OUT/obj/ext_dec_gemv_nc$nc.o" & pids="$pids $!"; done
for f in ext_dec ext_gemm_qi; do $CXX $FLAGS -c "$OUT/src/$f.hip" -o "$OUT/obj/$f.o" & pids="$pids $!"; done
for p in $pids; do wait $p; done
rm -f "$OUT/obj/ext_dec2.o"
$CXX -shared -Wl,-Bsymbolic -o "$OUT/libhiphost.so" "$OUT"/obj/*.o
echo "relinked oracle/_hiphost/libhiphost.so"
