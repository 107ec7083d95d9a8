#!/bin/bash
# round 6, call S: ablations of the matrix-core record arithmetic (variant libraries built from patched copies of ext_dec_mm.hip: results are wrong on purpose, the times say
# which part of a record a wave pays for): no LDS reads of the A operands / no scale products / no nibble unpack / no arithmetic at all
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
for v in base nolds noprod nounpack nocompute; do
  if [ $v = base ]; then unset MRS_EXT_LIB; else export MRS_EXT_LIB=libmrs_hip_ext_$v.so; fi
  timeout 300 python scripts/bench_dec.py --b 8 --mm --phases o,gate_up,down4 > $O/dec_mm_$v.log 2>&1
  echo "== $v"; python - $O/dec_mm_$v.log <<'PY'
import json, sys
print("  ".join(f"{j['phase']} {j['us']}" for j in (json.loads(l) for l in open(sys.argv[1]) if l.startswith("{"))) or open(sys.argv[1]).read()[-300:])
PY
done
