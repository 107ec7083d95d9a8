#!/bin/bash
# round 6, call T: ext_dec_mm.hip without wave-dead image requests and with rings of 5 / 3 records on the launches of <= 256 units (MRS_DEC_MM_DEEP, default 1)
export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
timeout 900 python -m pytest tests/test_dec_mm.py -q -m gpu -x -rf > $O/pytest_mm.log 2>&1; tail -3 $O/pytest_mm.log | cut -c1-300
for w in 1 0; do for b in 8 3; do
  MRS_DEC_MM_DEEP=$w timeout 300 python scripts/bench_dec.py --b $b --mm > $O/dec_mm_b${b}_d$w.log 2>&1
  echo "== b=$b mm DEEP=$w"; python - $O/dec_mm_b${b}_d$w.log <<'PY'
import json, sys
print("  ".join(f"{j['phase']} {j['us']}" for j in (json.loads(l) for l in open(sys.argv[1]) if l.startswith("{"))))
PY
done; done
run() { name=$1; shift; (timeout 700 python bench.py --no-cpu-baseline --no-dropin --no-extra "$@" 2>&1 | tail -1) > $O/line_$name.log; python - "$O/line_$name.log" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "tok/s", j["value"], "ms", j["ms_per_step"], "step_frac", j.get("step_roofline_frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
run b8 --batch 8 --steps 64
run b4 --batch 4 --steps 64
run b3 --batch 3 --steps 64
MRS_DEC_MM_MIN_B=2 run b2 --batch 2 --steps 64
