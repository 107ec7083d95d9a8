#!/bin/bash
# round 6, call M: ext_dec_mm.hip with the image batch ahead of the ring and grouped MFMAs: tests, per-phase times + timeline, batched bench lines
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 900 python -m pytest tests/test_dec_mm.py -q -m gpu -x -rf > $O/pytest_mm.log 2>&1; tail -3 $O/pytest_mm.log | cut -c1-300
for b in 8 2; do
  timeout 300 python scripts/bench_dec.py --b $b --mm > $O/dec_mm_b${b}.log 2>&1
  echo "== b=$b mm"; python - $O/dec_mm_b${b}.log <<'PY'
import json, sys
print("  ".join(f"{j['phase']} {j['us']}" for j in (json.loads(l) for l in open(sys.argv[1]) if l.startswith("{"))))
PY
done
run() { name=$1; shift; (timeout 700 python bench.py --no-cpu-baseline --no-dropin --no-extra "$@" 2>&1 | tail -1) > $O/line_$name.log; python - "$O/line_$name.log" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "tok/s", j["value"], "ms", j["ms_per_step"], "step_frac", j.get("step_roofline_frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
run b8_mm --batch 8 --steps 64
run b4_mm --batch 4 --steps 64
run b2_mm --batch 2 --steps 64
