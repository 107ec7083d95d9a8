#!/bin/bash
# round 6, call E: activation fragments read one group ahead in the exact prompt GEMMs (Q4_K, Q8_0): parity + TTFT
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_qi.py tests/test_prefill_exact.py -q -m gpu -x -rf > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log | cut -c1-300
(timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 2>&1 | tail -1) > $O/bench_512.log
(timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 32 --quant q8_0_isq 2>&1 | tail -1) > $O/bench_q80.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin --steps 16 > $O/kt.log 2>&1
python - <<'PY'
import json, csv, collections
for f in ("bench_512", "bench_q80"):
    j = json.loads(open(f"gpurun_out/r6e/{f}.log").read().strip().splitlines()[-1])
    print(f, "tok/s", j["value"], "ttft", j["ttft_ms"], "frac", j["prefill_roofline"]["frac"], "exact", j["prefill_arithmetic"][:30])
rows = list(csv.DictReader(open("gpurun_out/r6e/kt/r_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if "gemm_qi" in k or "gemm_q80" in k or "qi_quantize" in k or "prefill_attn_mfma" in k:
        print(f"{len(v):6d} {sum(v) / 1e3:9.3f} ms avg {sum(v) / len(v):8.2f}  {k[:110]}")
PY
