#!/bin/bash
# round 6, call D: fused qkv / gate+up library GEMMs with candidate timing on the bf16 shadow path; (longest, shortest) tile pairing in the MFMA prompt attention
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 600 python -m pytest tests/test_bf16_shadow.py tests/test_prefill_exact.py tests/test_dec_model.py -q -m gpu -x -rf > $O/pytest_new.log 2>&1; tail -6 $O/pytest_new.log | cut -c1-300
(timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --steps 64 2>&1 | tail -1) > $O/bench_512.log; cut -c1-200 $O/bench_512.log
(timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-extra --prompt-len 2048 --steps 64 2>&1 | tail -1) > $O/bench_p2048.log; cut -c1-200 $O/bench_p2048.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python bench.py --no-cpu-baseline --no-extra --no-dropin --steps 16 > $O/kt.log 2>&1; tail -1 $O/kt.log | cut -c1-200
python - <<'PY'
import json
for f in ("bench_512", "bench_p2048"):
    j = json.loads(open(f"gpurun_out/r6d/{f}.log").read().strip().splitlines()[-1])
    print(f, "tok/s", j["value"], "ttft", j["ttft_ms"], "frac", j["prefill_roofline"]["frac"], "bf16", json.dumps(j.get("prefill_bf16"))[:400])
PY
