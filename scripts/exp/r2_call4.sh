mkdir -p gpurun_out/r2c4
(timeout 300 python scripts/bench_dec.py --old --reps 8
echo UPW=1; MRS_DEC_UPW=1 timeout 200 python scripts/bench_dec.py --phases qkv,o,down4 --reps 8
echo UPW=4; MRS_DEC_UPW=4 timeout 200 python scripts/bench_dec.py --phases qkv,o,down4,down6,gate_up --reps 8
echo UPW=14; MRS_DEC_UPW=14 timeout 200 python scripts/bench_dec.py --phases gate_up --reps 8
echo b=4; timeout 200 python scripts/bench_dec.py --b 4 --phases o,gate_up,down4 --reps 8) > gpurun_out/r2c4/bench_dec.log 2>&1
cat gpurun_out/r2c4/bench_dec.log
