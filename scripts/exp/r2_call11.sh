mkdir -p gpurun_out/r2c11
export TMPDIR=/tmp
MRS_DEC_PERSIST=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2c11/prof -- python bench.py --no-cpu-baseline --steps 12 --warmup 2 > gpurun_out/r2c11/bench.log 2>&1
tail -1 gpurun_out/r2c11/bench.log | cut -c1-200
python scripts/exp/phase_times.py gpurun_out/r2c11/prof 32
rm -rf gpurun_out/r2c11/prof
