# scratch GPU job of the moment (rewritten between gpurun calls)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2e
for prec in bf16x2 bf16; do
  rm -rf gpurun_out/r2e/prof_$prec
  rocprofv3 --kernel-trace --stats -d gpurun_out/r2e/prof_$prec -o trace -- python3 bench.py --leads 61 --steps 3 --warmup 1 --prec $prec --no-cpu-baseline --no-alt > gpurun_out/r2e/bench_cfg2_$prec.log 2>&1
  DB=$(find gpurun_out/r2e/prof_$prec -name "*.db" | head -1)
  python tools/prof_summary.py $DB 30 > gpurun_out/r2e/cfg2_kernel_stats_$prec.txt
  find gpurun_out/r2e/prof_$prec -name "*.db" -delete
done
python bench.py --leads 61 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/r2e/bench_cfg2_bf16x2.json 2> gpurun_out/r2e/bench_cfg2.err
python bench.py --leads 61 --steps 5 --warmup 2 --prec bf16 --no-cpu-baseline --no-alt > gpurun_out/r2e/bench_cfg2_bf16.json 2>> gpurun_out/r2e/bench_cfg2.err
python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py -q -m gpu -k "two_ranks or full_grid_all or default_init or full_size_61 or bench_starts" -s 2>&1 | tail -30 > gpurun_out/r2e/tests.log
cat gpurun_out/r2e/cfg2_kernel_stats_bf16x2.txt; tail -12 gpurun_out/r2e/tests.log; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r2e/bench_cfg2_*.json
