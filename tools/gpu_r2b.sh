cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_step.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r2b/step_tests.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_grid_all" -s 2>&1 | tail -40 > gpurun_out/r2b/fullsize_tests.log
python tools/timeline_probe.py bf16x2 > gpurun_out/r2b/timeline_bf16x2.txt 2>&1
python tools/timeline_probe.py bf16 > gpurun_out/r2b/timeline_bf16.txt 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
tail -n 12 gpurun_out/r2b/step_tests.log; tail -n 14 gpurun_out/r2b/fullsize_tests.log; head -c 1200 gpurun_out/r2b/bench.json
