# round 2, first GPU pass: new step-machinery tests, full-size parity tests, bench (single graph, split graphs, self-launched 2 ranks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_step.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r2a/step_tests.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_grid_all or full_size_61" -s 2>&1 | tail -40 > gpurun_out/r2a/fullsize_tests.log
python -m pytest tests/test_sampler.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r2a/sampler_tests.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
DPN_BENCH_SPLIT_STEP=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt > gpurun_out/r2a/bench_split.json 2> gpurun_out/r2a/bench_split.err
DPN_BENCH_ONE_DEVICE=1 DPN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r2a/bench_2ranks_onedev.json 2> gpurun_out/r2a/bench_2ranks_onedev.err
tail -3 gpurun_out/r2a/*.log; cat gpurun_out/r2a/bench.json | head -c 1500; echo; cat gpurun_out/r2a/bench_split.json | head -c 600; echo; cat gpurun_out/r2a/bench_2ranks_onedev.json | head -c 600; tail -5 gpurun_out/r2a/*.err
