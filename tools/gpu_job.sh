cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
timeout 1500 python -u -m pytest tests/test_gpu_step.py -q -m gpu -x -k "rccl or starts_its_own" 2>&1 | tail -n 40 > gpurun_out/r2j/rccl_test.log
DPN_BENCH_RCCL_ONE_RANK=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 timeout 900 python -u bench.py --no-cpu-baseline --no-alt > gpurun_out/r2j/bench_rccl1.json 2> gpurun_out/r2j/bench_rccl1.err
tail -n 30 gpurun_out/r2j/rccl_test.log; head -c 1500 gpurun_out/r2j/bench_rccl1.json; tail -n 5 gpurun_out/r2j/bench_rccl1.err
