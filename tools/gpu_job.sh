cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
timeout 2400 python -u -m pytest tests -q -m gpu -x 2>&1 | tail -n 12 > gpurun_out/r2j/gpu_tests.log
tail -n 12 gpurun_out/r2j/gpu_tests.log
python -u bench.py > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; head -c 3400 gpurun_out/r2j/bench.json; tail -n 3 gpurun_out/r2j/bench.err
