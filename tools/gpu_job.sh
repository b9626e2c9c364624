cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2f
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "kink_flips or longest_lead" -s 2>&1 | grep -E "^n = |F10:|after removing|passed|failed|Error|assert" > gpurun_out/r2f/tests.log
cat gpurun_out/r2f/tests.log
