cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fp8" 2>&1 | tail -12
python tools/fp8_encoder_experiment.py > gpurun_out/r2i/fp8_encoder_experiment.json 2> gpurun_out/r2i/fp8.err; tail -3 gpurun_out/r2i/fp8.err; cat gpurun_out/r2i/fp8_encoder_experiment.json
