cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
(timeout 1500 python -u tools/soak.py bf16x2 1000; timeout 1500 python -u tools/soak.py bf16 2000) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2j/soak.txt
tail -n 12 gpurun_out/r2j/soak.txt
