cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
timeout 2400 python -u -m pytest tests -q -m gpu -x 2>&1 | tail -n 5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
