cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
for p in bf16x2 bf16; do
timeout 900 python -u tools/wgrad_phase_probe.py $p 2>&1 | grep -v amdgpu.ids > gpurun_out/r2j/wgrad_phases_$p.txt
grep "product\|kernel" gpurun_out/r2j/wgrad_phases_$p.txt | cut -c1-200
done
timeout 1500 python -u -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -n 3
python -u bench.py --no-cpu-baseline > gpurun_out/r2j/b.json 2>/dev/null
python - <<PY
import json
d = json.load(open('gpurun_out/r2j/b.json'))
o = d['other_precision_mode']
print('x2: ms %.4f fwd %.1f wgrad %.1f bwd %.1f | bf16: ms %.4f fwd %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms'] * 1e3, d['roofline_hbm_kernel']['kernel_ms'] * 1e3, d['roofline_hbm_kernel']['bwd_points_kernel_ms'] * 1e3, o['ms_per_step'], o['roofline']['kernel_ms'] * 1e3))
PY
