cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2j
timeout 1500 python -u -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -n 2
for i in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then export DPN_LIB=$PWD/deepphysinet_amd/libdpn_hip_prev.so; else unset DPN_LIB; fi
  python -u bench.py --no-cpu-baseline > gpurun_out/r2j/ab_$v.json 2>/dev/null
  python - <<PY
import json
d = json.load(open('gpurun_out/r2j/ab_$v.json'))
o = d['other_precision_mode']
print('$v', 'x2: ms %.4f fwd %.1f wgrad %.1f bwd %.1f frac %.3f | bf16: ms %.4f fwd %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'] * 1e3, d['roofline_hbm_kernel']['kernel_ms'] * 1e3, d['roofline_hbm_kernel']['bwd_points_kernel_ms'] * 1e3, d['roofline']['frac'], o['ms_per_step'], o['roofline']['kernel_ms'] * 1e3, o['roofline']['frac']))
PY
done; done
