cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2g
timeout 200 python -u tools/feat_debug.py bf16 2>&1 | grep -E "ok|equal|Kernel Name|error" | tail -4
timeout 200 python -u tools/feat_debug.py bf16x2 2>&1 | grep -E "ok|equal|Kernel Name|error" | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "vs_oracle or golden or clip_masks or deterministic or hipgraph or full_grid_properties or kink or config2_lead" 2>&1 | tail -8 > gpurun_out/r2g/parity.log
timeout 300 python tools/timeline_probe.py bf16 > gpurun_out/r2g/timeline_bf16.txt 2>&1
timeout 300 python tools/timeline_probe.py bf16x2 > gpurun_out/r2g/timeline_bf16x2.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
tail -n 5 gpurun_out/r2g/parity.log; head -12 gpurun_out/r2g/timeline_bf16.txt; head -12 gpurun_out/r2g/timeline_bf16x2.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('other_precision_mode'), d['pde_losses'])
PY
