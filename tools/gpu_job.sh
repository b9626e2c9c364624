cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2h
cp deepphysinet_amd/libdpn_hip_timeline.so /tmp/tl_base.so
for v in base dma; do
  if [ $v = dma ]; then cp deepphysinet_amd/libdpn_hip_tl_dma.so deepphysinet_amd/libdpn_hip_timeline.so; fi
  for prec in bf16 bf16x2; do timeout 300 python tools/timeline_probe.py $prec > gpurun_out/r2h/timeline_${v}_$prec.txt 2>&1; head -12 gpurun_out/r2h/timeline_${v}_$prec.txt | grep -v amdgpu; done
done
DPN_LIB=$PWD/deepphysinet_amd/libdpn_hip_tl_dma.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "vs_oracle or golden or deterministic" 2>&1 | tail -3
