cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in bf16x2 bf16; do
for v in full HALFDMA QUARTERDMA; do
  L=$PWD/deepphysinet_amd/libdpn_hip_abl_$v.so; [ $v = full ] && L=$PWD/deepphysinet_amd/libdpn_hip_timeline.so
  cp $L /tmp/lib_probe.so
  echo "== $p $v"; python - <<PY 2>&1 | grep -v amdgpu.ids | grep "kernel + pack\|cycles per wave\|^L2 \|^fc1 \|^v \|^L1 " | cut -c1-150
import os, sys, runpy, shutil
sys.argv = ['tools/timeline_probe.py', '$p']
import importlib.util
src = open('tools/timeline_probe.py').read().replace("LIB = os.path.join(ROOT, 'deepphysinet_amd', 'libdpn_hip_timeline.so')", "LIB = '/tmp/lib_probe.so'")
exec(compile(src, 'tools/timeline_probe.py', 'exec'))
PY
done; done
