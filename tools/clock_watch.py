"""Sample the GPU's clocks / power (rocm-smi) while a command runs: median and spread of sclk, mclk, socket power during the run.
usage: clock_watch.py <tag> -- <command ...>"""
import json, subprocess, sys, time, statistics
i = sys.argv.index('--')
tag, cmd = sys.argv[1], sys.argv[i + 1:]
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
rows = []
while p.poll() is None:
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        rows.append({k: v for k, v in c.items()})
    except Exception as e:          # noqa
        rows.append({'error': str(e)})
    time.sleep(0.05)
out = p.stdout.read()
keys = sorted({k for r in rows for k in r})
print('== %s: %d samples' % (tag, len(rows)))
for k in keys:
    vals = [r[k] for r in rows if k in r]
    nums = []
    for v in vals:
        try:
            nums.append(float(str(v).strip('()MhzW C').split('M')[0].split('W')[0]))
        except ValueError:
            pass
    if nums:
        print('   %-55s median %8.1f  min %8.1f  max %8.1f' % (k, statistics.median(nums), min(nums), max(nums)))
    else:
        print('   %-55s %s' % (k, sorted(set(map(str, vals)))[:4]))
print(out.strip().splitlines()[-1][:300] if out.strip() else '(no output)')
