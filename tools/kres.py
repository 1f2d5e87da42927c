"""Per-kernel register / LDS / spill summary of one HIP source (cross-compiles, no GPU needed).
Usage: python tools/kres.py classifying-vae-lstm_amd/csrc/lstm_pair.hip [filter-regex] [-- extra hipcc flags]
Also leaves the ISA in /tmp/kres/<name>.s for reading."""
import os
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--')
    args, extra = args[:i], args[i + 1:]
src = args[0]
filt = re.compile(args[1]) if len(args) > 1 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs('/tmp/kres', exist_ok=True)
out = '/tmp/kres/' + os.path.basename(src).replace('.hip', '.s')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only',
       '-I' + os.path.join(root, 'include'), '-Rpass-analysis=kernel-resource-usage', '-o', out, src] + extra
r = subprocess.run(cmd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=600)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r'remark:\s+(.*?)(?:\s+\[-Rpass.*)?$', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
if not rows:                      # compile error: show it (c++filt without names would wait on stdin)
    print(r.stderr[-4000:])
    sys.exit(1)
demangle = subprocess.run(['c++filt'] + [r_['name'] for r_ in rows], capture_output=True, text=True,
                          stdin=subprocess.DEVNULL)
names = demangle.stdout.splitlines() if demangle.returncode == 0 else [r_['name'] for r_ in rows]
print("%-100s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "spill", "scr", "occ", "LDS"))
for r_, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('clv::', '')
    if filt and not filt.search(n):
        continue
    print("%-100s %5s %5s %6s %6s %4s %7s" % (n[:100], r_.get('VGPRs'), r_.get('AGPRs'), r_.get('VGPRs Spill'),
                                            r_.get('ScratchSize [bytes/lane]'), r_.get('Occupancy [waves/SIMD]'),
                                            r_.get('LDS Size [bytes/block]')))
print("ISA:", out)
