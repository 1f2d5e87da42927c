#!/bin/bash
# Round 5: do lstm_mx's per-step stores / loads camp on a few HBM channels?  At T = 256 a batch row's records are
# 88 / 44 / 22 pages of 4 KB apart (coef / aux / hs) and every workgroup touches the same t at about the same time.
# The same kernels at T = 255, 257, 250, 264: per-step time (us per launch / T) should not depend on T if they do not.
cd /root/repo; G=gpurun_out; O=$G/r05_mx_stride.txt; : > $O
for i in 1 2; do
  for T in 256 257 255 250 264 248; do
    echo -n "T=$T  " >> $O
    timeout 300 python tools/mx_bench.py 1024 $T 32 2>&1 | grep -E "new_|copy" | tr '\n' ' ' | python3 -c "
import sys,re
s=sys.stdin.read(); T=$T
v={k:float(x) for k,x in re.findall(r'(\w+)\s+([0-9.]+)', s)}
c=v.get('copy',0)
print(' '.join('%s %.1f (%.4f us/step)' % (k, v[k]-(c if 'bwd' in k else 0), (v[k]-(c if 'bwd' in k else 0))/T) for k in ('new_fwd_z0','new_fwd_z1','new_bwd_z0','new_bwd_z1') if k in v))" >> $O
  done
done
cat $O
