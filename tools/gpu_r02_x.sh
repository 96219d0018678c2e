#!/bin/bash
# Round-2 GPU trip X: kernel trace of the rVAE training step (config 4): what is there besides the decoder kernels?
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/rvae.py <<'PY'
import sys; sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo')
import bench_extra as B
B.bench_rvae(steps=6, warmup=3)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02x_prof -o rvae -- python /tmp/rvae.py ) > gpurun_out/r02x_rocprof.log 2>&1
python - <<'PY'
import sqlite3, re, collections, glob
db = sqlite3.connect(glob.glob('/root/repo/gpurun_out/r02x_prof/*.db')[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
# steady part: last 6 of 9 steps -> take the last two thirds of the kernels
rows = rows[len(rows) // 3:]
agg = collections.OrderedDict()
for n, s, e in rows:
    n = re.sub(r'\(.*', '', n)[:70]
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values()); span = rows[-1][2] - rows[0][1]
out = open('/root/repo/gpurun_out/r02x_rvae_kernels.txt', 'w')
hdr = f"steady part: {len(rows)} kernels, busy {tot/1e6:.2f} ms over a span of {span/1e6:.2f} ms (6 steps)"
print(hdr); out.write(hdr + "\n")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    line = f"{k:70s} calls {a[0]:5d} avg {a[1]/a[0]/1e3:9.1f} us  per step {a[1]/6e3:9.1f} us {100*a[1]/tot:5.1f}%"
    print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/r02x_prof
