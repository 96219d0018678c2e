#!/bin/bash
# Round-2 GPU trip M: kernel trace of a dilnet training step (where does the time go?)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/dil_train.py <<'PY'
import sys; sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo')
import bench_extra as B
B.bench_segfamily(models=('dilnet',), steps=4, warmup=2)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02m_prof -o dil -- python /tmp/dil_train.py ) > gpurun_out/r02m_rocprof.log 2>&1
python - <<'PY'
import sqlite3, re, collections, glob
db = sqlite3.connect(glob.glob('/root/repo/gpurun_out/r02m_prof/*.db')[0])
rows = list(db.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
agg = collections.OrderedDict()
for n, s, e, gx, gy in rows:
    n = re.sub(r'\(.*', '', n)[:64]
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
out = open('/root/repo/gpurun_out/r02m_dilnet_train_kernels.txt', 'w')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    line = f"{k:64s} calls {a[0]:5d} avg {a[1]/a[0]/1e3:9.1f} us  total {a[1]/1e6:8.2f} ms {100*a[1]/tot:5.1f}%"
    print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/r02m_prof
