#!/bin/bash
# rocprofv3 --kernel-trace of the U-Net training step (bs 32, 512^2): per-kernel start / end of ONE steady step with its
# hardware queue, so that the overlap of the side stream (weight gradients) with the main stream is visible.
#   usage: tools/gpu_step_timeline.sh <tag> [ENV=VAL ...]      -> gpurun_out/r06_step_timeline_<tag>.txt
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cat > /tmp/step.py <<'PY'
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, atomai_amd as aoi
rs = np.random.RandomState(0)
X = rs.rand(64, 512, 512).astype(np.float32); y = rs.randint(0, 3, (64, 512, 512))
m = aoi.models.Segmentor("Unet", nb_classes=3, seed=1)
m.compile_trainer((X, y, X[:32], y[:32]), training_cycles=10, batch_size=32)
for i in range(8): m.train_step(m.X_train[i % 2], m.y_train[i % 2])
torch.cuda.synchronize()
PY
rm -rf gpurun_out/prof_tl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof_tl -o tl -- python /tmp/step.py ) > gpurun_out/prof_tl.log 2>&1
python - "$TAG" "$*" <<'PY' > gpurun_out/r06_step_timeline_$TAG.txt
import csv, glob, sys
f = glob.glob('/root/repo/gpurun_out/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
adam = [i for i, k in enumerate(ks) if 'adam_flat' in k[2]]
a0, a1 = adam[-2], adam[-1]
step = ks[a0 + 1:a1 + 1]
t0 = step[0][0]
print(f"# tag {sys.argv[1]}  env: {sys.argv[2]}")
print(f"# one steady step: {(step[-1][1] - t0) / 1e3:.1f} us from the first kernel's start to the end of adam; columns: start us, duration us, queue, kernel")
busy = {}
for s, e, n, q in step:
    short = n.split('(')[0].replace('void ', '')[:60]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  q{q:>3s}  {short}")
    busy[q] = busy.get(q, 0) + (e - s)
print("# busy per queue (us):", {q: round(v / 1e3, 1) for q, v in busy.items()})
PY
rm -rf gpurun_out/prof_tl
tail -3 gpurun_out/r06_step_timeline_$TAG.txt
