"""Timeline of the chunked predict pipeline from a rocprofv3 --kernel-trace --memory-copy-trace database (dev tool):
per stream busy time, gaps of the compute stream, overlap of the transfers with compute.
   usage: python tools/gpu_predict_timeline.py <results.db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
ks = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
print(len(ks), "kernels")
try:
    cps = list(db.execute("select name, start, end from memory_copies order by start"))
except Exception as e:
    cps = []
    print("no memory_copies view:", e, [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')") if 'cop' in r[0].lower()])
t0 = min(k[1] for k in ks)
by_stream = collections.defaultdict(list)
for n, s, e, st, q in ks:
    by_stream[(st, q)].append((s - t0, e - t0, n))
for key, v in by_stream.items():
    busy = sum(e - s for s, e, _ in v)
    span = v[-1][1] - v[0][0]
    names = collections.Counter(n.split("(")[0][:40] for _, _, n in v).most_common(3)
    print(f"stream {key}: {len(v)} kernels, busy {busy/1e6:.1f} ms over a span of {span/1e6:.1f} ms; top: {names}")
# compute stream = the one with most kernels; gaps > 50 us
main = max(by_stream.values(), key=len)
main = main[len(main) // 4:]                      # skip the warm-up quarter
gaps = [(main[i + 1][0] - main[i][1], main[i][2][:40], main[i + 1][2][:40]) for i in range(len(main) - 1)]
big = [g for g in gaps if g[0] > 50e3]
print(f"compute stream (steady part): {len(main)} kernels, busy {sum(e - s for s, e, _ in main)/1e6:.1f} ms, span {(main[-1][1] - main[0][0])/1e6:.1f} ms, "
      f"gaps > 50 us: {len(big)} totalling {sum(g[0] for g in big)/1e6:.1f} ms")
for i in range(len(main) - 1):
    g = main[i + 1][0] - main[i][1]
    if g > 50e3:
        print(f"   gap of {g/1e6:7.2f} ms at t = {main[i][1]/1e6:8.1f} ms after {main[i][2][:24]!r}")
chunk_starts = [s for s, e, n in main if n.startswith("sub_div")]
print("chunk starts (ms):", [round(x / 1e6, 1) for x in chunk_starts])
px_ends = [e for s, e, n in by_stream[max(by_stream, key=lambda k: len(by_stream[k]))] if n.startswith("px_fwd")]
for key, v in by_stream.items():
    if len(v) < 50:
        for s_, e_, n in v:
            prev = max([x for x in px_ends if x <= s_] or [0])
            print(f"   {n[:14]:14s} start {s_/1e6:8.1f} end {e_/1e6:8.1f} ms ({(e_-s_)/1e6:5.2f} ms), {(s_-prev)/1e6:6.1f} ms after the last px_fwd end")
agg = collections.Counter()
for g in big:
    agg[(g[1], g[2])] += g[0]
for (a, b), t in agg.most_common(8):
    print(f"   {t/1e6:8.2f} ms between {a!r} and {b!r}")
if cps:
    print(len(cps), "copies, total", sum(e - s for _, s, e in cps) / 1e6, "ms; first few:", [(n, (e - s) / 1e3) for n, s, e in cps[:6]])
