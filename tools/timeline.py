"""Timeline statistics of a rocprofv3 kernel trace (rocpd database) of bench.py under hipGraph replay: per queue (stream) the
number of kernels, summed kernel time, covered span and the gaps between consecutive kernels -- i.e. how much of the wall time
of one image is kernels and how much is launch boundaries.

    python tools/timeline.py <kt_results.db> [n_tail_kernels]     # analyses the last n kernels (default: last 9000 ~ one image)"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
c = sqlite3.connect(db)
rows = c.execute("select start, end, queue_id, name from kernels order by start").fetchall()     # every kernel, torch's own included
# the window ends with the last image's uint8 conversion (what follows is the benchmark's own bookkeeping, not the pipeline)
ends = [i for i, r in enumerate(rows) if "to_u8" in r[3]]
if ends:
    rows = rows[:ends[-1] + 1]
rows = rows[-tail:]
print(f"of which not libsdeo's: {sum(1 for r in rows if 'sdeo' not in r[3])} kernels, {sum(r[1] - r[0] for r in rows if 'sdeo' not in r[3]) / 1e6:.2f} ms")
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms")
byq = defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    gaps_small = sorted(g for g in gaps if 0 <= g < 50_000)
    neg = sum(1 for g in gaps if g < 0)
    med = gaps_small[len(gaps_small) // 2] if gaps_small else 0
    print(f"queue {q}: {len(ks)} kernels, kernel time {busy / 1e6:.2f} ms, span {(ks[-1][1] - ks[0][0]) / 1e6:.2f} ms, "
          f"gaps<50us: n={len(gaps_small)} sum {sum(gaps_small) / 1e6:.2f} ms median {med / 1e3:.2f} us mean {sum(gaps_small) / max(len(gaps_small), 1) / 1e3:.2f} us; "
          f"overlapping launches {neg}; long gaps (>=50us) {sum(1 for g in gaps if g >= 50_000)} sum {sum(g for g in gaps if g >= 50_000) / 1e6:.2f} ms")
# union coverage: time with at least one kernel running / with two or more
ev = []
for s, e, q, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, one, two = 0, ev[0][0], 0, 0
for t, d in ev:
    if cur >= 1: one += t - last
    if cur >= 2: two += t - last
    cur += d; last = t
print(f"wall {(t1 - t0) / 1e6:.2f} ms: >=1 kernel running {one / 1e6:.2f} ms ({100 * one / (t1 - t0):.1f} %), >=2 running {two / 1e6:.2f} ms, idle {(t1 - t0 - one) / 1e6:.2f} ms")
# gap after each kernel type (who is followed by the longest bubbles)
after = defaultdict(lambda: [0, 0])
for q, ks in byq.items():
    for i in range(len(ks) - 1):
        g = ks[i + 1][0] - ks[i][1]
        if 0 <= g < 50_000:
            nm = ks[i][2].split("(")[0].replace("void ", "").replace("sdeo::", "")[:60]
            after[nm][0] += g; after[nm][1] += 1
for nm, (g, n) in sorted(after.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  gap after {nm:60s} n={n:5d} mean {g / n / 1e3:5.2f} us total {g / 1e6:6.2f} ms")

# chip-wide idle intervals (no kernel of any queue running): which kernels stand on either side of the long ones
def short(n):
    return n.split("(")[0].replace("void ", "").replace("sdeo::", "")[:44]
ends = sorted(rows, key=lambda r: r[0])
idle = []
cur_end, cur_name = ends[0][1], ends[0][3]
for s_, e_, q_, n_ in ends[1:]:
    if s_ > cur_end:
        idle.append((s_ - cur_end, cur_name, n_))
    if e_ > cur_end:
        cur_end, cur_name = e_, n_
agg = defaultdict(lambda: [0, 0])
for g, a, b in idle:
    agg[(short(a), short(b))][0] += g; agg[(short(a), short(b))][1] += 1
print(f"chip-wide idle: {len(idle)} intervals, {sum(g for g, _, _ in idle) / 1e6:.2f} ms; by (kernel before -> kernel after):")
for (a, b), (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {a:44s} -> {b:44s} n={n:5d} mean {g / n / 1e3:7.2f} us total {g / 1e6:6.2f} ms")
