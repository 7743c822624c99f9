#!/usr/bin/env python3
"""How much of a pipelined step has which stages' kernels running?  From a rocprofv3 kernel-trace .db of the pipelined
bench: per queue (= HIP stream) the busy time, and the time during which 0 / 1 / 2 / 3+ queues have a kernel running,
over the steady-state middle of the trace.  usage: python scripts/pipeline_overlap.py t_results.db"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute(f"select name, start, end, {qcol}, grid_x/workgroup_x * grid_y * grid_z from kernels order by start").fetchall()
marks = [r[1] for r in rows if "emit_descriptors" in r[0]]      # one per batch: the end of its feature stage
lo, hi = marks[len(marks) // 2 - 1], marks[-3]                 # steady state: skip initialisation / warm-up and the drain
steps = sum(1 for m in marks if lo <= m < hi)
rows = [r for r in rows if r[2] > lo and r[1] < hi]
rows = [(n, max(s, lo), min(e, hi), q, w) for n, s, e, q, w in rows]
span = (hi - lo) / 1e3
busy = collections.Counter()
names = collections.defaultdict(collections.Counter)
ev = []
for name, s, e, q, wgs in rows:
    busy[q] += (e - s) / 1e3
    names[q][re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", name))[:40]] += (e - s) / 1e3
    ev.append((s, 1, q)), ev.append((e, -1, q))
ev.sort()
act = collections.Counter()
hist = collections.Counter()
combo = collections.Counter()
prev = ev[0][0]
for t, d, q in ev:
    n = sum(1 for v in act.values() if v > 0)
    hist[min(n, 3)] += (t - prev) / 1e3
    combo[tuple(sorted(k for k, v in act.items() if v > 0))] += (t - prev) / 1e3
    act[q] += d
    prev = t
print(f"window {span:.0f} us, {steps} steps -> {span / max(steps, 1):.0f} us per step")
for q in sorted(busy):
    top = ", ".join(f"{k} {v / max(steps, 1):.0f}" for k, v in names[q].most_common(3))
    print(f"queue {q}: busy {busy[q] / max(steps, 1):7.0f} us per step ({100 * busy[q] / span:4.1f} % of the window)   top: {top}")
for n in range(4):
    print(f"{n}{'+' if n == 3 else ' '} queues active: {hist[n] / max(steps, 1):7.0f} us per step ({100 * hist[n] / span:4.1f} %)")
print("by combination of active queues (us per step):")
for c, v in combo.most_common(8):
    print(f"  {c}: {v / max(steps, 1):.0f}")
