#!/usr/bin/env python3
"""Ordered kernel sequence of the LAST step in a rocprofv3 kernel-trace .db: start offset, duration, gap to the
previous kernel on the same queue.  usage: python scripts/kernel_sequence.py t_results.db [first_kernel_name]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "prepare_points_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, {qcol or 0}, (grid_x/workgroup_x)*(grid_y/workgroup_y)*(grid_z/workgroup_z), workgroup_x from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if first in r[0]]
lo = starts[-1]
rows = rows[lo:]
# what follows the step's last kernel (infomat_finalize_kernel) is bench.py's parity gate reading the results back: not the step
ends = [i for i, r in enumerate(rows) if "infomat_finalize_kernel" in r[0]]
if ends:
    rows = rows[:ends[-1] + 1]
t0 = rows[0][1]
last_end = {}
tot = gap_tot = 0
print("| # | queue | start us | dur us | gap us | WGs x block | kernel |\n|---|---|---|---|---|---|---|")
for i, (name, s, e, q, wgs, blk) in enumerate(rows):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*$", "", short)[:64]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0)
    print(f"| {i} | {q} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | {wgs}x{blk} | {short} |")
print(f"\n{len(rows)} dispatches, kernel time {tot:.0f} us, gaps {gap_tot:.0f} us, span {(rows[-1][2] - t0) / 1e3:.0f} us")
