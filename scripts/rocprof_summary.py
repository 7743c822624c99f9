#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (sqlite, --kernel-trace --stats) into a per-kernel summary table.
usage: python scripts/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_bench_kernels.md"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _steady import steady_clause

db = sqlite3.connect(sys.argv[1])
steady = steady_clause(db)   # the measured steps only: bench.py's two-frame initialisation step is left out
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(grid_x*1.0/workgroup_x), max(workgroup_x), max(vgpr_count), max(lds_size) "
                  f"from kernels where 1 {steady} group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print(f"| kernel | calls | total ms | avg us | min us | max us | % | max WGs | block | vgpr | lds B |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, n, tot, avg, mn, mx, wgs, blk, vg, lds in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*$", "", short)[:70]
    print(f"| {short} | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} | {wgs:.0f} | {blk} | {vg} | {lds} |")
print(f"\ntotal kernel time {total/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches" + (" (from the first full-batch sampling launch on)" if steady else ""))
