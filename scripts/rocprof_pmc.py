#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 --pmc results .db.  usage: rocprof_pmc.py <db> [<db> ...]"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _steady import steady_clause

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('pmc_events')")]
    # join PMC samples to kernel names through the dispatch id
    q = ("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events where 1 " + steady_clause(db, "pmc_events") +
         " group by name, counter_name order by sum(counter_value) desc")
    try:
        rows = db.execute(q).fetchall()
    except sqlite3.OperationalError as e:
        print(path, "schema:", cols, e)
        continue
    print(f"## {path}")
    print("| kernel | counter | dispatches | sum | avg per dispatch |")
    print("|---|---|---|---|---|")
    for name, ctr, n, tot, avg in rows[:25]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*$", "", short)[:60]
        print(f"| {short} | {ctr} | {n} | {tot:.4g} | {avg:.4g} |")
