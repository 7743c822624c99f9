#!/usr/bin/env python3
"""Per-kernel evidence table: duration (kernel trace), L2-fabric bytes (PMC FETCH_SIZE / WRITE_SIZE passes) -> GB/s,
MFMA busy cycles -> utilisation.  usage: evidence_table.py <trace.db> <fetch.db> <write.db> <mfma.db> <steps_in_trace>"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _steady import steady_clause

trace, fetch, write, mfma = (sqlite3.connect(p) for p in sys.argv[1:5])
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", n))[:48]
dur = {short(n): (c, t) for n, c, t in trace.execute(f"select name, count(*), sum(duration) from kernels where 1 {steady_clause(trace)} group by name")}


def pmc(db, ctr):
    """name -> (dispatches in that PMC run, rows, summed value)"""
    return {short(n): (d, c, v) for n, d, c, v in db.execute(
        f"select name, count(distinct dispatch_id), count(*), sum(counter_value) from pmc_events where counter_name=? {steady_clause(db, 'pmc_events')} "
        "group by name", (ctr,))}


F, W = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
MB, GA, MO = pmc(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), pmc(mfma, "GRBM_GUI_ACTIVE"), pmc(mfma, "SQ_INSTS_VALU_MFMA_MOPS_F32")
n_disp = {k: v[0] for k, v in dur.items()}
print("| kernel | calls/trace | avg us | L2-fabric bytes/call (2*FETCH+WRITE) | GB/s | % of 8 TB/s | MFMA util % | fp32 MFMA TFLOP/s |")
print("|---|---|---|---|---|---|---|---|")
for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    if t / 1e6 < 0.3:
        continue
    avg_s = t / c / 1e9
    fd, _, fv = F.get(k, (1, 0, 0))
    wd, _, wv = W.get(k, (1, 0, 0))
    fbytes = (2 * fv / max(fd, 1) + wv / max(wd, 1)) * 1024   # per dispatch
    gbs = fbytes / avg_s / 1e9
    util = tf = ""
    if k in MB and MB[k][2] > 0:
        md, mrows, mbusy = MB[k]
        gd, grows, gsum = GA[k]
        gui = gsum / max(grows / gd, 1)          # GUI_ACTIVE cycles summed over the dispatches (one instance each)
        util = f"{100 * mbusy / (gui * 1024):.1f}"  # MfmaUtil: busy cycles / (active cycles * 1024 SIMDs)
        if k in MO:
            tf = f"{MO[k][2] * 512 / MO[k][0] / avg_s / 1e12:.1f}"
    print(f"| {k} | {c} | {t / c / 1e3:.1f} | {fbytes:.3g} | {gbs:.0f} | {100 * gbs / 8000:.1f} | {util} | {tf} |")
