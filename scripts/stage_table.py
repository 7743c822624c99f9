#!/usr/bin/env python3
"""Stage table of one unpipelined step: kernel-time sums of the geometry (G), feature (F) and registration
(R) stages, plain, weighted by each kernel's matrix-pipe busy fraction and by its vector-ALU busy fraction -- so that "sum
against max" (profiles/r04_corun.md) can be read off one table.
usage: stage_table.py <trace.db> <pmc_mfma.db> <pmc_valu.db>"""
import re
import sqlite3
import sys

trace, mfma, valu = (sqlite3.connect(p) for p in sys.argv[1:4])
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", n))[:64]


def frac(db, ctr, scale):
    """kernel -> busy fraction: sum(counter) * scale / (GRBM_GUI_ACTIVE cycles * 1024 SIMDs), over the run's dispatches"""
    val = {short(n): v for n, v in db.execute("select name, sum(counter_value) from pmc_events where counter_name=? group by name", (ctr,))}
    out = {}
    for n, d, c, v in db.execute("select name, count(distinct dispatch_id), count(*), sum(counter_value) from pmc_events "
                                 "where counter_name='GRBM_GUI_ACTIVE' group by name"):
        k = short(n)
        gui = v / max(c / d, 1)
        if gui > 0 and k in val:
            out[k] = min(1.0, val[k] * scale / (gui * 1024))
    return out


MF = frac(mfma, "SQ_VALU_MFMA_BUSY_CYCLES", 1.0)
VA = frac(valu, "SQ_ACTIVE_INST_VALU", 4.0)   # quad-cycles -> cycles
rows = trace.execute("select name, start, end from kernels order by start").fetchall()
lo = [i for i, r in enumerate(rows) if "prepare_points_kernel" in r[0]][-1]
rows = [(short(n), (e - s) / 1e3) for n, s, e in rows[lo:]]
r_from = next(i for i, (n, _) in enumerate(rows) if n.startswith("posemb_kernel"))
GEO = ("prepare_points", "str_", "fps_", "nested_levels", "knn_grid_build", "grid_bounds", "grid_setup", "grid_rows", "grid_offsets",
       "grid_cells", "count_valid")
st = {s: [0.0, 0.0, 0.0, 0] for s in "GFR"}
per = {}
for i, (n, us) in enumerate(rows):
    base = n.replace("void ", "")
    s = "G" if base.startswith(GEO) else ("R" if i >= r_from else "F")
    m, v = MF.get(n, 0.0), VA.get(n, 0.0)
    st[s][0] += us; st[s][1] += us * m; st[s][2] += us * v; st[s][3] += 1
    p = per.setdefault((s, n), [0.0, 0, m, v]); p[0] += us; p[1] += 1
print("| stage | dispatches | kernel time us | x matrix-pipe busy fraction (us) | x vector-ALU busy fraction (us) |\n|---|---|---|---|---|")
for s, name in (("G", "geometry: staging, sampling, search grids (own streams)"), ("F", "features: searches, gathers, GEMMs -> descriptors"),
                ("R", "registration: decoder, match, Kabsch, information matrices")):
    t, m, v, c = st[s]
    print(f"| {s} ({name}) | {c} | {t:.0f} | {m:.0f} | {v:.0f} |")
t = [sum(st[s][k] for s in "FR") for k in range(3)]
print(f"| F + R | | {t[0]:.0f} | {t[1]:.0f} | {t[2]:.0f} |")
print("\nA SIMD's time is matrix-busy + vector-busy + idle (profiles/r04_corun.md): the last two columns of the F + R row are the cycles "
      "no schedule can overlap, the rest of its kernel time is what co-residence can fill.\n")
print("| stage | kernel | calls | us | matrix-pipe busy % | vector-ALU busy % |\n|---|---|---|---|---|---|")
for (s, n), (us, c, m, v) in sorted(per.items(), key=lambda kv: (kv[0][0], -kv[1][0])):
    if us >= 20:
        print(f"| {s} | {n} | {c} | {us:.0f} | {100 * m:.0f} | {100 * v:.0f} |")
