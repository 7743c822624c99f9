"""Per-kernel LDS picture of one unpipelined step from a rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE): python scripts/lds_conflicts.py p_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = {}
for kn, cn, nd, tot in db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name"):
    rows.setdefault(kn, {})[cn] = (tot, nd)
print("| kernel | launches | LDS-active cycles per CU / kernel cycles | bank conflicts / LDS-active | wave life waiting for LDS |\n|---|---|---|---|---|")
out = []
for kn, c in rows.items():
    g = c.get("GRBM_GUI_ACTIVE", (0, 1)); a = c.get("SQ_LDS_IDX_ACTIVE", (0, 1)); b = c.get("SQ_LDS_BANK_CONFLICT", (0, 1))
    w = c.get("SQ_WAIT_INST_LDS", (0, 1)); wc = c.get("SQ_WAVE_CYCLES", (1, 1))
    if a[0] <= 0:
        continue
    busy = (a[0] / 256) / (g[0] / 8) if g[0] else 0
    out.append((a[0], f"| {kn[:70]} | {g[1]} | {busy:.2f} | {b[0] / a[0]:.2f} | {w[0] / max(wc[0], 1):.2f} |"))
for _, l in sorted(out, reverse=True):
    print(l)
