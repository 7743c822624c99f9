"""Which dispatches of a rocprofv3 results .db belong to the measured steps: everything from the first FULL-BATCH first-level
sampling launch on (bench.py's initialisation runs a two-frame step first; its dispatches diluted every "avg us" and "bytes per call"
of rounds 1-5: one number per quantity across profiles/ needs them out).  Used by rocprof_summary.py, evidence_table.py,
rocprof_pmc.py, stage_table.py."""
import sqlite3


def _cols(db, table):
    return [r[1] for r in db.execute(f"pragma table_info('{table}')")]


def steady_start(db: sqlite3.Connection):
    """start timestamp of the first sampling launch with the largest grid, or None when the trace has no such kernel"""
    if "kernels" not in [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]:
        return None
    rows = db.execute("select start, grid_x * 1.0 / workgroup_x from kernels where name like '%fps_bucket_kernel%' order by start").fetchall()
    if not rows:
        return None
    full = max(g for _, g in rows)
    return min(s for s, g in rows if g == full)


def steady_clause(db: sqlite3.Connection, table: str = "kernels") -> str:
    """a SQL condition (with leading 'and') that keeps the measured steps' rows of `table` ('kernels' or 'pmc_events')"""
    t0 = steady_start(db)
    if t0 is None:
        return ""
    if table == "kernels":
        return f" and start >= {t0}"
    cols = _cols(db, table)
    if "dispatch_id" in cols and "dispatch_id" in _cols(db, "kernels"):
        return f" and dispatch_id in (select dispatch_id from kernels where start >= {t0})"
    if "start" in cols:
        return f" and start >= {t0}"
    return ""
