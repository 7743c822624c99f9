#!/bin/bash
# r05: where do the sampling rounds' bucket re-reads hit?  fps_bucket_kernel alone at B frames per launch (one 16-wave workgroup
# per frame, frames dealt round-robin to the 8 XCDs: B / 8 frames of 1.31 MB share one 4 MB L2): kernel-trace pass for the
# duration, one --pmc pass for TCC_HIT / TCC_MISS / TCC_EA0_RDREQ.  -> gpurun_out/r05/fps_l2.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05/fps_l2; mkdir -p $out
echo "| frames per launch | frames per XCD | state per XCD L2 (MB) | fps_bucket_kernel ms | us per round | TCC hit | TCC miss | hit rate | EA read requests |" > $out/table.md
echo "|---|---|---|---|---|---|---|---|---|" >> $out/table.md
for B in ${@:-8 16 24 32 64}; do
  rocprofv3 --kernel-trace --stats -d $out/t$B -o t -- python scripts/fps_algo_bench.py $B 5 > $out/t$B.log 2>&1
  ms=$(python - <<PY
import sqlite3
db = sqlite3.connect("$out/t$B/t_results.db")
r = db.execute("select avg(duration) from kernels where name like '%fps_bucket_kernel%'").fetchone()
print(f"{r[0] / 1e6:.4f}")
PY
)
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $out/p$B -o p -- python scripts/fps_algo_bench.py $B 5 > $out/p$B.log 2>&1
  python - <<PY >> $out/table.md
import sqlite3
db = sqlite3.connect("$out/p$B/p_results.db")
c = {}
for kn, cn, nd, tot in db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name"):
    if "fps_bucket_kernel" in kn:
        c[cn] = tot / nd
h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
B, ms = $B, $ms
print(f"| {B} | {B / 8:g} | {B / 8 * 65536 * 20 / 1e6:.2f} | {ms:.3f} | {ms * 1e3 / 4095:.3f} | {h:.4g} | {m:.4g} | {h / max(h + m, 1):.3f} | {c.get('TCC_EA0_RDREQ_sum', 0):.4g} |")
PY
  rm -rf $out/t$B $out/p$B
done
cat $out/table.md
