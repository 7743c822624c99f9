"""Per-kernel mean of every counter in rocprofv3 --pmc sqlite dbs: python scripts/pmc_dump.py <substr> db [db ...]"""
import sqlite3, sys
sub = sys.argv[1]
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    q = ("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events "
         "group by name, counter_name")
    for kn, cn, nd, tot in db.execute(q):
        if sub in kn:
            print(f"{kn[:48]:48s} {cn:34s} {tot / nd:16.1f}  ({nd} launches)")
