#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh: per-dispatch KB of the
first-level sampling kernels (the kernels bench.py's `roofline` object describes: the packing's five kernels and the
round kernel), largest dispatch of each = the 64-frame one.
usage: make_pmc_traffic.py <fetch.db> <write.db> <frames_per_launch> > profiles/pmc_traffic.json"""
import json, sqlite3, sys
fetch, write, frames = sqlite3.connect(sys.argv[1]), sqlite3.connect(sys.argv[2]), int(sys.argv[3])
# the sampling rounds, and the five kernels of the Sort-Tile-Recursive packing that precede them (csrc/fps_tree.hip)
KERNELS = {"fps_bucket_kernel": ["fps_bucket_kernel"],
           "fps_str_sort_kernels": ["str_chunk_kernel<0>", "str_chunk_kernel<1>", "str_chunk_kernel<2>", "str_xoffsets_kernel",
                                    "str_ysort_kernel"]}


def per_dispatch(db, ctr, sub):
    rows = db.execute("select dispatch_id, sum(counter_value) from pmc_events where counter_name=? and name like ? group by dispatch_id",
                      (ctr, f"%{sub}%")).fetchall()
    return max(v for _, v in rows) if rows else None


out = {"_how": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-pipeline "
               "--no-extras ; same with WRITE_SIZE (separate passes; scripts/collect_profiles.sh).  Values are KB per dispatch "
               "(the 64-frame dispatch).  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: the x2 on FETCH_SIZE is the gfx950 correction "
               "of MI355X_MICROARCH.md (HBM section) for 16-byte-per-lane reads; WRITE_SIZE is uncalibrated.  The counters sit on "
               "the L2 fabric side, so Infinity-Cache hits are included.",
       "frames_per_launch": frames}
for key, subs in KERNELS.items():
    out[key] = {"FETCH_SIZE_KB": sum(per_dispatch(fetch, "FETCH_SIZE", sub) or 0 for sub in subs),
                "WRITE_SIZE_KB": sum(per_dispatch(write, "WRITE_SIZE", sub) or 0 for sub in subs)}
print(json.dumps(out, indent=2))
