"""Static instruction counts of one kernel in a built library: python scripts/isa_count.py <lib.so> <kernel-name-substring>.
(vector / scalar / DPP / memory instructions of the device code; a first reading before the counters on the GPU)"""
import collections, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeppointmap_amd", "csrc"))
import isa_lint
obj, pat = sys.argv[1], sys.argv[2]
s = "\n".join(isa_lint.device_disassembly(obj))
for f in re.split(r"\n(?=[0-9a-f]+ <)", s):
    m = re.match(r"[0-9a-f]+ <([^>]+)>", f)
    if not m or pat not in m.group(1):
        continue
    lines = [l.strip() for l in f.splitlines() if re.match(r"\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)", l)]
    c = collections.Counter(l.split()[0] for l in lines)
    print(f"{m.group(1)[:90]}: {len(lines)} instructions, vector {sum(v for k, v in c.items() if k.startswith('v_'))}, scalar {sum(v for k, v in c.items() if k.startswith('s_'))}, "
          f"dpp {sum(1 for l in lines if 'quad_perm' in l or 'row_' in l)}, global/buffer loads {sum(v for k, v in c.items() if 'load' in k and not k.startswith('s_'))}, lds {sum(v for k, v in c.items() if k.startswith('ds_'))}")
