import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests", "golden"))
import torch, time
from raw_scan import raw_scan
from deeppointmap_amd import preprocess as P
x = raw_scan().cuda()
for _ in range(3):
    pts, pad = P.preprocess_scan(x, outlier=(10, 3.0), lowpass=(0.5, 16, 2.0, 4))
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10):
    pts, pad = P.preprocess_scan(x, outlier=(10, 3.0), lowpass=(0.5, 16, 2.0, 4))
torch.cuda.synchronize(); print("full chain ms/scan", (time.perf_counter()-t)*100, pts.shape)
