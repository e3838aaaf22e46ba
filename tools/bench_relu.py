import sys, os
sys.path.insert(0, "/root/repo")
import torch
from erd_amd import kernels as K
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (n, h, w, c) in [(4,100,168,512),(4,50,84,1024),(4,25,42,2048),(4,100,168,256)]:
    y = torch.randn(n,h,w,c,device="cuda"); dy = torch.randn_like(y)
    cs = torch.zeros(c, device="cuda")
    t = timeit(lambda: K.relu_bwd_colsum(y, dy, True, True, colsum_into=cs))
    print(f"{n*h*w:7d} px x {c:5d}: {t:7.1f} us  {3*y.numel()*4/t/1e6:6.2f} TB/s")
