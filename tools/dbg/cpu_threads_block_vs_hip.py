import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, torch.nn.functional as F
import golden_inputs as G
from erd_amd.modules import Bottleneck

def run(nt, cin=1024, planes=512, stride=2, down=True):
    torch.set_num_threads(nt)
    N, H, W = 2, 12, 14
    blk = Bottleneck(cin, planes, stride, down)
    sd = {}
    for i, (k, v) in enumerate(blk.state_dict().items()):
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked": sd[k] = v
        elif leaf == "running_var" or (leaf == "weight" and v.dim() == 1): sd[k] = 0.5 + G.rand(100 + i, *v.shape)
        elif v.dim() == 4: sd[k] = G.randn(100 + i, *v.shape, scale=(2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5)
        else: sd[k] = G.randn(100 + i, *v.shape, scale=0.2)
    blk.load_state_dict(sd)
    blk = blk.cuda()
    x = G.randn(90, N, cin, H, W).requires_grad_(True)
    ref = {k: v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v for k, v in sd.items()}
    bn = lambda t, p: F.batch_norm(t, ref[p + ".running_mean"], ref[p + ".running_var"], ref[p + ".weight"], ref[p + ".bias"], False, 0.0, 1e-5)
    o = F.relu(bn(F.conv2d(x, ref["conv1.weight"]), "bn1"))
    o = F.relu(bn(F.conv2d(o, ref["conv2.weight"], None, stride, 1), "bn2"))
    o = bn(F.conv2d(o, ref["conv3.weight"]), "bn3")
    idn = bn(F.conv2d(x, ref["downsample.0.weight"], None, stride), "downsample.1")
    y = F.relu(o + idn)
    dy = G.randn(91, *y.shape)
    y.backward(dy)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    out = blk(xg)
    out.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    return dict(cpu_y=y.detach(), cpu_dx=x.grad.clone(), hip_y=out.detach().permute(0, 3, 1, 2).cpu(), hip_dx=xg.grad.permute(0, 3, 1, 2).cpu(),
                sd={k: v.clone() for k, v in sd.items() if v.dtype == torch.float32})

rel = lambda a, b: float((a - b).norm() / b.norm())
a, b = run(128), run(16)
for k in ("cpu_y", "cpu_dx", "hip_y", "hip_dx"):
    print(k, "16 vs 128 threads:", rel(b[k], a[k]))
print("state dict differs:", [k for k in a["sd"] if not torch.equal(a["sd"][k], b["sd"][k])][:5])
for nt, d in ((128, a), (16, b)):
    print(nt, "hip vs cpu: y", rel(d["hip_y"], d["cpu_y"]), "dx", rel(d["hip_dx"], d["cpu_dx"]))
