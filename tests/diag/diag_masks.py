import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.nn.functional as F
import golden_inputs as G
from oracle import erd_oracle as O
import erd_amd
from erd_amd import MODELS
from erd_amd.modules import Bottleneck
sd_full = O.procedural_state_dict(40, seed=0)
sd = {k[len('backbone.'):]: v for k, v in sd_full.items() if k.startswith('backbone.')}
m = MODELS.build(dict(type='ResNet', depth=50, frozen_stages=1, norm_eval=True))
m.load_state_dict(sd, strict=True); m = m.cuda().train()
x = G.randn(1, 2, 3, 128, 160)
# CPU reference activations per conv-bn(-relu)
def bn(x, p): return F.batch_norm(x, sd[p+'.running_mean'], sd[p+'.running_var'], sd[p+'.weight'], sd[p+'.bias'], False, 0., 1e-5)
ref = {}
h = F.max_pool2d(F.relu(bn(F.conv2d(x, sd['conv1.weight'], None, 2, 3), 'bn1')), 3, 2, 1)
for li, nb in enumerate((3,4,6,3)):
    for b in range(nb):
        p = f'layer{li+1}.{b}'; s = 2 if (b==0 and li>0) else 1
        o1 = F.relu(bn(F.conv2d(h, sd[p+'.conv1.weight']), p+'.bn1')); ref[p+'.1'] = o1
        o2 = F.relu(bn(F.conv2d(o1, sd[p+'.conv2.weight'], None, s, 1), p+'.bn2')); ref[p+'.2'] = o2
        o3 = bn(F.conv2d(o2, sd[p+'.conv3.weight']), p+'.bn3')
        idn = h if b else bn(F.conv2d(h, sd[p+'.downsample.0.weight'], None, s), p+'.downsample.1')
        h = F.relu(o3 + idn); ref[p+'.3'] = h
# GPU activations via hooks on my Function outputs
got = {}
orig = Bottleneck._cba
import erd_amd.functional as Fn
counter = {}
def fwd(self, xx):
    name = self._name
    out = orig(xx, self.conv1, self.bn1, None, True); got[name+'.1'] = out
    out = orig(out, self.conv2, self.bn2, None, True); got[name+'.2'] = out
    idn = xx
    if self.downsample is not None: idn = orig(xx, self.downsample[0], self.downsample[1], None, False)
    y = orig(out, self.conv3, self.bn3, idn, True); got[name+'.3'] = y
    return y
for n, mod in m.named_modules():
    if isinstance(mod, Bottleneck): mod._name = n
Bottleneck.forward = fwd
with torch.no_grad(): m(x.cuda())
tot = 0
for k in ref:
    a = got[k].permute(0,3,1,2).cpu(); b = ref[k]
    mism = int(((a>0) != (b>0)).sum())
    tot += mism
    print(k, 'relerr %.1e' % float((a-b).abs().max()/b.abs().max()), 'mask mismatches', mism, 'of', b.numel())
print('total mismatches', tot)
