import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import test_gpu_bucket_update_ab as T
def d(a, b): return [(float((x[1]-y[1]).double().norm()), float((x[2]-y[2]).double().norm())) for x, y in zip(a, b)]
ref = T._run(False, 1, 4); twin = T._run(False, 1, 4)
print("default/no-update twin      ", d(ref, twin))
s = torch.cuda.Stream()
a0 = T._run(False, 1, 4, stream=s); a0b = T._run(False, 1, 4, stream=s)
print("stream/no-update vs ref     ", d(ref, a0))
print("stream/no-update twin       ", d(a0, a0b))
a1 = T._run(True, 1, 4, stream=s); a1b = T._run(True, 1, 4, stream=s)
print("stream/update vs stream/no  ", d(a0, a1))
print("stream/update twin          ", d(a1, a1b))
print("stream/update vs ref        ", d(ref, a1))
b1 = T._run(True, 1, 4)
print("default/update vs ref       ", d(ref, b1))
