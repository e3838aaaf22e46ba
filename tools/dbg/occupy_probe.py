"""why does ANY resident spin kernel slow the step 3.4x?  step times with occupiers of different shapes"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
src = r'''
#include <hip/hip_runtime.h>
__global__ void occ(long long cycles, unsigned long long* sink) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) { acc += 1; __builtin_amdgcn_s_sleep(32); }
    if (acc == 0xffffffffffffull) sink[0] = acc;
}
extern "C" int occupy(int blocks, int threads, int lds, long long cycles, void* sink, void* stream) {
    hipLaunchKernelGGL(occ, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, cycles, (unsigned long long*)sink);
    return (int)hipGetLastError();
}
'''
open("/tmp/occ.hip", "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "/tmp/occ.hip", "-o", "/tmp/occ.so"], check=True)
lib = ctypes.CDLL("/tmp/occ.so")
lib.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
import bench
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0, "r50_40_40")
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay, base_batch_size=cfg.auto_scale_lr.base_batch_size,
                batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg, num_new=40, H=bench.H, W=bench.W) for i in range(2)]
for j in range(4): tr.train_step(*batches[j % 2], next_batch=batches[(j + 1) % 2])
tr.flush(); torch.cuda.synchronize()
side = torch.cuda.Stream(); sink = torch.zeros(1, dtype=torch.int64, device=dev)
def run(tag, blocks, threads, lds):
    torch.cuda.synchronize()
    if blocks: lib.occupy(blocks, threads, lds, int(1.5 * 2.0e9), sink.data_ptr(), side.cuda_stream)
    ts = []
    for j in range(8):
        t0 = time.perf_counter()
        tr.train_step(*batches[j % 2], next_batch=batches[(j + 1) % 2])
        torch.cuda.current_stream().synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    held = (not side.query()) if blocks else None
    torch.cuda.synchronize()
    print(f"{tag:44s} ms per step: " + " ".join("%.1f" % t for t in ts) + f"   still held: {held}", flush=True)
run("no occupier", 0, 0, 0)
run("1 workgroup x 64 threads, no LDS", 1, 64, 0)
run("8 workgroups x 64 threads, no LDS", 8, 64, 0)
run("8 workgroups x 1024 threads, no LDS", 8, 1024, 0)
run("8 workgroups x 1024 threads, 64 KB LDS", 8, 1024, 65536)
run("32 workgroups x 1024 threads, 64 KB LDS", 32, 1024, 65536)
run("no occupier", 0, 0, 0)
