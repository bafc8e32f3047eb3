"""Upper-bound experiment: two INDEPENDENT TTA instances (own weights, own streams, one Python thread each) on one GPU versus
the same two instances one after the other - how much do the HBM-bound passes of one hide behind the MFMA-bound kernels of
the other when the hardware schedules two whole passes freely?"""
import os, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = types.SimpleNamespace(size=128, accum=16, copt=16, dtype="bf16", gpus=1, impl=0)
dev = torch.device("cuda:0")
runners = [bench.EpochRunner(args, dev, r) for r in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
N = 3
for r, s in zip(runners, streams):
    with torch.cuda.stream(s):
        r.epoch()
torch.cuda.synchronize()

t0 = time.perf_counter()
for r, s in zip(runners, streams):
    with torch.cuda.stream(s):
        for _ in range(N):
            r.epoch()
torch.cuda.synchronize()
seq = time.perf_counter() - t0

def work(r, s):
    with torch.cuda.stream(s):
        for _ in range(N):
            r.epoch()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(r, s)) for r, s in zip(runners, streams)]
[t.start() for t in th]
[t.join() for t in th]
torch.cuda.synchronize()
par = time.perf_counter() - t0
print(f"2 x {N} epochs one after the other: {seq*1e3/(2*N):.1f} ms per epoch; two threads / streams: {par*1e3/(2*N):.1f} ms per epoch")
