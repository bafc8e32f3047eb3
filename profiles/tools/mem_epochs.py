"""Peak / reserved device memory over N TTA epochs of the bench workload (checks that the side streams' deferred frees do
not make the caching allocator grow): mem_epochs.py [N]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
args = types.SimpleNamespace(size=128, accum=16, copt=16, dtype="bf16", gpus=1, impl=0)
torch.manual_seed(1234)
np.random.seed(1234)
r = bench.EpochRunner(args, torch.device("cuda:0"), 0)
for i in range(n):
    r.epoch()
    torch.cuda.synchronize()
    print(f"epoch {i}: loss {r.losses[-1]:.8f} dice {r.dice:.8f}  allocated {torch.cuda.memory_allocated()/2**30:6.2f} GiB  peak "
          f"{torch.cuda.max_memory_allocated()/2**30:6.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:6.2f} GiB", flush=True)
