import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py']
import bench, argparse
ap = argparse.ArgumentParser()
for k, v in dict(size=128, accum=16, copt=16, dtype='bf16', seed=0, impl=0).items():
    ap.add_argument(f'--{k}', type=type(v), default=v)
args, _ = ap.parse_known_args()
for k in ('gpus', 'steps', 'warmup'):
    setattr(args, k, 1)
r = bench.EpochRunner(args, torch.device('cuda:0'), 0)
r.epoch(); r.epoch()
torch.cuda.synchronize()
print(f"peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB")
