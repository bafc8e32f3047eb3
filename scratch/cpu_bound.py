"""Diagnostic: is an epoch CPU(enqueue)-bound?  Times epoch() up to its first host sync vs. the whole epoch."""
import inspect, sys, textwrap, time, types
sys.path.insert(0, '.')
sys.argv = ['bench.py']
import torch, bench
src = textwrap.dedent(inspect.getsource(bench.EpochRunner.epoch))
src = src.replace("    self.losses.append(", "    self.t_mid = time.perf_counter()\n    self.losses.append(")
assert "t_mid" in src
ns = dict(bench.__dict__)
exec(src, ns)
bench.EpochRunner.epoch = ns['epoch']
import argparse
ap = argparse.ArgumentParser()
for k, v in dict(size=128, accum=16, copt=16, dtype='bf16', seed=0, impl=0).items():
    ap.add_argument(f'--{k}', type=type(v), default=v)
args, _ = ap.parse_known_args()
for k in ('gpus', 'steps', 'warmup'):
    setattr(args, k, 1)
dev = torch.device('cuda:0')
r = bench.EpochRunner(args, dev, 0)
for _ in range(2):
    r.epoch()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); r.epoch(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"epoch {1e3*(t1-t0):.1f} ms; enqueue of the 16 accumulation steps + optimizer returned after {1e3*(r.t_mid-t0):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); r.epoch(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
