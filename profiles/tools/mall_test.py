import torch, time
dev = "cuda:0"
def bench(nbuf, mb, iters=40):
    n = mb * 1024 * 1024 // 2
    bufs = [torch.randn(n, device=dev, dtype=torch.float32)[: n].to(torch.bfloat16) for _ in range(nbuf)]
    outs = [torch.empty_like(b) for b in bufs]
    torch.cuda.synchronize()
    for i in range(nbuf): torch.mul(bufs[i], 1.5, out=outs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(iters):
        i = it % nbuf
        torch.mul(bufs[i], 1.5, out=outs[i])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{nbuf:2d} buffer pair(s) of {mb} MB: {ms*1e3:7.1f} us per pass, {2*mb/1024/ms*1e3/1e3:6.2f} TB/s (footprint {2*mb*nbuf} MB)")
for mb in (16, 32, 64, 128):
    for nbuf in (1, 16):
        bench(nbuf, mb)
# producer -> consumer on the same buffer (write then read back)
n = 128 * 1024 * 1024 // 2
a = torch.randn(n, device=dev).to(torch.bfloat16); b = torch.empty_like(a); c = torch.empty_like(a)
big = torch.empty(1024 * 1024 * 1024, dtype=torch.uint8, device=dev)
for flush in (False, True):
    ts = []
    for _ in range(10):
        torch.mul(a, 1.5, out=b)
        if flush: big.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.mul(b, 1.5, out=c); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"consumer of a freshly written 128 MB tensor, flush between = {flush}: {ts[len(ts)//2]*1e3:.1f} us")
