"""Launch floor of a linear chain of tiny kernels: eager vs hipGraph replay (us per kernel)."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(256 * 256 * 8, device=dev)
def chain(t, n):
    for _ in range(n):
        t.add_(1.0)
for name, t in (("64 elems", x), ("512k elems", big)):
    n = 600
    chain(t, n); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(t, n); torch.cuda.synchronize(); te = time.perf_counter() - t0
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(t, 10); 
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(t, n)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-12s eager %.2f us/kernel   graph %.2f us/kernel" % (name, te / n * 1e6, e0.elapsed_time(e1) / 5 / n * 1e3), flush=True)
