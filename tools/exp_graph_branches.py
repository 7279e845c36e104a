"""Does a hipGraph with independent branches (captured across forked streams) run them concurrently on this ROCm?  Two / four chains of
small dependent kernels, captured on one stream and on forked streams; replay time of each (us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def chain(x, n):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x

def timed(g, reps=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for size, n in ((1 << 10, 40), (1 << 20, 40), (1 << 24, 10)):
    for nb in (2, 4):
        xs = [torch.rand(size, device='cuda') for _ in range(nb)]
        streams = [torch.cuda.Stream() for _ in range(nb)]
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for x in xs:
                chain(x, 2)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            outs1 = [chain(x, n) for x in xs]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cur = torch.cuda.current_stream()
            outs2 = []
            for x, s in zip(xs, streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs2.append(chain(x, n))
            for s in streams:
                cur.wait_stream(s)
        t1, t2 = timed(g1), timed(g2)
        same = all(torch.equal(a, b) for a, b in zip(outs1, outs2))
        print(f'{nb} chains of {n} kernels on {size} floats: one stream {t1:.1f} us, forked streams {t2:.1f} us, same results {same}', flush=True)
print('env', {k: v for k, v in os.environ.items() if 'GRAPH' in k or 'HIP_' in k})
