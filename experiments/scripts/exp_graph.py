"""Eager vs hipGraph replay of a small coronagraph-like chain (512^2, Q=2): time per model evaluation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import propagation as P, graph
from prysm_amd.conf import config

for prec, n in ((32, 256), (32, 512), (64, 512), (32, 1024)):
    config.precision = prec
    rdt = np.float32 if prec == 32 else np.float64
    rng = np.random.default_rng(1)
    amp = (rng.random((n, n)) > 0.2).astype(rdt)
    opd = (rng.standard_normal((n, n)) * 40).astype(rdt)

    def model(a, o):
        wf = P.Wavefront.from_amp_and_phase(a, o, 0.6328, 10.0 / n)
        E = wf.focus(100.0, Q=2)
        back = E.unfocus(100.0, Q=1)
        again = back.focus(100.0, Q=1)
        return again.intensity.data

    a_d, o_d = torch.from_numpy(amp).cuda(), torch.from_numpy(opd).cuda()
    for _ in range(5): model(a_d, o_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): out = model(a_d, o_d)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 200
    m = graph.capture(model, a_d, o_d)
    for _ in range(5): m.graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): m.graph.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 200
    ok = torch.equal(m(a_d, o_d), model(a_d, o_d))
    print(f'precision {prec} n={n}: 5-kernel-pair chain eager {t_eager * 1e6:.1f} us, hipGraph replay {t_graph * 1e6:.1f} us, identical={ok}', flush=True)
config.precision = 64
