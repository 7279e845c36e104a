"""Throughput of a SEQUENCE of independent 4096^2 propagations on one stream against the same sequence alternating between two
streams (the drain of one field's column pass overlaps the fill of the next field's row pass).
usage: python tools/exp_two_streams.py [n]"""
import sys
import time

import numpy as np
import torch

from prysm_amd import propagation as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
xs = [torch.from_numpy((rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)).cuda() for _ in range(2)]
K = 400


def one_stream():
    f = None
    for i in range(K):
        f = None
        f = P.focus(xs[i & 1], 1)
    return f


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def two_streams():
    fs = [None, None]
    for i in range(K):
        with torch.cuda.stream(streams[i & 1]):
            fs[i & 1] = None
            fs[i & 1] = P.focus(xs[i & 1], 1)
    return fs


for rnd in range(3):
    for name, fn in (('one stream', one_stream), ('two streams', two_streams)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'n={n} {name:12s}: {dt / K * 1e6:7.2f} us per propagation  {K / dt:8.0f} /s', flush=True)
