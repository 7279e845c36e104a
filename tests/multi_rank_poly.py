"""1 or 2 ranks (WORLD_SIZE): the polychromatic driver's sharded paths (stacks and field-by-field, FFT and matrix-DFT variants; all-reduce,
reduce to root and the all-to-all reduce) against the oracle's single-process sum.  PM_TEST_BACKEND=nccl: one GPU per rank over
RCCL; gloo (default): both ranks on GPU 0.  Launch with torch.distributed.run --nproc-per-node 2 -- or 1 with PM_TEST_BACKEND=nccl:
a process group of ONE rank still runs every collective (RCCL init, reduce, all_to_all_single, gather, all_reduce), which is how a
one-GPU box executes the code path of the 8-GPU node.  Also: PsfPipeline (a frame's reduce on a side stream under the next
frame's transforms) against the same sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from oracle import prysm_oracle as O

rank = int(os.environ['RANK'])
backend = os.environ.get('PM_TEST_BACKEND', 'gloo')
dev = int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0
torch.cuda.set_device(dev)
if backend == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
else:
    dist.init_process_group('gloo')
from prysm_amd.polychromatic import polychromatic_psf, PsfPipeline
from prysm_amd.mathops import array_to_true_numpy as tonp

n = 128
x, y = O.make_xy_grid(n, diameter=10)
r, _ = O.cart_to_polar(x, y)
amp = O.circle(5, r)
opd = O.hopkins_w040(r / 5, 300.0)
dx = float(x[0, 1] - x[0, 0])
wvls = np.linspace(0.5, 0.7, 7)      # uneven shards: 4 + 3
wts = np.linspace(1.0, 2.0, 7)
comps = [O.intensity(O.focus(O.from_amp_and_phase(amp, opd, float(w)), 2)) for w in wvls]
want = O.sum_of_2d_modes(np.asarray(comps), wts)
worst = 0.0
for batched in (True, False):
    got = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=2, batched=batched))
    worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
# root-only results: one reduce, and the all-to-all of slices + ordered local sum + gather
for method in ('reduce', 'a2a', 'rs'):
    got = polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=2, batched=False, reduce_to_all=False, reduce_method=method)
    if rank == 0:
        worst = max(worst, float(np.abs(tonp(got) - want).max() / np.abs(want).max()))
# float32 maps: each rank's share of the loop is ONE pm_fft2_spectral call (groups of wavelengths per launch pair)
got = tonp(polychromatic_psf(amp.astype(np.float32), opd.astype(np.float32), wvls, wts, dx, 100.0, Q=2))
want32 = O.sum_of_2d_modes(np.asarray([O.intensity(O.focus(O.from_amp_and_phase(amp.astype(np.float32).astype(np.float64),
                                                                              opd.astype(np.float32).astype(np.float64), float(w)), 2))
                                       for w in wvls]), wts)
worst32 = float(np.abs(got - want32).max() / np.abs(want32).max())
if worst32 > 2e-5:
    worst = max(worst, worst32)     # fp32 transforms: their own tolerance; a miss fails the run
comps = []
for w in wvls:
    P = O.from_amp_and_phase(amp, opd, float(w))
    comps.append(O.intensity(O.prepare_executor(dx, P.shape, 0.55 * 10 / 4, (64, 64), float(w), 100.0)(P)))
want_m = O.sum_of_2d_modes(np.asarray(comps), wts)
got = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, focal_dx=0.55 * 10 / 4, samples=64, kind='mdft'))
worst = max(worst, float(np.abs(got - want_m).max() / np.abs(want_m).max()))
# a sequence of frames through the pipeline: every reduce form, the reduce of frame f overlapping the transforms of frame f + 1
scales = (300.0, 150.0, 450.0, 300.0)
wantf = {}
for sc in set(scales):
    o2 = O.hopkins_w040(r / 5, sc)
    wantf[sc] = O.sum_of_2d_modes(np.asarray([O.intensity(O.focus(O.from_amp_and_phase(amp, o2, float(w)), 2)) for w in wvls]), wts)
amp_t = torch.from_numpy(amp).cuda()
for kw in (dict(reduce_to_all=True), dict(reduce_method='reduce'), dict(reduce_method='a2a'), dict(reduce_method='rs')):
    pipe = PsfPipeline(wvls, wts, dx, 100.0, Q=2, batched=False, depth=2, **kw)
    pend = [pipe.submit(amp_t, torch.from_numpy(O.hopkins_w040(r / 5, sc)).cuda()) for sc in scales]
    imgs = [p.result() for p in pend]
    torch.cuda.synchronize()
    if rank == 0 or kw.get('reduce_to_all'):
        for sc, im in zip(scales, imgs):
            worst = max(worst, float(np.abs(tonp(im) - wantf[sc]).max() / np.abs(wantf[sc]).max()))
    pipe.drain()
print(f'rank {rank}: polychromatic {dist.get_world_size()}-rank sum vs oracle, worst relative error {worst:.2e}', 'OK' if worst < 1e-10 else 'FAIL', flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if worst < 1e-10 else 1)
