import numpy as np, torch, time
from prysm_amd.polychromatic import polychromatic_psf
from prysm_amd import _lib
lib=_lib.load()
rng=np.random.default_rng(1)
for n in (500, 1000, 1536):
    amp=torch.from_numpy((rng.random((n,n))>0.3).astype(np.float32)).cuda(); opd=torch.from_numpy((200*rng.standard_normal((n,n))).astype(np.float32)).cuda()
    wv=np.linspace(0.5,0.7,16); wt=np.ones(16)
    def run(**kw): return polychromatic_psf(amp,opd,wv,wt,0.04,100.0,Q=1,**kw)
    ref=run(batched=False, spectral=False)
    for name,kw,eng in (('loop (accumulate epilogue)',dict(batched=False,spectral=False),1),('stacks, register engine',dict(batched=True),1),('stacks, general kernel',dict(batched=True),0),('default',dict(),1)):
        _lib.check(lib.pm_set_tuning(b'mix_engine',eng))
        y=run(**kw); err=((y-ref).abs().max()/ref.abs().max()).item()
        for _ in range(3): run(**kw)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): run(**kw)
        torch.cuda.synchronize(); t=(time.perf_counter()-t0)/20*1e6
        print('POLY %4d^2 x 16 wavelengths  %-28s %8.1f us per PSF (%.1f per wavelength)  rel diff to the loop %.1e'%(n,name,t,t/16,err),flush=True)
    _lib.check(lib.pm_set_tuning(b'mix_engine',1))
