"""config-4 GEMM pair under rocprofv3 --kernel-trace: per-kernel durations and gaps (run via tools/gpu_r2d.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _ops
torch.manual_seed(0)
Ey = torch.randn(512, 2048, dtype=torch.complex64, device='cuda')
ary = torch.randn(2048, 2048, dtype=torch.complex64, device='cuda')
Ex = torch.randn(512, 2048, dtype=torch.complex64, device='cuda')
for _ in range(40):
    T = _ops.cgemm(Ey, ary)
    out = _ops.cgemm(T, Ex, 0, 2)
torch.cuda.synchronize()
