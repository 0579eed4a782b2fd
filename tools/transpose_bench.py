import sys; sys.path.insert(0, "/root/repo")
import torch
from visper_lm_amd import ops
for (R, C) in ((16384, 28672), (16384, 4096), (4096, 14336), (4608, 1024), (333, 777)):
    x = torch.randn(R, C, device="cuda", dtype=torch.bfloat16)
    y = ops.transpose(x)
    assert torch.equal(y, x.t().contiguous())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.transpose(x, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(R, C, f"{ms*1e3:.1f} us  {4.0*R*C/ms/1e9:.2f} TB/s")
