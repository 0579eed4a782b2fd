"""Dev tool (gpurun): vp_dwconv7x7_nhwc at the four ConvNeXt-XXL stage shapes of configs[3] (B = 8, 768 px)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visper_lm_amd import ops
for (H, C) in ((192, 384), (96, 768), (48, 1536), (24, 3072)):
    x = torch.randn(8, H, H, C, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(49, C, device="cuda", dtype=torch.bfloat16) * 0.1
    b = torch.randn(C, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.dwconv7x7_nhwc(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dwconv7x7_nhwc(x, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = 2 * x.numel() * 2 / 1e9
    print(f"{os.environ.get('TAG','')} H={H} C={C}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s (in + out once)  {x.numel() * 98 / us / 1e6:6.1f} TFLOP/s")
