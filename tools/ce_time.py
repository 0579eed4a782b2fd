"""Dev tool (gpurun): vp_ce_fwd_bwd at the configs[1] shape (11232 labelled rows x V 128256), register-resident rows (default) vs VP_CE_REG=0; bitwise compare."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1:
    from visper_lm_amd import ops, _lib
    rows, V = 11232, int(sys.argv[2])
    g = torch.Generator(device="cuda").manual_seed(1)
    base = (torch.randn(rows, V, device="cuda", generator=g) * 2.0).to(torch.bfloat16)
    labels = torch.randint(0, V, (rows,), device="cuda", generator=g)
    labels[::7] = -100
    loss = torch.empty(rows, device="cuda")
    call = lambda lg: _lib.call("vp_ce_fwd_bwd", rows, V, ops._p(lg), V, ops._p(labels), ops._p(loss), 0.5, 1, ops._stream())
    lg = base.clone(); call(lg); torch.cuda.synchronize()
    torch.save((lg.cpu(), loss.cpu()), f"/tmp/ce_{sys.argv[1]}.pt")
    bufs = [base.clone() for _ in range(3)]
    for b_ in bufs: call(b_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for r in range(3):
        for i, b_ in enumerate(bufs): b_.copy_(base)
        torch.cuda.synchronize(); e0.record()
        for b_ in bufs: call(b_)
        e1.record(); torch.cuda.synchronize(); res.append(round(e0.elapsed_time(e1) / 3, 3))
    print(sys.argv[1], "V", V, res, "ms")
else:
    for V in (128256, 32064):
        for tag, env in (("reg", "1"), ("old", "0")):
            subprocess.run([sys.executable, __file__, tag, str(V)], env=dict(os.environ, VP_CE_REG=env))
        a, b = torch.load("/tmp/ce_reg.pt"), torch.load("/tmp/ce_old.pt")
        print("V", V, "dlogits identical", bool(torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))), "loss identical", bool(torch.equal(a[1], b[1])))
