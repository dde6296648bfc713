"""Achievable HBM WRITE bandwidth on this chip (the Gram kernels are pure stores): hipMemsetAsync and torch fill of 512 MB / 4 GB."""
import time, torch
for nbytes in (512 << 20, 4 << 30):
    x = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    for name, fn in (("fill_(1.5)", lambda: x.fill_(1.5)), ("zero_()", lambda: x.zero_()), ("copy_ (read+write)", None)):
        if fn is None:
            y = torch.empty_like(x)
            fn = lambda: y.copy_(x)
            factor = 2
        else:
            factor = 1
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-20s %5d MB: %.1f us -> %.2f TB/s" % (name, nbytes >> 20, ms * 1e3, factor * nbytes / (ms * 1e-3) / 1e12))
