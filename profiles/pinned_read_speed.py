#!/usr/bin/env python3
"""How fast does the HOST read memory the GPU copies into?  torch's pin_memory (hipHostMalloc, coherent) against a plain tensor registered with
hipHostRegister, against pageable memory: the default sampled path sums a 1 MB map per reference on the host (upstream's normaliser)."""
import time

import numpy as np
import torch


def t(fn, n=200):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


dev = torch.device("cuda:0")
src = torch.rand((512, 512), device=dev)
plain = torch.empty((512, 512))
pinned = torch.empty((512, 512)).pin_memory()
reg = torch.empty((512, 512))
rc = torch.cuda.cudart().cudaHostRegister(reg.data_ptr(), reg.numel() * 4, 0)
print("cudaHostRegister rc", rc, "is_pinned", reg.is_pinned())
for name, h in (("pageable", plain), ("pin_memory", pinned), ("host-registered", reg)):
    h.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ok = bool(torch.equal(h, src.cpu()))
    rd = t(lambda: float(h.sum()))
    rd_np = t(lambda: float(h.numpy().sum()))

    def d2h():
        h.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
    print(f"{name:16s} copy ok {ok}  torch.sum {rd:8.1f} us  numpy sum {rd_np:8.1f} us  D2H 1 MB + sync {t(d2h):8.1f} us")
torch.cuda.cudart().cudaHostUnregister(reg.data_ptr())
