#!/usr/bin/env python3
"""Cost of handing work to a second HIP stream and back (event record -> wait -> kernel -> record -> wait), the
pattern of the overlapped ghost exchange (csrc/shk_comm.hip halo_begin / halo_end), against the same kernels on one
stream.  GPU box; prints microseconds per round trip."""
import time

import torch

x = torch.zeros(1024, device="cuda")
y = torch.zeros(1024, device="cuda")
big = torch.zeros(64 << 20, device="cuda")
A = torch.cuda.Stream()
B = torch.cuda.Stream()
ea = torch.cuda.Event()
eb = torch.cuda.Event()


def run(n, handoff, interior):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(A):
            x.add_(1.0)
            if handoff:
                ea.record(A)
                if interior:
                    big.add_(1.0)          # ~70 us of "interior" work on the main stream
                B.wait_event(ea)
                with torch.cuda.stream(B):
                    y.add_(1.0)            # the "exchange"
                    eb.record(B)
                A.wait_event(eb)
            else:
                if interior:
                    big.add_(1.0)
                y.add_(1.0)
            x.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for interior in (False, True):
    for handoff in (False, True):
        run(200, handoff, interior)
        print(f"interior work {interior!s:5}  second stream {handoff!s:5}  {run(2000, handoff, interior):8.2f} us per round")
