#!/usr/bin/env python3
"""Time the LOCAL stages of one rank of a P-rank slab decomposition on a single GPU (the exchange is
replaced by a device copy of the rank's own send buffer, same byte count as the receive buffer), to
size the compute side of the 512^3 / 8-GPU configuration without an 8-GPU node.
usage: slab_local_bench.py [P] [n] [steps] [nsub] [carry 0|1] [fused-run 0|1] [experiment mask] [global grid gx,gy,gz]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import grid_for, splitmix64_uniform  # noqa: E402
from marlin_amd.api import ch_params  # noqa: E402
from marlin_amd.slab import SlabCahnHilliard  # noqa: E402


class _Copy:
    def __init__(self, sc, rc):
        assert sum(sc) == sum(rc)

    def run(self, send, recv, async_op=False):
        recv.copy_(send)
        return None


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    nsub = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    carry = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
    fused = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
    exp = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    shape = [int(x) for x in sys.argv[8].split(",")] if len(sys.argv) > 8 else grid_for(P, n)   # argv[8]: the GLOBAL grid gx,gy,gz
    dx = 8.0 * np.pi / 200.0
    s = SlabCahnHilliard(3, shape, [x * dx for x in shape], ch_params(), P, 0, exchange_factory=lambda a, b: _Copy(a, b), nsub=nsub,
                         carry=carry, exp=exp)
    s.set_initial(lambda count, offset: splitmix64_uniform(count, offset=offset))
    for _ in range(5):
        s.substep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fused:
        s.run(steps, advance=True, advance_after=True)
    else:
        for _ in range(steps):
            s.substep()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    s.ctx.set_profiling(True)
    if fused:
        s.run(10, advance=True, advance_after=True)
    else:
        for _ in range(10):
            s.substep()
    torch.cuda.synchronize()
    prof = [k for k in s.ctx.get_profile() if k["launches"]]
    for k in prof:
        k["avg_ms"] = k["ms"] / k["launches"]
        k["GBps"] = k["bytes_per_launch"] / k["avg_ms"] / 1e6
    print(json.dumps({"P": P, "exp": exp, "nsub": nsub, "carry": carry, "fused_run": fused, "global_grid": shape, "local_real": s.st.real_shape, "ms_per_substep_local_incl_copies": ms,
                      "kernels": [{k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in k.items()} for k in prof]}))


if __name__ == "__main__":
    main()
