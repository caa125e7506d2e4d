#!/usr/bin/env python3
"""Basic-block scan of a hipcc -S dump: kernels whose body is cut into many basic blocks by per-element branches.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Iklab_multimodalmodel_amd/csrc -S --cuda-device-only -o k.s <file>.hip
    python tools/scan_blocks.py k.s [min_blocks]

An unrolled loop over a lane's 16-64 elements with an `if` per element (bounds, masks, an early return inside an inlined helper)
compiles to one exec-mask or scalar branch per element; hipcc then cannot overlap the loads / LDS gathers / transcendental
instructions of neighbouring elements.  Rewriting such tests as clamped reads + selects was worth 5 % of the configs[1] step and
6 % of configs[4]."""
import re
import sys


def main():
    src = open(sys.argv[1]).read()
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for m in re.finditer(r"^(_Z[\w]+):", src, re.M):
        end = src.find("s_endpgm", m.end())
        if end < 0:
            continue
        body = src[m.end():end].splitlines()
        nblk = sum(1 for l in body if re.match(r"\s*\.LBB\d+_\d+:", l))
        ninst = sum(1 for l in body if l.strip() and not l.strip().startswith((";", ".")))
        nbr = sum(1 for l in body if "s_cbranch" in l)
        if nblk > lim:
            print(f"{nblk:5d} blocks {nbr:5d} branches {ninst:6d} instructions  {m.group(1)[:110]}")


if __name__ == "__main__":
    main()
