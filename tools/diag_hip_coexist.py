"""Diagnostic: which HIP runtime ends up in the process when torch and liborbx are both loaded."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

order = sys.argv[1]


def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        for key in ("libamdhip64", "libhsa-runtime64", "liborbx"):
            if key in line:
                seen.add(line.split()[-1])
    return sorted(seen)


def use_orbx():
    import numpy as np
    import my_orb_slam2_amd as m
    k, d = m.ORBextractor(1000, 1.2, 8, 20, 7)(m.synth_frame(0))
    return len(k)


def use_torch():
    import torch
    ok = torch.cuda.is_available()
    n = torch.cuda.device_count()
    s = float(torch.ones(4, device="cuda").sum()) if ok else -1
    return ok, n, s


if order == "torch_first":
    print("torch:", use_torch()); print(maps()); print("orbx:", use_orbx())
elif order == "orbx_first":
    print("orbx:", use_orbx()); print(maps()); print("torch:", use_torch())
elif order == "import_torch_then_orbx_then_torch_cuda":
    import torch
    print("orbx:", use_orbx()); print(maps()); print("torch:", use_torch())
print(maps())
