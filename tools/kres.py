#!/usr/bin/env python3
"""Dev tool: per-kernel register/LDS/occupancy table of one HIP source
(hipcc -Rpass-analysis=kernel-resource-usage).  usage: kres.py kernels/foo.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "darknet_amd", "csrc")


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CS, "kernels"), "-I" + os.path.join(CS, "host"),
           "-mllvm", "-pragma-unroll-threshold=200000", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
    txt = subprocess.run(cmd, stderr=subprocess.PIPE).stderr.decode()
    for b in txt.split("Function Name: ")[1:]:
        name = b.split(" ")[0]
        dn = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
        if flt and not re.search(flt, dn):
            continue
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print("%-78s VGPR %3s AGPR %3s SGPR %3s occ %s LDS %6s scratch %s" % (
            dn[:78], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"Occupancy \[waves/SIMD\]"),
            g(r"LDS Size \[bytes/block\]"), g(r"ScratchSize \[bytes/lane\]")))


if __name__ == "__main__":
    main()
