#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one .hip file (no GPU needed):

    python tools/kernel_resources.py irs_mpc_amd/csrc/smooth.hip 'smooth_kernel<PlanarHand'

Compiles the file for gfx950 with the Makefile's flags plus -Rpass-analysis=kernel-resource-usage and
prints, per kernel whose demangled name matches the regular expression, VGPRs / AGPRs / scratch bytes per
lane / waves per SIMD / spilled VGPRs.  The contact kernels sit at the 256-register boundary between one
and two waves per SIMD: run this after touching csrc/contact_models.hpp."""
import re
import subprocess
import sys

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast"
SMOOTH_EXTRA = "-ffinite-math-only -fno-signed-zeros"       # csrc/Makefile: smooth.o only


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "."
    extra = SMOOTH_EXTRA if src.endswith("smooth.hip") else (SMOOTH_EXTRA + " -fno-slp-vectorize" if src.endswith("smooth_ug.hip") else "")
    cmd = "/opt/rocm/bin/hipcc %s %s -Rpass-analysis=kernel-resource-usage -c %s -o /tmp/kernel_resources.o" % (
        FLAGS, extra, src)
    out = subprocess.run(cmd, shell=True, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1)] = int(m.group(2))
    names = list(rows)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    keys = ("VGPRs", "AGPRs", "SGPRs", "ScratchSize", "Occupancy", "VGPRs Spill", "SGPRs Spill", "LDS Size")
    for name, k in zip(dem, names):
        if re.search(filt, name):
            print(name.replace("(anonymous namespace)::", "")[:100], {x: rows[k].get(x) for x in keys})


if __name__ == "__main__":
    main()
