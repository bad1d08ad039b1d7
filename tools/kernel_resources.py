#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of the kernels of one csrc/*.hip file whose mangled name matches a pattern
(hipcc -Rpass-analysis=kernel-resource-usage, compiled for gfx950 into /tmp):  python tools/kernel_resources.py zip.hip g3_write"""
import os, re, subprocess, sys

def main():
    src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "snerf_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result", "-Wno-unused-value",
           "-c", os.path.join(csrc, src), "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2); rows[cur] = {}
        elif cur:
            rows[cur][m.group(1)] = m.group(2)
    for k, v in rows.items():
        if pat in k:
            print(f"{k[:90]:90s} vgpr {v.get('VGPRs', '?'):>4} agpr {v.get('AGPRs', '?'):>3} spilled {v.get('VGPRs Spill', '?'):>4} scratch {v.get('ScratchSize [bytes/lane]', '?'):>5} "
                  f"occ {v.get('Occupancy [waves/SIMD]', '?')} lds {v.get('LDS Size [bytes/block]', '?')}")

if __name__ == "__main__":
    main()
