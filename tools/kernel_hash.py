#!/usr/bin/env python3
"""sha256 of the source text of ONE kernel of csrc/gemm.hip (from its template / __global__ line to the next top-level comment
block), so that a measurement in profiles/ can name exactly the code it was taken on: bench.py's roofline.traffic comes from
profiles/roofline_traffic.json and is reported only while this hash matches (tests/test_bench_contract.py fails on a mismatch).

    python tools/kernel_hash.py                      # prints the hash of gemm_nt8p_kernel
    python tools/kernel_hash.py --update FETCH WRITE SOURCE   # rewrite the JSON after a new PMC measurement (bytes per launch)
"""
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JSON = os.path.join(REPO, "profiles", "roofline_traffic.json")


def kernel_source(name="gemm_nt8p_kernel", path=None):
    src = open(path or os.path.join(REPO, "snerf_amd", "csrc", "gemm.hip")).read()
    m = re.search(r"(template <[^>]*>\s*)?__global__[^\n]*\bvoid " + re.escape(name) + r"\(", src)
    if m is None:
        raise ValueError(f"kernel {name} not found")
    depth, i = 0, src.index("{", m.end())
    for j in range(i, len(src)):
        depth += src[j] == "{"
        depth -= src[j] == "}"
        if depth == 0:
            return src[m.start():j + 1]
    raise ValueError("unbalanced braces")


def kernel_hash(name="gemm_nt8p_kernel"):
    return hashlib.sha256(kernel_source(name).encode()).hexdigest()


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--update":
        d = {"kernel": "gemm_nt8p_kernel", "launch": "M=524288 N=K=1024 bf16, bias + ReLU epilogue (tools/gemm_single.py 8 nt)",
             "fetch_bytes": float(sys.argv[2]), "write_bytes": float(sys.argv[3]), "algorithmic_bytes": 2.15e9,
             "counters": "FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE, separate rocprofv3 --pmc passes (tools/pmc_gemm_traffic.sh)",
             "source": sys.argv[4], "kernel_sha256": kernel_hash()}
        json.dump(d, open(JSON, "w"), indent=1)
        print("wrote", JSON)
    else:
        print(kernel_hash(sys.argv[1] if len(sys.argv) > 1 else "gemm_nt8p_kernel"))
