#!/usr/bin/env python3
"""Ablation builds of the binned table gradient (csrc/zip.hip), round 4: writes a patched copy of zip.hip whose record writers and
accumulate kernel drop one piece each under -DZB_ABL=n, builds libsnerf_hip_abl<n>.so next to the shipped library, and the GPU side runs
tools/bench_zip.py --train-only under rocprofv3 with SNERF_HIP_LIB pointing at each (profiles/r4_s_pathC_table_gradient_ablation.txt).

  1  direct writer (zip_emit_level): no global store            2  ... and no LDS slot atomic
  3  direct writer: stores to a thread-linear (coalesced) slot   4  staged writer: no global store
  5  staged writer: count walk only (no placement, no stream-out)
  6  accumulate: no LDS atomics                                  7  accumulate: no write-back

    python tools/probes/zip_table_gradient_ablation.py        # builds snerf_amd/lib/abl/libsnerf_hip_abl{1..7}.so (needs the shipped .o files)
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "snerf_amd", "csrc")
EDITS = [
    ("""      const int slot = atomicAdd(lds_cnt + bin, 1);
      if (WRITE) {
        const long r = lds_base[bin] + slot;
""", """#if ZB_ABL == 2
      const int slot = WRITE ? (int)(row & 7u) : atomicAdd(lds_cnt + bin, 1);
#else
      const int slot = atomicAdd(lds_cnt + bin, 1);
#endif
      if (WRITE) {
#if ZB_ABL == 1 || ZB_ABL == 2
        const long r = (lds_base[bin] + slot == -12345) ? 0 : b.capacity;
#elif ZB_ABL == 3
        const long r = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 * 8 + idx * 256 + threadIdx.x + (lds_base[bin] + slot == -12345);
#else
        const long r = lds_base[bin] + slot;
#endif
"""),
    ("""      const long r = base[bin] + (sidx - off[bin]);
""", """#if ZB_ABL == 4
      const long r = (base[bin] + (sidx - off[bin]) == -12345) ? 0 : b.capacity;
#else
      const long r = base[bin] + (sidx - off[bin]);
#endif
"""),
    ("""    walk(z, true);
    __syncthreads();
    for (int sidx = tid; sidx < total; sidx += 256) {""", """#if ZB_ABL != 5
    walk(z, true);
#endif
    __syncthreads();
#if ZB_ABL == 5
    if (total == -12345)
#endif
    for (int sidx = tid; sidx < total; sidx += 256) {"""),
    ("""        atomicAdd((unsigned long long*)(zb_acc + row[u] * C + c), (unsigned long long)__float2ll_rn(v));""", """#if ZB_ABL == 6
        if (__float2ll_rn(v) == 0x123456789LL) zb_acc[row[u] * C + c] = 1;
#else
        atomicAdd((unsigned long long*)(zb_acc + row[u] * C + c), (unsigned long long)__float2ll_rn(v));
#endif"""),
    ("""  if (K == 1) {                                             // the only workgroup that owns these rows""", """#if ZB_ABL == 7
  if (n != -12345) return;
#endif
  if (K == 1) {                                             // the only workgroup that owns these rows"""),
]


def main():
    src = open(os.path.join(CSRC, "zip.hip")).read()
    for a, b in EDITS:
        assert src.count(a) == 1, a
        src = src.replace(a, b)
    open(os.path.join(CSRC, "_zip_abl.hip"), "w").write(src)
    os.makedirs(os.path.join(REPO, "snerf_amd", "lib", "abl"), exist_ok=True)
    objs = [o for o in "gemm.o fmlp.o encode.o sampler.o composite.o elementwise.o grid.o callers.o ert.o foreground.o".split()]
    procs = []
    for n in range(1, 8):
        cmd = (f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DZB_ABL={n} "
               f"-c _zip_abl.hip -o /tmp/zip_abl{n}.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/abl/libsnerf_hip_abl{n}.so "
               + " ".join(objs) + f" /tmp/zip_abl{n}.o")
        procs.append(subprocess.Popen(cmd, shell=True, cwd=CSRC))
    rc = [p.wait() for p in procs]
    os.remove(os.path.join(CSRC, "_zip_abl.hip"))
    sys.exit(max(rc))


if __name__ == "__main__":
    main()
