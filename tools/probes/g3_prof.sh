#!/bin/bash
# Cycles per phase of g3_write_staged_kernel (probe build of the library with -DG3_PROF; the shipped build has no such hooks):
#   bash tools/probes/g3_prof.sh [extra -D flags, e.g. -DG3_ABL_NOATOMIC]          (on the GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
D=/tmp/g3prof; rm -rf $D; mkdir -p $D
cp snerf_amd/csrc/*.hip snerf_amd/csrc/*.h $D/; cp snerf_amd/csrc/*.o $D/ 2>/dev/null; rm -f $D/zip.o
( cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DG3_PROF "$@" -c zip.hip -o zip.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libsnerf_hip_prof.so gemm.o fmlp.o encode.o sampler.o composite.o elementwise.o grid.o zip.o callers.o ert.o foreground.o ) 2>&1 | grep -E "error|Error"
echo "flags: $*"
SNERF_HIP_LIB=$D/libsnerf_hip_prof.so python - <<'PY'
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from snerf_amd import ops, _lib
from snerf_amd.gridencoder import GridEncoder
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.snerf_g3_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L, C = 10, 4
x = (bench.grid_points(dev) + 1) / 2
B = x.shape[0]
enc = GridEncoder(input_dim=3, num_levels=L, level_dim=C, base_resolution=16, desired_resolution=8192, log2_hashmap_size=21, device=dev)
S, H = float(np.log2(enc.per_level_scale)), 16
oh = enc.offsets.cpu().numpy()
g = torch.Generator(device=dev).manual_seed(4)
w = (torch.randn(B, L * C, generator=g, device=dev) * 1e-3).half()
buf = (ctypes.c_ulonglong * 16)()
for ws in (None,):
    f = lambda: ops.grid_encode_bwd_binned(w, x, enc.offsets, C, L, S, H, out_dtype=torch.float16, offsets_host=oh, ws_bytes=ws)
    f(); torch.cuda.synchronize()
    lib.snerf_g3_prof_read(ctypes.addressof(buf), 1)
    f(); torch.cuda.synchronize()
    lib.snerf_g3_prof_read(ctypes.addressof(buf), 1)
    v = np.array(list(buf)[:7], dtype=np.float64)
    names = ["decode + gradient loads", "histogram load + scan", "values + classify", "place: barrier wait", "stream out (issue)", "place: store drain + pad prefill", "place: thread 0's own walk"]
    print(f"ws {ws}: s_memtime ticks per phase (thread 0 of every workgroup, summed), share of the total; whole backward {bench._timeit(f, 3, warm=1) * 1e3:.2f} ms")
    for n_, t_ in zip(names, v):
        print(f"  {n_:34s} {t_ / 1e6:10.1f} M  {100 * t_ / v.sum():5.1f} %")
PY
