"""snerf_amd -- MI355X-native volumetric-render hot path of S-NeRF (sample -> encode -> tiny MLP -> composite).

Everything computes in hand-written HIP kernels behind the C-ABI of include/snerf_hip.h (libsnerf_hip.so);
this package is the host-side mirror of the reference's Python operator API:

  snerf_amd.classic   render_rays / run_network / raw2outputs / sample_pdf / NeRF      (s-nerf/model/render.py, run_nerf_helpers.py)
  snerf_amd.mipnerf   MipNerfModel / make_mipnerf / render_image / Rays                (s-nerf/model/models.py, mip.py, math_ops.py)
  snerf_amd.trainer   fused train step + ray-sharded data parallelism over RCCL        (s-nerf/train.py hot loop)
"""
__version__ = "0.1.0"
