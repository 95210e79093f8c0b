"""pbrs_amd — MI355X-native wavefront path tracer behind the pbrs integrator seam.

Only the path-integrator hot path of plumer/pbrs: the HIP kernels + C ABI
(csrc/, include/pbrs_gpu.h), the host-side scene flattener (csrc/host, include/pbrs_host.h),
their ctypes bindings (api.py) and the synthetic scene generators (scenes.py).
"""
from .api import Context, HostScene, LoadedScene, PbrsError, gpu_lib, host_lib, lib_paths, load_pbrt, write_image  # noqa: F401
from . import scenes, spec  # noqa: F401
