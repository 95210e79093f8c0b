"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

import pbrs_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbrs_[a-z_0-9]+)\s*\(", text)))


def test_gpu_library_exports_every_declared_symbol():
    names = declared_functions("pbrs_gpu.h")
    assert set(names) == set(pbrs_amd.api.GPU_SYMBOLS), (names, pbrs_amd.api.GPU_SYMBOLS)
    lib = ctypes.CDLL(pbrs_amd.lib_paths()[1])
    for n in names:
        assert getattr(lib, n) is not None, n


def test_host_library_exports_every_declared_symbol():
    names = declared_functions("pbrs_host.h")
    assert set(names) == set(pbrs_amd.api.HOST_SYMBOLS)
    lib = ctypes.CDLL(pbrs_amd.lib_paths()[0])
    for n in names:
        assert getattr(lib, n) is not None, n


def test_struct_sizes_match_the_headers():
    """ctypes mirrors vs the C layout (checked by compiling a tiny C file against the headers)."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include "pbrs_gpu.h"
#include "pbrs_scene_spec.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pbrs_node), sizeof(pbrs_instance), sizeof(pbrs_shape),
    sizeof(pbrs_mesh), sizeof(pbrs_tri_verts), sizeof(pbrs_tri_shade), sizeof(pbrs_bxdf), sizeof(pbrs_material), sizeof(pbrs_area_light),
    sizeof(pbrs_delta_light), sizeof(pbrs_scene_desc), sizeof(pbrs_camera), sizeof(pbrs_stats), sizeof(pbrs_render_params), sizeof(pbrs_scene_spec));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert sizes[:10] == [32, 128, 48, 32, 48, 64, 64, 32, 64, 32]
    from pbrs_amd import api, spec
    assert sizes[10] == ctypes.sizeof(api.SceneDesc)
    assert sizes[11] == ctypes.sizeof(api.Camera) == 64
    assert sizes[12] == ctypes.sizeof(api.Stats)
    assert sizes[13] == ctypes.sizeof(api.RenderParams)
    assert sizes[14] == ctypes.sizeof(spec.SceneSpec)


def test_product_path_fails_loudly_without_the_hip_library(monkeypatch, tmp_path):
    """No CPU fallback: a missing libpbrs_gpu.so is an error, not a silent detour."""
    from pbrs_amd import api
    monkeypatch.setattr(api, "_gpu", None)
    monkeypatch.setattr(api, "_LIBDIR", str(tmp_path))
    with pytest.raises(api.PbrsError, match="no CPU fallback"):
        api.gpu_lib()


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pbrs_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pbrs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"(^|\n)\s*(from|import)\s+oracle\b", text), os.path.join(dirpath, f)
                assert "libpbrs_oracle" not in text and "oracle/" not in text.replace("the CPU oracle in oracle/", ""), os.path.join(dirpath, f)


def test_the_shipped_hip_library_does_not_read_the_environment():
    """Kernel selection must not depend on the box a bench runs on: the developer overrides live behind -DPBRS_DEV_OVERRIDES
    and the shipped libpbrs_gpu.so does not import getenv at all."""
    import subprocess
    from pbrs_amd import api
    syms = subprocess.run(["nm", "-D", "--undefined-only", api.lib_paths()[1]], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    src = open(os.path.join(ROOT, "pbrs_amd", "csrc", "pbrs_gpu.hip")).read()
    assert src.count("getenv(") == 1 and "#ifdef PBRS_DEV_OVERRIDES" in src  # the one call sits inside dev_env()
