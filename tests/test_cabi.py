"""The C-ABI library loads on a machine without a GPU and exports every symbol include/nvblox_hip.h declares;
struct layouts agree between the header (compiled with gcc) and the ctypes mirror.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nvblox_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nvbx_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["nvbx_mapper_create", "nvbx_integrate_depth", "nvbx_integrate_color", "nvbx_update_esdf",
                 "nvbx_update_color_mesh", "nvbx_esdf_slice_to_image", "nvbx_esdf_dirty_list", "nvbx_mark_esdf_dirty"]:
        assert must in syms
    assert len(syms) >= 30


def test_library_exports_every_declared_symbol(hip_lib):
    from isaac_ros_nvblox_amd import _lib
    for s in declared_symbols():
        assert hasattr(hip_lib, s), "libnvblox_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes mirror lacks %s" % s
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_header_is_plain_c_and_struct_layouts_match():
    from isaac_ros_nvblox_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "nvblox_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(nvbx_mapper_params), sizeof(nvbx_camera), sizeof(nvbx_index3d),
         sizeof(nvbx_counters), sizeof(nvbx_esdf_voxel), sizeof(nvbx_tsdf_voxel), sizeof(nvbx_color_voxel));
  printf("%zu %zu\n", offsetof(nvbx_mapper_params, esdf_slice_height), offsetof(nvbx_mapper_params, depth_interp_nearest));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c"); exe = os.path.join(td, "t")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = list(map(int, out))
    assert sizes[0] == C.sizeof(_lib.Params)
    assert sizes[1] == C.sizeof(_lib.Camera)
    assert sizes[2] == C.sizeof(_lib.Index3D) == 12
    assert sizes[3] == C.sizeof(_lib.Counters)
    assert sizes[4] == 20 and sizes[5] == 8 and sizes[6] == 8
    assert sizes[7] == _lib.Params.esdf_slice_height.offset and sizes[8] == _lib.Params.depth_interp_nearest.offset


def test_oracle_params_layout_equals_product_params(oracle_mod):
    from isaac_ros_nvblox_amd import _lib
    assert [n for n, _ in oracle_mod.OrcParams._fields_] == [n for n, _ in _lib.Params._fields_]
    assert C.sizeof(oracle_mod.OrcParams) == C.sizeof(_lib.Params)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under isaac_ros_nvblox_amd/ or include/ may reference it."""
    bad = []
    for base in ("isaac_ros_nvblox_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b", t, flags=re.M) or "libnvblox_oracle" in t or "orc_" in t:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_mapper_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isaac_ros_nvblox_amd import mapper as M
    with pytest.raises(M.NvbxError):
        M.Mapper(M.default_params())


def test_default_params_equal_python_mirror_and_device_view_layout(hip_lib):
    """nvbx_default_params (a pure host function: callable without a GPU) == mapper.default_params(); the nvbx_device_view
    struct of the header == the ctypes mirror; the device-side header compiles as plain C when no HIP compiler is used."""
    from isaac_ros_nvblox_amd import _lib, mapper as M
    p = _lib.Params()
    hip_lib.nvbx_default_params.restype = None
    hip_lib.nvbx_default_params.argtypes = [C.POINTER(_lib.Params)]
    hip_lib.nvbx_default_params(C.byref(p))
    q = M.default_params()
    for name, _ in _lib.Params._fields_:
        a, b = getattr(p, name), getattr(q, name)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), name
    src = r'''
#include <stdio.h>
#include "nvblox_hip_device.h"
int main(void) { printf("%zu\n", sizeof(nvbx_device_view)); return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c"); exe = os.path.join(td, "t")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        assert int(subprocess.check_output([exe]).decode()) == C.sizeof(_lib.DeviceView)


def test_xcd_affine_record_numbering_is_a_bijection_with_runs_on_one_xcd(tmp_path):
    """csrc/nvbx_numbering.h xcd_chunked (round 6): the function the TSDF-update / colour workers call to pick their record, compiled from the shipped
    header with gcc.  For every grid size: every record taken exactly once; inside the permuted part, the C consecutive records of a run are taken by
    workers with one `w & 7` (= one XCD, workgroups being dispatched round-robin) and the eight runs of a period by eight different ones; the tail that
    does not fill a period keeps its numbers; run length 0 = identity."""
    import ctypes
    src = tmp_path / "num.c"
    src.write_text('#include "nvbx_numbering.h"\nint32_t map_c(int32_t w, int32_t n, int32_t c) { return xcd_chunked_c(w, n, c); }\n'
                   'int32_t map_default(int32_t w, int32_t n) { return xcd_chunked(w, n); }\nint32_t run_default(void) { return NVBX_XCD_CHUNK; }\n')
    so = tmp_path / "num.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-Werror", "-I", os.path.join(ROOT, "isaac_ros_nvblox_amd", "csrc"), str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    run = lib.run_default()
    assert run in (8, 16, 32)
    for n in [0, 1, 7, 8, 63, 64, 65, 127, 128, 129, 255, 256, 300, 449, 450, 512, 1000, 1024, 2304]:
        for c in [0, 4, 8, 16, 32]:
            taken = [lib.map_c(w, n, c) for w in range(n)]
            assert sorted(taken) == list(range(n)), (n, c)
            if c == 0:
                assert taken == list(range(n))
                continue
            period = 8 * c; full = (n // period) * period
            assert taken[full:] == list(range(full, n)), (n, c)
            owner = {rec: w & 7 for w, rec in enumerate(taken)}
            for r0 in range(0, full, c):
                assert len({owner[r] for r in range(r0, r0 + c)}) == 1, (n, c, r0)
            for p0 in range(0, full, period):
                assert {owner[p0 + k * c] for k in range(8)} == set(range(8)), (n, c, p0)
        assert [lib.map_default(w, n) for w in range(n)] == [lib.map_c(w, n, run) for w in range(n)]
