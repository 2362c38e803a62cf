"""DESIGN.md 2.8's invariants of the fused launches, executed: the -DNVBX_CHECK_INVARIANTS variant of the library (tools/build_variant.sh inv) counts on
the device -- I1: a TSDF-reading rider of launch 1 (sphere tracing, colour candidates, ESDF marking) beside a running TSDF writer; I3: a colour worker of
launch 2 handed a candidate record whose slot does not name the block; I4: the marking pass taking a dirty-list entry of a slot without a layer -- and on
the host -- I8: a colour-reading launch set up on a library frame nobody holds.  Every mapper of the randomised call patterns (three seeds, repeated),
the steady-state pipeline and the API-sequence tests answers for its launches when it is closed (NVBX_CHECK_ON_CLOSE=1); a self test shows that the
counters count.  The product library compiles the checks to nothing (tests/test_cabi.py: its exports are unchanged)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "isaac_ros_nvblox_amd", "variants", "libnvblox_hip_inv.so")


def _variant():
    srcs = [os.path.join(ROOT, "isaac_ros_nvblox_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "isaac_ros_nvblox_amd", "csrc")) if f.endswith((".hip", ".h", ".inc"))]
    if not os.path.exists(VARIANT) or os.path.getmtime(VARIANT) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "inv", "-DNVBX_CHECK_INVARIANTS"], stdout=subprocess.DEVNULL)
    return VARIANT


def test_the_counters_count_and_the_fused_launches_keep_their_invariants():
    code = (
        "import sys, gc; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import functools, oracle, helpers, test_gpu_pipeline as P, test_gpu_sequences as Q\n"
        "from isaac_ros_nvblox_amd import _lib, mapper as M, synthetic as S\n"
        "lib = _lib.load()\n"
        "g = M.Mapper(M.default_params(), block_capacity=1 << 12)\n"
        "assert g.invariant_violations() == (0, 0, 0, 0, 0), g.invariant_violations()\n"
        "v = g.invariant_violations(selftest=True)\n"
        "assert v is not None and v[0] == 1 and v[3] == 0, v            # (a reader met a pretended writer: counted once; the writer count is back at 0)\n"
        "import os; os.environ['NVBX_CHECK_ON_CLOSE'] = '0'; g.close(); os.environ['NVBX_CHECK_ON_CLOSE'] = '1'      # (this one carries the self test's count)\n"
        "helpers.frames = functools.lru_cache(maxsize=None)(helpers.frames)\n"
        "for rep in range(4):\n"
        "    for seed in (0, 1, 2):\n"
        "        P.test_fused_colour_tsdf_launch_under_irregular_calls(oracle, lib, seed)\n"
        "for staged in (True, False):\n"
        "    P.test_steady_state_pipeline_equals_classic_and_oracle(oracle, lib, helpers.SMALL_CAM, staged)\n"
        "gc.collect()\n"
        "r = M.INVARIANT_REPORT\n"
        "assert r['mappers_checked'] >= 20 and not r['violations'], r\n"
        "print('INVARIANTS_OK', r['mappers_checked'])\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, NVBX_LIB=_variant(), NVBX_CHECK_ON_CLOSE="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0 and "INVARIANTS_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_the_product_library_carries_no_checks():
    from isaac_ros_nvblox_amd import _lib
    lib = _lib.load()
    assert not hasattr(lib, "nvbx_debug_invariants") or os.environ.get("NVBX_LIB")
