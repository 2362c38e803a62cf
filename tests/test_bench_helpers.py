"""bench.py's pure helpers (no GPU): kernel-name parsing, the SURVEY 8d byte formulas -- in particular that the fused launches of the
two-launch pipeline account for exactly the work of the classic launches they replace --, block statistics and the exploring bookkeeping."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

COUNTS = {"tsdf_blocks_in_view": 310.0, "color_blocks_updated": 240.0, "blocks_allocated": 1650.0, "esdf_columns_marked": 120.0, "esdf_blocks_swept": 108.0,
          "mesh_blocks_updated": 300.0, "mesh_vertices": 21000.0, "mesh_triangles": 14000.0}
R, C = 480, 640


def test_kernel_name_parsing():
    assert bench.short(" k_mark_view<Img, Sensor, NB> ") == "k_mark_view"
    assert bench.short("void k_integrate_tsdf_color<nvbx::DepthF32, nvbx::PixRgb8, 1, true>(nvbx::DMap, ...)") == "k_integrate_tsdf_color"
    assert bench.short("k_esdf_edt(nvbx::DMap, nvbx::EsdfArgs)") == "k_esdf_edt"


def test_fused_launches_account_for_the_work_they_replace():
    ab = lambda k, **kw: bench.algorithmic_bytes(k, COUNTS, R, C, **kw)
    classic = ab("k_mark_view") + ab("k_sphere_trace") + ab("k_integrate_color") + ab("k_integrate_tsdf") + ab("k_esdf_edt")
    three = ab("k_mark_view", trace_in_mark_view=True) + ab("k_integrate_color") + ab("k_integrate_tsdf") + ab("k_esdf_edt")
    two = ab("k_mark_view", fused=True) + ab("k_integrate_tsdf_color", fused=True)
    assert three == classic
    # two launches: the same work + the candidate discovery (flags / Index3D of every allocated slot, one 16-byte record per candidate written
    # by the riders and read by the colour workers)
    extra = COUNTS["blocks_allocated"] * 16 + 2 * COUNTS["color_blocks_updated"] * 16
    assert two == classic + extra
    assert ab("k_integrate_tsdf_color") > ab("k_integrate_tsdf") > 0 and ab("k_integrate_color") == ab("k_esdf_mark") + (ab("k_integrate_color") - ab("k_esdf_mark")) > 0


def test_lidar_launches_split_the_blocks():
    c = dict(COUNTS, tsdf_blocks_in_view=112000.0, lidar_blocks_beam_centric=68000.0)
    dense_only = bench.algorithmic_bytes("k_integrate_tsdf", dict(c, lidar_blocks_beam_centric=0.0), 64, 1024)
    dense = bench.algorithmic_bytes("k_integrate_tsdf", c, 64, 1024)
    sparse = bench.algorithmic_bytes("k_lidar_sparse", c, 64, 1024)
    assert dense_only - dense == 68000 * 4096 * 2          # the blocks the beam-centric launch takes are not credited to the dense one
    assert 0 < sparse < 68000 * 4096                       # ... and cost it a fraction of a block each


def test_block_stats_and_exploring_bookkeeping():
    st = bench.block_stats([0.002, 0.004, 0.003], 100)
    assert st["blocks"] == 3 and st["median"] == 0.03 and st["min"] == 0.02 and st["max"] == 0.04 and abs(st["timed_ms_total"] - 9.0) < 1e-9
    # driver-like: 20 steps per block, 200 unique poses -> 10 blocks per loop; the third loop was cut short and does not count
    tags = [20 * (i % 10) for i in range(27)]
    dts = [1.0 + 0.5 * (t == 0) for t in tags]
    per_loop, starts, whole, kept = bench.complete_loops(tags, dts, 200, 20)
    assert per_loop == 10 and starts == [0, 10, 20] and whole == [0, 10] and len(kept) == 20 and abs(sum(kept) - 21.0) < 1e-9
    # one block per loop (K = nu), and K > nu (every block starts from an empty map)
    assert bench.complete_loops([0, 0, 0], [1.0, 2.0, 3.0], 200, 200) == (1, [0, 1, 2], [0, 1, 2], [1.0, 2.0, 3.0])
    assert bench.complete_loops([0, 0], [1.0, 2.0], 200, 500)[3] == [1.0, 2.0]
    # no loop completed: every block counts
    assert bench.complete_loops([0, 20], [1.0, 2.0], 200, 20)[3] == [1.0, 2.0]


def test_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` without WORLD_SIZE re-executes itself through torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1
    (bench.py --gpus 8 as the driver types it for N = 1 died on the WORLD_SIZE assertion)."""
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    free = bench.launch_command(2, [])              # (no port given: a free one is picked)
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536
    # and the re-exec really happens: a 2-rank launch of `bench.py --help`-like failure is reported through the exit code, not an assertion in rank 0
    import subprocess
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env["NVBX_BENCH_LAUNCH_DRY"] = "1"
    p = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "3"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0 and "torch.distributed.run" in p.stdout and "--nproc-per-node 2" in p.stdout, (p.stdout, p.stderr[-500:])


def test_round4_kernels_have_algorithmic_bytes():
    """The launches round 4 added are priced by what they move: the dynamic-mapping front end's three launches (image-sized passes).  The staged colour
    deferral's copy of RAW-pointer images (image read + written, per camera of a batch) is OVERHEAD, not algorithm -- SURVEY 8d has no such term
    (VERDICT r04): 0 algorithmic bytes, reported as overhead_bytes; images in library frames are not copied at all."""
    c = dict(tsdf_blocks_in_view=300, color_blocks_updated=200, blocks_allocated=1600, esdf_columns_marked=60, esdf_blocks_swept=200)
    rows, cols = 480, 640
    assert bench.algorithmic_bytes("k_stage_color", c, rows, cols) == 0
    assert bench.overhead_bytes("k_stage_color", rows, cols) == rows * cols * 3 * 2
    assert bench.overhead_bytes("k_stage_color", rows, cols, n_cam=8) == 8 * rows * cols * 3 * 2 and bench.overhead_bytes("k_mark_view", rows, cols) == 0
    for k in ("k_dyn_detect_union", "k_cc_count", "k_dyn_filter_split"):
        b = bench.algorithmic_bytes(k, c, rows, cols)
        assert rows * cols * 4 < b < rows * cols * 64, (k, b)
    two = sum(bench.algorithmic_bytes(k, c, rows, cols, trace_in_mark_view=True, fused=True) for k in ("k_mark_view", "k_integrate_tsdf_color"))
    assert two > 5e6 and bench.overhead_bytes("k_stage_color", rows, cols) < 0.4 * two


def test_round5_lidar_view_launches_have_algorithmic_bytes():
    """The three launches of the LiDAR view calculation over the dense grid: the marking launch reads the sub-sampled range image and
    stores two bytes per block in view, the scan reads the coarse cell map and 16-byte records go out, the resolving launch touches one record and one
    hash entry per block -- and `k_mark_view_grid` must not fall into `k_mark_view`'s formula (a prefix of its name)."""
    c = dict(COUNTS, tsdf_blocks_in_view=112000.0, view_grid_cells=127 * 127 * 50)
    ab = lambda k: bench.algorithmic_bytes(k, c, 64, 1024, sub_ray=2)
    assert ab("k_mark_view_grid") == 32 * 512 * 4 + 112000 * 2
    assert ab("k_scan_view_grid") == 127 * 127 * 50 + 112000 * 18
    assert ab("k_resolve_view") == 112000 * 40
    assert ab("k_mark_view") != ab("k_mark_view_grid")
    assert bench.short("void k_mark_view_grid<nvbx::DepthF32>(nvbx::DMap, ...)") == "k_mark_view_grid"
    # the staged copy is overhead, never algorithm
    assert bench.algorithmic_bytes("k_stage_color", COUNTS, R, C) == 0 and bench.overhead_bytes("k_stage_color", R, C) == R * C * 3 * 2


def test_pair_launch_byte_formulas():
    """nvbx_integrate_depth_pair (round 6): a pair launch moves the first mapper's bytes (two-launch frame formulas) + the second mapper's plain ones."""
    import bench
    c = {"tsdf_blocks_in_view": 400, "color_blocks_updated": 300, "blocks_allocated": 2000, "esdf_columns_marked": 200, "esdf_blocks_swept": 250,
         "b_tsdf_blocks_in_view": 10, "b_esdf_columns_marked": 3, "b_esdf_blocks_swept": 20, "b_blocks_allocated": 15}
    cb = {k[2:]: v for k, v in c.items() if k.startswith("b_")}; cb["color_blocks_updated"] = 0
    rows, cols = 480, 640
    assert bench.algorithmic_bytes("k_mark_view_pair", c, rows, cols, fused=True) == \
        bench.algorithmic_bytes("k_mark_view", c, rows, cols, fused=True) + bench.algorithmic_bytes("k_mark_view", cb, rows, cols)
    second = bench.algorithmic_bytes("k_integrate_tsdf_color", cb, rows, cols) - (rows * cols * 3 + (rows // 4) * (cols // 4) * 4)      # (no colour image on an occupancy mapper)
    assert bench.algorithmic_bytes("k_integrate_tsdf_color_pair", c, rows, cols, fused=True) == bench.algorithmic_bytes("k_integrate_tsdf_color", c, rows, cols) + second
    assert bench.short("void k_integrate_tsdf_color_pair<nvbx::DepthF32, nvbx::PixRgb8>(FusedArgs<...>)") == "k_integrate_tsdf_color_pair"
