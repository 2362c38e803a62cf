"""Ownership transfer of colour images (csrc/frames.hip, include/nvblox_hip.h nvbx_frame_*): an image that lives in a library-owned frame is RETAINED by
a mapper that holds integrateColor back -- no staging copy -- and the writer rotates to another frame while the mapper (or its launches) still use the
last one.  Same calls, same map as the classic order, bit for bit; no k_stage_color launch; a host that scribbles over "its" image right after
integrateColor cannot reach the held-back frame.  Call site served: nvblox_ros/src/lib/nvblox_node.cpp:1237-1264."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _bit_equal(M, a, b, tag=""):
    from test_gpu_pipeline import _equal_maps
    _equal_maps(M, a, b, tag)


def _count(prof, name):
    return sum(v["count"] for k_, v in prof.items() if name in k_)


@pytest.mark.parametrize("writer", ["mapper_stream", "other_stream", "host"])
@pytest.mark.parametrize("cam", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_recycled_image_in_a_library_frame_is_not_copied_and_not_lost(oracle_mod, hip_lib, cam, writer):
    """The node's ONE colour image, refilled right before every integrateColor and scribbled over right after it -- on the mapper's own stream (the
    reference's converter), on another stream, or by a blocking host copy.  The mapper runs its default setting (staged deferral)."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    dev = torch.device("cuda", 0)
    ms = torch.cuda.Stream(dev); other = torch.cuda.Stream(dev)
    pg = M.default_params()
    with torch.cuda.stream(ms):
        classic = M.Mapper(pg, block_capacity=1 << 13, stream=ms.cuda_stream); owned = M.Mapper(pg, block_capacity=1 << 13, stream=ms.cuda_stream)
        classic.set_color_deferral(False)
        owned.set_profiling(True)                         # (a new mapper: deferral on, staged form)
        img = M.ColorFrame(cam[5], cam[4], 3, 0)
        noise = torch.randint(0, 255, (cam[5], cam[4], 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        stats0 = M.frame_pool_stats()
        wstream = {"mapper_stream": ms.cuda_stream, "other_stream": other.cuda_stream, "host": None}[writer]
        seen = set()
        for k, (d, rgb, T) in enumerate(H.frames(12, cam, stride=7)):
            d_dev = torch.from_numpy(d).to(dev)
            rgb_dev = torch.from_numpy(rgb).to(dev) if writer != "host" else None
            torch.cuda.synchronize(dev)                   # (the uploads above are torch's business, not the subject)
            classic.integrate_depth(d_dev, T, cam); classic.integrate_color(torch.from_numpy(rgb).to(dev), T, cam); classic.update_esdf()
            owned.integrate_depth(d_dev, T, cam)
            img.write(rgb_dev if rgb_dev is not None else rgb, wstream)
            if writer == "other_stream":
                ms.wait_stream(other)                     # the caller orders ITS write before the mapper's reads (as with any buffer it fills elsewhere)
            seen.add(img.ptr)
            owned.integrate_color(img, T, cam)
            assert img.shared()                           # the mapper holds the frame: nothing was copied
            img.write(noise, wstream)                     # recycled at once: must land in ANOTHER frame
            assert not img.shared()
            owned.update_esdf()
        classic.synchronize(); owned.synchronize()
    _bit_equal(M, classic, owned, "frames / %s" % writer)
    prof = owned.profile()
    assert _count(prof, "k_stage_color") == 0, {k_: v["count"] for k_, v in prof.items()}
    assert _count(prof, "k_integrate_tsdf_color") >= 10           # two launches per frame all along
    assert len(seen) >= 2                                         # the image did rotate
    held, free, nbytes, created, waits, syncs = M.frame_pool_stats()
    assert created - stats0[3] <= 10 and syncs == stats0[5], (created - stats0[3], waits - stats0[4], syncs - stats0[5])
    img.close()


def test_raw_pointers_are_still_staged_and_pool_frames_are_reused(oracle_mod, hip_lib):
    """Raw device pointers under the default setting: one k_stage_color launch per held-back frame, into frames of the same pool (only as many as
    are in flight, not MAX_BATCH per mapper -- ADVICE r04), and a batch of 2 costs ONE copy launch."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    lib = hip_lib
    lib.nvbx_frame_pool_trim(-1)
    c0 = M.frame_pool_stats()[3]
    g = M.Mapper(M.default_params(), block_capacity=1 << 13); g.set_profiling(True)
    fr = H.frames(8, cam, stride=9)
    for d, rgb, T in fr:
        g.integrate_depth(d, T, cam); g.integrate_color(rgb, T, cam); g.update_esdf()
    g.integrate_depth_batch([fr[0][0], fr[1][0]], [fr[0][2], fr[1][2]], cam)
    g.integrate_color_batch([fr[0][1], fr[1][1]], [fr[0][2], fr[1][2]], cam)
    g.synchronize()
    prof = g.profile()
    assert _count(prof, "k_stage_color") == 9, {k_: v["count"] for k_, v in prof.items()}
    held, free, nbytes, created, waits, syncs = M.frame_pool_stats()
    assert held == 0 and created - c0 <= 4, (held, free, created - c0)          # same-stream reuse: the copy of frame i+1 lands in the frame of i-1
    g.close()
    assert lib.nvbx_frame_pool_trim(-1) >= 1 and M.frame_pool_stats()[1] == 0


def test_frame_api_contract(oracle_mod, hip_lib):
    """Reference counting, writability, errors, and nvbx_integrate_color_owned."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M, _lib
    lib = hip_lib
    cam = H.SMALL_CAM
    nbytes = cam[4] * cam[5] * 3
    p = C.c_void_p()
    assert lib.nvbx_frame_acquire(0, nbytes, _lib.STREAM_UNKNOWN, C.byref(p)) == 0 and p.value
    assert lib.nvbx_frame_refcount(p) == 1 and lib.nvbx_frame_writable(p, _lib.STREAM_UNKNOWN) == 1
    assert lib.nvbx_frame_retain(p) == 0 and lib.nvbx_frame_refcount(p) == 2 and lib.nvbx_frame_writable(p, _lib.STREAM_UNKNOWN) == 0
    assert lib.nvbx_frame_release(p) == 0 and lib.nvbx_frame_refcount(p) == 1
    bogus = C.c_void_p(p.value + 64)
    assert lib.nvbx_frame_refcount(bogus) == -1 and lib.nvbx_frame_retain(bogus) < 0 and lib.nvbx_frame_release(bogus) < 0
    assert lib.nvbx_frame_acquire(0, 0, _lib.STREAM_UNKNOWN, C.byref(C.c_void_p())) < 0
    g = M.Mapper(M.default_params(), block_capacity=1 << 13); ref = M.Mapper(M.default_params(), block_capacity=1 << 13); ref.set_color_deferral(False)
    d, rgb, T = H.frames(1, cam)[0]
    k = M.Camera(*[float(v) for v in cam[:4]], int(cam[4]), int(cam[5])); Tm = np.ascontiguousarray(np.asarray(T, np.float32).reshape(4, 4))
    for m_ in (g, ref):
        m_.integrate_depth(d, T, cam)
    ref.integrate_color(rgb, T, cam)
    # a frame smaller than the image is an argument error of the call that made it, and the caller keeps the frame
    small = C.c_void_p(); assert lib.nvbx_frame_acquire(0, 1000, _lib.STREAM_UNKNOWN, C.byref(small)) == 0
    assert lib.nvbx_integrate_color_owned(g._h, small, 3, cam[5], cam[4], Tm.ctypes.data_as(C.c_void_p), C.byref(k)) < 0
    assert lib.nvbx_frame_refcount(small) == 1 and lib.nvbx_frame_release(small) == 0
    # ownership passes with the call: the converter's frame (nvbx_color_image_acquire) is the mapper's afterwards, and returns to the pool when it is done
    q = C.c_void_p(); assert lib.nvbx_color_image_acquire(g._h, cam[5], cam[4], 3, C.byref(q)) == 0
    host = np.ascontiguousarray(rgb); assert lib.nvbx_frame_upload(q, host.ctypes.data_as(C.c_void_p), host.nbytes, C.c_void_p(g.stream_handle())) == 0
    assert lib.nvbx_integrate_color_owned(g._h, q, 3, cam[5], cam[4], Tm.ctypes.data_as(C.c_void_p), C.byref(k)) == 0
    assert lib.nvbx_frame_refcount(q) == 1                    # the mapper's (held back)
    g.synchronize()
    assert lib.nvbx_frame_refcount(q) == 0                    # carried out: back in the pool
    _bit_equal(M, ref, g, "owned")
    # deferral off: the frame is read by launches enqueued at once and let go of behind them
    g.set_color_deferral(False)
    q2 = C.c_void_p(); assert lib.nvbx_color_image_acquire(g._h, cam[5], cam[4], 3, C.byref(q2)) == 0
    assert lib.nvbx_frame_upload(q2, host.ctypes.data_as(C.c_void_p), host.nbytes, C.c_void_p(g.stream_handle())) == 0
    assert lib.nvbx_integrate_color_owned(g._h, q2, 3, cam[5], cam[4], Tm.ctypes.data_as(C.c_void_p), C.byref(k)) == 0
    assert lib.nvbx_frame_refcount(q2) == 0
    ref.integrate_color(rgb, T, cam)
    _bit_equal(M, ref, g, "owned, classic order")
    assert lib.nvbx_frame_release(p) == 0
    g.close(); ref.close()
    torch.cuda.synchronize()


def test_pool_backpressure_when_the_host_runs_far_ahead(oracle_mod, hip_lib, monkeypatch):
    """A host that is several frames ahead of the GPU (here: the mapper's stream is kept busy by a ~0.2 s spin kernel first) with a writer on a
    stream the mapper knows nothing about: the pool grows to its cap (2 here), then nvbx_frame_acquire WAITS for the oldest fence -- it polls the
    progress word, the queue is not drained -- and the map still equals the classic one."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    cam = S.REPLICA_LIKE_CAM
    hip_lib.nvbx_frame_pool_trim(-1)
    monkeypatch.setenv("NVBX_FRAME_POOL_MAX", "2")      # (read where a frame would be created)
    dev = torch.device("cuda", 0)
    ms = torch.cuda.Stream(dev); aux = torch.cuda.Stream(dev)
    pg = M.default_params()
    classic = M.Mapper(pg, block_capacity=1 << 13); owned = M.Mapper(pg, block_capacity=1 << 13, stream=ms.cuda_stream)
    classic.set_color_deferral(False)
    fr = H.frames(6, cam, stride=11)
    d_dev = [torch.from_numpy(d).to(dev) for d, _, _ in fr]; c_dev = [torch.from_numpy(c).to(dev) for _, c, _ in fr]
    torch.cuda.synchronize(dev)
    for k in range(6):
        classic.integrate_depth(d_dev[k], fr[k][2], cam); classic.integrate_color(c_dev[k], fr[k][2], cam); classic.update_esdf()
    classic.synchronize()
    # (two frames first, then an empty map again: the lazily allocated buffers of the pipelined frame exist -- a hipMalloc in the middle of the
    #  loop below would wait for the busy stream and the host would never get ahead)
    for k in range(2):
        owned.integrate_depth(d_dev[k], fr[k][2], cam); owned.integrate_color(c_dev[k], fr[k][2], cam); owned.update_esdf()
    owned.integrate_depth(d_dev[2], fr[2][2], cam); owned.clear(); owned.synchronize()
    stats0 = M.frame_pool_stats()
    img = M.ColorFrame(cam[5], cam[4], 3, 0)
    with torch.cuda.stream(ms):
        torch.cuda._sleep(int(4e8))                      # the mapper's stream is busy (a spin kernel, ~0.2 s): every launch below queues up behind this
    for k in range(6):
        T = fr[k][2]
        owned.integrate_depth(d_dev[k], T, cam)
        img.write(c_dev[k], aux.cuda_stream)             # a stream the mapper knows nothing about
        aux.synchronize()
        owned.integrate_color(img, T, cam); owned.update_esdf()
    owned.synchronize()
    _bit_equal(M, classic, owned, "back-pressure")
    held, free, nbytes, created, waits, syncs = M.frame_pool_stats()
    assert held == 1 and free <= 2 and created - stats0[3] <= 3 and waits > stats0[4], (held, free, created - stats0[3], waits - stats0[4], syncs - stats0[5])
    img.close(); classic.close(); owned.close()


def test_a_frame_let_go_of_without_a_fence_waits_for_the_streams_the_library_knows(oracle_mod, hip_lib):
    """nvbx_frame_release used to put the frame back into the pool at once (hipFree, which the reference's buffers end in, waits for the
    device).  Now the release that brings the count to 0 records an event on every stream the library knows -- here a live mapper's stream, kept busy
    by a spin kernel -- and nvbx_frame_acquire skips the frame until the events are reached; a writer on that very stream may have it (stream order).
    nvbx_frame_release_on names the one stream instead.  nvbx_frame_device answers where a frame lives."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M, _lib
    lib = hip_lib
    dev = torch.device("cuda", 0)
    ms = torch.cuda.Stream(dev)
    g = M.Mapper(M.default_params(), block_capacity=1 << 12, stream=ms.cuda_stream)
    g.synchronize(); torch.cuda.synchronize(dev)
    lib.nvbx_frame_pool_trim(-1)
    nbytes = 777 * 1024                                  # (a size class of this test's own)
    def acquire(stream=_lib.STREAM_UNKNOWN):
        q = C.c_void_p(); assert lib.nvbx_frame_acquire(0, nbytes, stream, C.byref(q)) == 0 and q.value; return q
    msh = C.c_void_p(ms.cuda_stream)
    for release_on in (False, True):
        p = acquire()
        assert lib.nvbx_frame_device(p) == 0 and lib.nvbx_frame_device(C.c_void_p(p.value + 64)) == -1
        with torch.cuda.stream(ms):
            torch.cuda._sleep(int(3e8))                  # the mapper's stream is busy for ~0.15 s
        assert (lib.nvbx_frame_release_on(p, msh) if release_on else lib.nvbx_frame_release(p)) == 0
        assert lib.nvbx_frame_refcount(p) == 0
        assert not ms.query()                            # (still busy: the checks below mean something)
        q = acquire()                                    # a writer on an unknown stream: OTHER memory
        assert q.value != p.value, "a frame whose stream is still busy was handed out"
        if release_on:                                   # a writer on the very stream the release named: stream order does it
            r = acquire(msh)
            assert r.value == p.value
            assert lib.nvbx_frame_release_on(r, msh) == 0
        assert not ms.query()
        ms.synchronize()
        assert lib.nvbx_frame_release(q) == 0
        torch.cuda.synchronize(dev)                      # (q's own events, just recorded on idle streams, have been reached too)
        got = {acquire().value for _ in range(2)}        # idle: both frames are cool again
        assert got == {p.value, q.value}, (got, p.value, q.value)
        for v in got:
            assert lib.nvbx_frame_release(C.c_void_p(v)) == 0
        torch.cuda.synchronize(dev)
        lib.nvbx_frame_pool_trim(-1)
    g.close()
