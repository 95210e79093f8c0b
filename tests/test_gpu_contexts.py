"""Several contexts in one process (include/pbrs_gpu.h, "Threading"): the shape a Rust host takes when it drives one context per
device from one thread each (the reference's rayon workers share `&Scene`, src/main.rs:219-224).  Nothing a
context does may depend on what another context of the process holds: in particular the dynamic-LDS limit of the traversal
kernels is per-function state of the process and is raised once per device to the cap, not per uploaded scene."""
import ctypes
import threading

import numpy as np
import pytest

import pbrs_amd
from common import bits
from oracle.binding import OracleScene
from pbrs_amd import scenes, tiling

pytestmark = pytest.mark.gpu


def test_two_contexts_with_different_stack_depths_interleaved():
    """C4 (a 23-level BLAS: 26 KB of LDS stacks per block) and C2 (a few levels) in two contexts on device 0, renders
    interleaved, C2 uploaded last: each still gets the LDS its own walks need and both match the oracle bit for bit."""
    sb4, c4 = scenes.build_config("c4")
    sb2, c2 = scenes.build_config("c2")
    hs4, hs2 = pbrs_amd.HostScene(sb4), pbrs_amd.HostScene(sb2)
    assert hs4.stack_depth > 2 * hs2.stack_depth
    a, b = pbrs_amd.Context(0), pbrs_amd.Context(0)
    try:
        a.upload(hs4)
        b.upload(hs2)  # round 2: this upload lowered the limit of the kernels context `a` is about to launch
        w4, w2 = (900, 600, 64, 8), (480, 500, 64, 8)
        ref4, _ = OracleScene(sb4).render(2, 2, c4["depth"], 7, tile=w4)
        ref2, _ = OracleScene(sb2).render(2, 2, c2["depth"], 7, tile=w2)
        for _ in range(2):
            i4, _ = a.render(2, 2, c4["depth"], 7, tile=w4)
            i2, _ = b.render(2, 2, c2["depth"], 7, tile=w2)
            assert (bits(i4) == bits(ref4)).all()
            assert (bits(i2) == bits(ref2)).all()
        # asynchronous renders of both contexts in flight at once, each on its own stream, into caller-owned device memory
        hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself runs on
        nbytes = 8 * 64 * 3 * 4
        o4, o2 = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(o4), ctypes.c_size_t(nbytes)) == 0 and hip.hipMalloc(ctypes.byref(o2), ctypes.c_size_t(nbytes)) == 0
        try:
            a.render_device(o4.value, 2, 2, c4["depth"], 7, tile=w4)
            b.render_device(o2.value, 2, 2, c2["depth"], 7, tile=w2)
            a.collect_stats()  # waits for the context's own stream: the output is valid from here on
            b.collect_stats()
            h4, h2 = np.empty((8, 64, 3), dtype=np.float32), np.empty((8, 64, 3), dtype=np.float32)
            assert hip.hipMemcpy(ctypes.c_void_p(h4.ctypes.data), o4, ctypes.c_size_t(nbytes), 2) == 0  # hipMemcpyDeviceToHost
            assert hip.hipMemcpy(ctypes.c_void_p(h2.ctypes.data), o2, ctypes.c_size_t(nbytes), 2) == 0
            assert (bits(h4) == bits(ref4)).all() and (bits(h2) == bits(ref2)).all()
        finally:
            hip.hipFree(o4)
            hip.hipFree(o2)
    finally:
        a.close()
        b.close()


def test_two_threads_two_contexts_disjoint_bands_of_one_frame():
    """Two host threads, each with its own context on device 0, render the two interleaved band sets of one C3 frame at the same
    time (ctypes releases the GIL inside the C ABI); the assembled frame equals the frame of a single context bit for bit."""
    sb, c = scenes.build_config("c3", width=256, height=192)
    hs = pbrs_amd.HostScene(sb)
    W, H, depth = c["width"], c["height"], c["depth"]
    one = pbrs_amd.Context(0)
    one.upload(hs)
    whole, _ = one.render(4, 4, depth, 5)
    one.close()
    shares, errors = [None, None], []

    def work(rank):
        try:
            ctx = pbrs_amd.Context(0)
            ctx.upload(hs)
            rows = tiling.packed_height(H, 2, rank)
            for _ in range(3):  # several frames each, so that the two threads' calls overlap in time
                img, _ = ctx.render(4, 4, depth, 5, tile=(0, 0, W, rows), bands=(tiling.BAND_ROWS, 2, rank))
            shares[rank] = img
            ctx.close()
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    frame = tiling.assemble(shares, W, H, 2)
    assert (bits(frame) == bits(whole)).all()


def test_the_environment_does_not_change_a_render(monkeypatch):
    """The developer overrides of round 2 (kernel selection, pass layout) are compiled out of the shipped library: a context
    created, loaded and run under them renders the same frame in the same number of passes and stage launches."""
    sb, c = scenes.build_config("c3", width=128, height=96)
    hs = pbrs_amd.HostScene(sb)
    base = pbrs_amd.Context(0)
    base.upload(hs)
    ref, st0 = base.render(4, 4, c["depth"], 3, timing=True)
    base.close()
    for k, v in {"PBRS_SORT_CLASSES": "0", "PBRS_SPLIT_LAMBERT": "0", "PBRS_RAYGEN_CHUNK": "0", "PBRS_RAYGEN_TILES8": "0", "PBRS_REFILL_BELOW": "64",
                 "PBRS_LONG_WALKS": "1", "PBRS_SHADE_SPEC": "0", "PBRS_LDS_MIN": "60000"}.items():
        monkeypatch.setenv(k, v)
    ctx = pbrs_amd.Context(0)
    ctx.upload(hs)
    img, st1 = ctx.render(4, 4, c["depth"], 3, timing=True)
    ctx.close()
    assert (bits(img) == bits(ref)).all()
    for k in ("passes", "launches_extend", "launches_shade", "launches_shadow"):
        assert st0[k] == st1[k]


@pytest.mark.parametrize("n_threads", [3, 8])
def test_n_threads_n_contexts_assemble_the_single_context_frame(n_threads):
    """pbrs_amd/threads.py, the in-process driver INTEGRATION.md §2 describes for a Rust host (one context per device, one host
    thread each, interleaved 8-row bands, one frame buffer): N contexts — here all on device 0, on a node one per GPU — render
    one C3 frame together and the frame equals a single context's bit for bit, for a thread count that divides the bands
    unevenly (3) and for the node's eight.  The report carries what a scaling run needs to explain itself: rows and bands per
    thread and the band imbalance."""
    from pbrs_amd import threads
    sb, c = scenes.build_config("c3", width=192, height=136)  # 17 bands of 8 rows: 8 threads get 3 / 2 bands, 3 threads 6 / 6 / 5
    hs = pbrs_amd.HostScene(sb)
    one = pbrs_amd.Context(0)
    one.upload(hs)
    whole, _ = one.render(3, 3, c["depth"], 11)
    one.close()
    tf = threads.ThreadedFrame(hs, [0] * n_threads)
    try:
        for _ in range(2):  # consecutive frames through the same contexts
            frame, rep = tf.render(3, 3, c["depth"], 11)
            assert (bits(frame) == bits(whole)).all()
    finally:
        tf.close()
    rows = [t["rows"] for t in rep["threads"]]
    assert sum(rows) == 136 and len(rows) == n_threads
    assert max(t["bands"] for t in rep["threads"]) - min(t["bands"] for t in rep["threads"]) <= 1
    assert abs(rep["band_imbalance"] - max(rows) / (136 / n_threads)) < 1e-9
    assert all(t["gpu_ms"] > 0 for t in rep["threads"])


def test_a_callers_stream_orders_the_overlapped_passes():
    """pbrs_set_stream: the pipeline runs in the order of a stream the host owns (here one of torch's).  With several passes a pass's
    late bounces run on the context's second stream — forked from and joined back into the caller's stream by events — so work the
    caller queues on ITS stream after pbrs_render_tile_device (a device-to-host copy, no pbrs_collect_stats) sees the finished
    frame; two such frames back to back, then the context's own stream again."""
    import torch
    sb, c = scenes.build_config("c3", width=128, height=96)
    hs = pbrs_amd.HostScene(sb)
    ref, _ = OracleScene(sb).render(3, 3, c["depth"], 31)
    ctx = pbrs_amd.Context(0)
    try:
        ctx.upload(hs)
        stream = torch.cuda.Stream(device="cuda:0")
        ctx.set_stream(stream.cuda_stream)
        dev = torch.zeros((96, 128, 3), dtype=torch.float32, device="cuda:0")
        host = [torch.empty((96, 128, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        with torch.cuda.stream(stream):
            for k in range(2):
                ctx.render_device(dev.data_ptr(), 3, 3, c["depth"], 31, samples_per_pass=2)  # five passes, overlapped
                host[k].copy_(dev, non_blocking=True)  # ordered after the frame by the stream alone
                dev.zero_()                            # ... and before the next frame's writes
        stream.synchronize()
        for k in range(2):
            assert (bits(host[k].numpy()) == bits(ref)).all(), k
        ctx.set_stream(0)
        img, _ = ctx.render(3, 3, c["depth"], 31, samples_per_pass=2)
        assert (bits(img) == bits(ref)).all()
    finally:
        ctx.close()
