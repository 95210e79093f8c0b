"""One process, N host threads, N contexts: the shape a Rust host takes.

The reference parallelises its frame loop over rows with rayon inside ONE process (src/main.rs:219-224: `par_iter` over the rows,
`&Scene` shared read-only).  Its drop-in counterpart keeps that shape: one `pbrs_ctx` per device, each driven by its own host thread,
the frame's rows dealt to the contexts in interleaved 8-row bands (pbrs_amd/tiling.py), every thread copying its packed rows into
the one frame buffer of the process — no inter-process transport at all, no collective.  `bench.py` launches one process per GPU
because the driver's contract says so; this module is the in-process alternative, with the same partition and therefore the same
frame, bit for bit, for every N (the RNG stream is keyed by film pixel and sample index).

ctypes releases the GIL inside the C ABI, so the threads' renders overlap exactly as a native host's would.
"""
import threading
import time

import numpy as np

from . import api, tiling


class ThreadedFrame:
    """N contexts on `devices` (one per thread; devices may repeat: several contexts on one device is legal and is how the
    suite tests this on a one-GPU box), the scene uploaded to each, frames rendered by all threads at once."""

    def __init__(self, host_scene, devices, band_rows=tiling.BAND_ROWS):
        self.hs = host_scene
        self.devices = list(devices)
        self.world = len(self.devices)
        self.band_rows = band_rows
        self.ctxs = [None] * self.world
        errors = []

        def setup(r):
            try:
                ctx = api.Context(self.devices[r])
                ctx.upload(host_scene)
                self.ctxs[r] = ctx
            except Exception as e:  # noqa: BLE001 (re-raised by the caller's thread)
                errors.append((r, e))

        self._run(setup)
        if errors:
            self.close()
            raise errors[0][1]

    def _run(self, fn):
        threads = [threading.Thread(target=fn, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()

    def render(self, strata_x, strata_y, depth, seed, integrator="path", samples_per_pass=0):
        """One frame.  Returns (frame (H, W, 3) float32, report): per-thread rows, bands, wall milliseconds of its render call and
        the hand-over of its rows, the frame's wall time and the band imbalance (most rows of a thread over the mean)."""
        W, H = self.hs.width, self.hs.height
        frame = np.empty((H, W, 3), dtype=np.float32)
        per = [None] * self.world
        errors = []

        def work(r):
            try:
                rows = tiling.owned_rows(H, self.world, r, self.band_rows)
                t0 = time.perf_counter()
                if len(rows):
                    bands = (self.band_rows, self.world, r) if self.world > 1 else None
                    img, st = self.ctxs[r].render(strata_x, strata_y, depth, seed, tile=(0, 0, W, len(rows)), bands=bands, integrator=integrator,
                                                  samples_per_pass=samples_per_pass, timing=True)
                    t1 = time.perf_counter()
                    frame[rows] = img  # disjoint rows per thread: no lock
                    gpu_ms = st["ms_total"]
                else:
                    t1, gpu_ms = t0, 0.0
                t2 = time.perf_counter()
                per[r] = {"thread": r, "device": self.devices[r], "rows": int(len(rows)), "bands": int(-(-len(rows) // self.band_rows)),
                          "gpu_ms": float(gpu_ms), "render_call_ms": (t1 - t0) * 1e3, "handover_ms": (t2 - t1) * 1e3}
            except Exception as e:  # noqa: BLE001
                errors.append((r, e))

        t0 = time.perf_counter()
        self._run(work)
        wall = (time.perf_counter() - t0) * 1e3
        if errors:
            raise errors[0][1]
        rows = [p["rows"] for p in per]
        return frame, {"threads": per, "frame_wall_ms": wall, "band_imbalance": max(rows) / (sum(rows) / len(rows)) if sum(rows) else 1.0}

    def close(self):
        for c in self.ctxs:
            if c is not None:
                c.close()
        self.ctxs = [None] * self.world
