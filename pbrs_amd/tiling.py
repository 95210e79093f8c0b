"""Multi-GPU partition of the frame: interleaved row bands, host-side gather, no collective on the
data path.

The reference parallelises over image rows with rayon and shares the scene read-only
(src/main.rs:219-224); every pixel-sample is independent.  Across G GPUs the frame's rows are cut
into bands of `band_rows` rows dealt round-robin to ranks (interleaving balances cost, which varies
smoothly over the image); each rank renders its bands as one packed tile (pbrs_render_params
band_*), and rank 0 reassembles the frame on the host.  The RNG stream is keyed by
(seed, film pixel index, sample index), so the frame is bit-identical for every G.
"""
import numpy as np

BAND_ROWS = 8


def owned_rows(height, world, rank, band_rows=BAND_ROWS):
    """Film rows of `rank`, in the packed order the renderer emits them."""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world == rank]


def packed_height(height, world, rank, band_rows=BAND_ROWS):
    return int(len(owned_rows(height, world, rank, band_rows)))


def render_share(render_fn, width, height, world, rank, band_rows=BAND_ROWS):
    """render_fn(tile=(x0, y0, w, h), bands=(band_rows, band_count, band_index)) -> (h, w, 3) array
    of this rank's packed rows."""
    h = packed_height(height, world, rank, band_rows)
    if h == 0:
        return np.zeros((0, width, 3), dtype=np.float32)
    bands = (band_rows, world, rank) if world > 1 else None
    return render_fn(tile=(0, 0, width, h), bands=bands)


def assemble(shares, width, height, world, band_rows=BAND_ROWS):
    """shares[r] = packed rows of rank r -> full (height, width, 3) frame."""
    frame = np.empty((height, width, 3), dtype=np.float32)
    for r in range(world):
        rows = owned_rows(height, world, r, band_rows)
        assert shares[r].shape == (len(rows), width, 3), (shares[r].shape, len(rows))
        frame[rows] = shares[r]
    return frame


def gather_frame(share, width, height, world, rank, group=None, band_rows=BAND_ROWS):
    """Host-side gather of the packed shares to rank 0 over torch.distributed (gloo: CPU tensors).
    Returns the frame on rank 0 and None elsewhere."""
    if world == 1:
        return assemble([share], width, height, 1, band_rows)
    import torch
    import torch.distributed as dist
    # gloo's gather wants equal shapes: pad every share to the tallest one (shares differ by at most one band)
    heights = [packed_height(height, world, r, band_rows) for r in range(world)]
    tallest = max(heights)
    mine = torch.zeros((tallest, width, 3), dtype=torch.float32)
    mine[:heights[rank]] = torch.from_numpy(np.ascontiguousarray(share))
    if rank == 0:
        bufs = [torch.empty((tallest, width, 3), dtype=torch.float32) for _ in range(world)]
        dist.gather(mine, gather_list=bufs, dst=0, group=group)
        return assemble([b.numpy()[:heights[r]] for r, b in enumerate(bufs)], width, height, world, band_rows)
    dist.gather(mine, gather_list=None, dst=0, group=group)
    return None


class SharedFrame:
    """The frame buffers of a one-node job in POSIX shared memory: every rank writes its own rows in place and bumps its
    sequence number in the same mapping; frame n is complete when every rank's number has reached n — nothing is sent.
    (`gather_frame` moves 12 B per pixel through gloo's TCP loopback, ≈30 ms for a 1024² frame on 8 ranks — more than
    half of a rank's render time at 8 GPUs; this costs a row scatter of the rank's own share plus a poll.)

    Two buffers, used alternately (frame n lives in buffer n % 2).  `publish(n)` returns only when EVERY rank has
    published frame n — rank 0 included — and rank 0 publishes frame n + 1 only after it is done with the view of frame n
    it was given (the view is valid until rank 0's next `publish`).  So when a fast rank writes frame n + 2 into the
    buffer frame n lived in, rank 0 has already entered publish(n + 1), i.e. let go of frame n: no rank ever writes into
    a buffer that is being read, and no extra "consumed" word is needed.  The poll backs off (sleep) so that ranks
    sharing cores do not starve the ones they wait for.  The file is unlinked as soon as every rank has mapped it: a crash
    leaks nothing.  All ranks must be on the node that created it: `create` returns None when /dev/shm cannot be used
    and the caller keeps `gather_frame`."""

    def __init__(self, path, width, height, world, rank, band_rows=BAND_ROWS):
        self.world, self.rank = world, rank
        fb = self._frame_bytes(width, height)
        self.frames = [np.memmap(path, dtype=np.float32, mode="r+", offset=k * fb, shape=(height, width, 3)) for k in range(2)]
        # one sequence number per rank behind the pixels: "my rows of frame n are in place"
        self.seq = np.memmap(path, dtype=np.int64, mode="r+", offset=2 * fb, shape=(world,))
        self.step = 0
        self.rows = owned_rows(height, world, rank, band_rows)
        self.wait_s = 0.0  # time spent in the poll of the last publish (bench.py reports it)

    @staticmethod
    def _frame_bytes(width, height):
        return (width * height * 3 * 4 + 4095) // 4096 * 4096

    @classmethod
    def create(cls, width, height, world, rank, group=None, band_rows=BAND_ROWS):
        import os
        import torch.distributed as dist
        name = [None]
        if rank == 0:
            try:
                path = f"/dev/shm/pbrs_frame_{os.getpid()}"
                with open(path, "wb") as f:
                    f.truncate(2 * cls._frame_bytes(width, height) + 8 * world)  # zero-filled: every sequence number starts at 0
                name[0] = path
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=0, group=group)
        sf, ok = None, False
        try:
            if name[0] is not None and os.path.exists(name[0]):  # a rank on another node does not see the file
                sf = cls(name[0], width, height, world, rank, band_rows=band_rows)
                ok = True
        except (OSError, ValueError):
            sf = None
        oks = [None] * world
        dist.all_gather_object(oks, ok, group=group)  # also: every rank that can map the file has done so by now
        if rank == 0 and name[0]:
            try:
                os.unlink(name[0])  # the mappings keep the memory alive; nothing is left behind if a rank dies
            except OSError:
                pass
        if not all(oks):
            if sf is not None:
                sf.close()
            return None
        return sf

    def publish(self, share, timeout=120.0):
        """Writes this rank's packed rows into the current frame and waits until every rank has done so.  Returns that
        frame on rank 0 (a view of the shared buffer, valid until rank 0's next publish) and None elsewhere.  x86 keeps a
        rank's row stores ahead of its sequence store, and a reader's sequence load ahead of its row loads."""
        import time
        self.step += 1
        frame = self.frames[self.step % 2]
        if len(self.rows):
            frame[self.rows] = share
        self.seq[self.rank] = self.step
        t0 = time.monotonic()
        deadline, pause = t0 + timeout, 0.0
        while int(self.seq.min()) < self.step:
            if time.monotonic() > deadline:
                raise RuntimeError(f"rank {self.rank}: frame {self.step} incomplete after {timeout} s (sequence numbers {self.seq.tolist()})")
            time.sleep(pause)  # 0 first (yield), then up to 200 us
            pause = min(200e-6, pause * 2 + 10e-6)
        self.wait_s = time.monotonic() - t0
        return frame if self.rank == 0 else None

    def close(self):
        self.frames = None
        self.seq = None
