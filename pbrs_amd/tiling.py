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
    """The frame buffer of a one-node job in POSIX shared memory: every rank writes its own rows in place and bumps its
    sequence number in the same mapping; the frame is complete when every rank's number has reached it — nothing is sent.
    (`gather_frame` moves 12 B per pixel through gloo's TCP loopback, ≈30 ms for a 1024² frame on 8 ranks — more than
    half of a rank's render time at 8 GPUs; this costs a row scatter of the rank's own share plus a poll.)  All ranks
    must be on the node that created it: `create` falls back to None when /dev/shm cannot be used, and the caller then
    keeps `gather_frame`."""

    def __init__(self, path, width, height, world, rank, owner, band_rows=BAND_ROWS):
        self.path, self.owner, self.world, self.rank = path, owner, world, rank
        self.frame = np.memmap(path, dtype=np.float32, mode="r+", shape=(height, width, 3))
        # one sequence number per rank behind the pixels: "my rows of frame n are in place"
        self.seq = np.memmap(path, dtype=np.int64, mode="r+", offset=self._frame_bytes(width, height), shape=(world,))
        self.step = 0
        self.rows = owned_rows(height, world, rank, band_rows)

    @staticmethod
    def _frame_bytes(width, height):
        return (width * height * 3 * 4 + 63) // 64 * 64

    @classmethod
    def create(cls, width, height, world, rank, group=None, band_rows=BAND_ROWS):
        import os
        import torch.distributed as dist
        name = [None]
        if rank == 0:
            try:
                path = f"/dev/shm/pbrs_frame_{os.getpid()}"
                with open(path, "wb") as f:
                    f.truncate(cls._frame_bytes(width, height) + 8 * world)  # zero-filled: every sequence number starts at 0
                name[0] = path
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=0, group=group)
        ok = [name[0] is not None and os.path.exists(name[0])]  # a rank on another node does not see the file
        oks = [None] * world
        dist.all_gather_object(oks, ok[0], group=group)
        if not all(oks):
            if rank == 0 and name[0]:
                os.unlink(name[0])
            return None
        return cls(name[0], width, height, world, rank, owner=(rank == 0), band_rows=band_rows)

    def publish(self, share, group=None, timeout=120.0):
        """Writes this rank's packed rows into the frame and waits until every rank has done so for this frame.  Returns
        the frame (a view of the shared buffer) on rank 0, None elsewhere.  The wait is a poll of the ranks' sequence
        numbers in the same shared mapping (tens of microseconds; a gloo barrier is ≈1 ms at 8 ranks): x86 keeps a
        rank's row stores ahead of its sequence store, and a reader's sequence load ahead of its row loads."""
        import time
        if len(self.rows):
            self.frame[self.rows] = share
        self.step += 1
        self.seq[self.rank] = self.step
        deadline = time.monotonic() + timeout
        while int(self.seq.min()) < self.step:
            if time.monotonic() > deadline:
                raise RuntimeError(f"rank {self.rank}: frame {self.step} incomplete after {timeout} s (sequence numbers {self.seq.tolist()})")
        return self.frame if self.rank == 0 else None

    def close(self):
        import os
        del self.frame
        del self.seq
        if self.owner:
            try:
                os.unlink(self.path)
            except OSError:
                pass
