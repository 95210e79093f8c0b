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
