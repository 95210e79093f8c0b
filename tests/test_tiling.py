"""Multi-GPU partition (pbrs_amd/tiling.py): interleaved row bands + host-side gather.  The N > 1 path is
exercised with two gloo ranks on the CPU; the per-rank renderer is the oracle here (test infrastructure), which
implements the same `tile` + `bands` contract as pbrs_render_params."""
import os
import socket

import numpy as np
import pytest

from pbrs_amd import tiling


def test_partition_covers_every_row_once():
    for height in (1, 7, 8, 9, 64, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            rows = np.concatenate([tiling.owned_rows(height, world, r) for r in range(world)])
            assert sorted(rows.tolist()) == list(range(height))
            sizes = [tiling.packed_height(height, world, r) for r in range(world)]
            assert max(sizes) - min(sizes) <= tiling.BAND_ROWS


def test_assemble_inverts_the_split():
    rs = np.random.RandomState(0)
    frame = rs.rand(37, 5, 3).astype(np.float32)
    for world in (1, 2, 4):
        shares = [frame[tiling.owned_rows(37, world, r)] for r in range(world)]
        assert (tiling.assemble(shares, 5, 37, world) == frame).all()


def _oracle_band_renderer(osc, sx, sy, depth, seed, height):
    """Oracle stand-in for Context.render: renders the packed rows of a band set row by row."""
    def render(tile, bands):
        x0, y0, w, h = tile
        if bands is None:
            return osc.render(sx, sy, depth, seed, tile=tile, nthreads=1)[0]
        band_rows, count, index = bands
        out = np.empty((h, w, 3), dtype=np.float32)
        for vr in range(h):
            row = y0 + ((vr // band_rows) * count + index) * band_rows + vr % band_rows
            out[vr] = osc.render(sx, sy, depth, seed, tile=(x0, row, w, 1), nthreads=1)[0][0]
        return out
    return render


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from common import SEED, golden_case, load_golden
    from oracle.binding import OracleScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb, (w, h, sx, sy, depth) = golden_case("c2_cornell_diffuse")
    osc = OracleScene(sb)
    share = tiling.render_share(_oracle_band_renderer(osc, sx, sy, depth, SEED, h), w, h, world, rank)
    frame = tiling.gather_frame(share, w, h, world, rank)
    shared = tiling.SharedFrame.create(w, h, world, rank)  # the one-node path: rows written in place, one barrier
    assert shared is not None
    frame2 = shared.publish(share)
    if rank == 0:
        g = load_golden("c2_cornell_diffuse")
        q.put(bool((frame.view(np.uint32) == g["image"].view(np.uint32)).all() and (np.asarray(frame2).view(np.uint32) == g["image"].view(np.uint32)).all()))
    dist.barrier()
    shared.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_reproduces_the_single_process_frame():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok, "frame gathered from 2 ranks differs from the single-process frame"
