#!/usr/bin/env python3
"""bench.py — Msamples/s (+ Mrays/s) of the MI355X wavefront path tracer on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W          (N = 1: plain process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one whole frame of the workload: every rank renders its interleaved row bands of the frame
on its own GPU (scene already resident in HBM), copies them to the host and rank 0 gathers them into
the frame buffer (host-side gather over gloo; no RCCL: pixels are independent).
Total work is fixed as N grows, so scaling is reported as "strong".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", help="c1..c5 (BASELINE.json configs[0..4]); default c2 = configs[1]")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--strata", type=int, nargs=2, default=None, metavar=("SX", "SY"))
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--integrator", default="path", choices=["path", "direct"],
                    help="path = src/pathintegrator.rs (the BASELINE metric); direct = direct_lighting_integrator (src/directlighting.rs:14-47)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, nargs=3, default=None, metavar=("W", "H", "MSAA"), help="CPU baseline sample (default 384 384 4)")
    return ap.parse_args()


def host_cores():
    """CPU share of this process: scheduler affinity capped by the cgroup quota (a 1-GPU box grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("PBRS_CPU_THREADS", "16"))))


def cpu_baseline(config_name, depth, seed, sample, integrator="path"):
    """The oracle (C++ restatement of the reference, "port") timed on this box's host cores on a bounded
    sample of the same scene; row-parallel like the reference's rayon loop (src/main.rs:219-224)."""
    from oracle.binding import OracleScene
    from pbrs_amd import scenes
    w, h, msaa = sample
    sb, _ = scenes.build_config(config_name, width=w, height=h)
    osc = OracleScene(sb)
    cores = host_cores()
    t = time.perf_counter()
    _, st = osc.render(msaa, msaa, depth, seed, nthreads=cores, integrator=integrator)
    dt = time.perf_counter() - t
    return {
        "value": st["samples"] / dt / 1e6,
        "unit": "Msamples/s",
        "mrays_per_s": (st["closest_rays"] + st["shadow_rays"]) / dt / 1e6,
        "cores": cores,
        "kind": "port",
        "sample": f"{config_name} scene at {w}x{h}, {msaa * msaa} spp, depth {depth}: {st['samples']} samples in {dt:.2f} s "
                  f"(C++ restatement of pbrs, g++ -O2 -ffp-contract=off, {cores} threads over rows)",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

    import pbrs_amd
    from pbrs_amd import roofline, scenes, tiling

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)  # host-side barrier/gather only
    # one rank per GPU; a rehearsal with more ranks than GPUs (development box) shares the devices round-robin
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sb, cfg = scenes.build_config(args.config, width=args.width, height=args.height)
    if args.strata:
        cfg["strata_x"], cfg["strata_y"] = args.strata
    if args.depth:
        cfg["depth"] = args.depth
    W, H, sx, sy, depth = cfg["width"], cfg["height"], cfg["strata_x"], cfg["strata_y"], cfg["depth"]
    spp = sx * sy
    hs = pbrs_amd.HostScene(sb)
    ctx = pbrs_amd.Context(local_rank)
    ctx.upload(hs)  # scene resident in HBM before any timed region

    my_rows = tiling.packed_height(H, world, rank)
    out_dev = torch.empty((max(my_rows, 1), W, 3), dtype=torch.float32, device=dev)
    out_host = torch.empty((max(my_rows, 1), W, 3), dtype=torch.float32).pin_memory()
    bands = (tiling.BAND_ROWS, world, rank) if world > 1 else None

    # one node: the ranks write their rows into a frame in shared memory (no data through gloo); otherwise gloo gather
    shared = tiling.SharedFrame.create(W, H, world, rank) if world > 1 else None

    def step(timing=False, counters=False):
        if my_rows:
            ctx.render_device(out_dev.data_ptr(), sx, sy, depth, args.seed, tile=(0, 0, W, my_rows), bands=bands,
                              samples_per_pass=args.samples_per_pass, timing=timing, counters=counters, integrator=args.integrator)
        torch.cuda.synchronize()
        out_host.copy_(out_dev)
        if shared is not None:
            return shared.publish(out_host.numpy()[:my_rows])
        return tiling.gather_frame(out_host.numpy()[:my_rows], W, H, world, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the first frame allocates the path state (tens of GB): it is never a timed one, whatever --warmup says
    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    stage_ms = None
    for _ in range(args.steps):
        step(timing=True)
        st = ctx.collect_stats() if my_rows else None
        if st is not None:
            if stage_ms is None:
                stage_ms = {k: 0.0 for k in st if k.startswith("ms_")}
                launches = {k: 0 for k in st if k.startswith("launches_")}
            for k in stage_ms:
                stage_ms[k] += st[k]
            for k in launches:
                launches[k] += st[k]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Untimed: instrumented frame for ray / node / primitive counts (deterministic, equal to the timed work).
    frame = step(counters=True)
    cst = ctx.collect_stats() if my_rows else None
    counts = np.array([cst["closest_rays"], cst["shadow_rays"], cst["samples"]] if cst else [0, 0, 0], dtype=np.float64)
    if world > 1:
        tc = torch.from_numpy(counts)
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        counts = tc.numpy()

    if rank == 0:
        samples_per_step = float(W) * H * spp
        assert counts[2] == samples_per_step, (counts, samples_per_step)
        rays_per_step = counts[0] + counts[1]
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step * args.steps / elapsed / 1e6
        times = dict(stage_ms)
        times.update(launches)
        rep = roofline.stage_report(cst, {k: (v / args.steps if k.startswith("ms_") else v // args.steps) for k, v in times.items()})
        dom_name, dom = roofline.dominant(rep)
        # HBM bytes per launch from the PMC counters cannot be collected in-process (rocprofv3 owns the counters); the
        # last measured figure for this workload, if any, is read from profiles/ (tools/traffic_from_pmc.py).
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"latest_traffic_{args.config}.json")
        full_size = (W, H, sx, sy, depth) == tuple(scenes.CONFIGS[args.config][2:7]) and args.integrator == "path"  # as when it was measured
        if os.path.exists(tpath) and world == 1 and full_size and not args.samples_per_pass:
            with open(tpath) as f:
                measured = json.load(f)["kernels"]
            # the uninstrumented instantiation(s) of the dominant kernel, e.g. "k_extend<false, 4u>", "k_shade<0u, false>"
            tk = [v for k, v in measured.items()
                  if k == dom["kernel"] or (k.startswith(dom["kernel"] + "<") and not k.startswith(dom["kernel"] + "<true"))]
            if tk:
                traffic = sum(v["hbm_total"] * v["launches"] for v in tk) / sum(v["launches"] for v in tk)
                traffic_src = os.path.relpath(tpath, ROOT)
        traversal_ms = (stage_ms["ms_extend"] + stage_ms["ms_shadow"]) / args.steps
        traversal_bytes = roofline.extend_bytes(cst) + roofline.shadow_bytes(cst)
        result = {
            "metric": "Msamples/s",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{int(args.config[1]) - 1}] ({args.config}): {W}x{H}, {spp} spp ({sx}x{sy} strata), "
                                   f"path depth {depth}, frame tiled over {world} GPU(s) in interleaved {tiling.BAND_ROWS}-row bands",
                       "scene": args.config, "width": W, "height": H, "spp": spp, "depth": depth, "seed": args.seed,
                       "integrator": args.integrator},
            "mrays_per_s": rays_per_step * args.steps / elapsed / 1e6,
            "rays_per_step": rays_per_step,
            "frame_mean_radiance": [float(x) for x in frame.reshape(-1, 3).mean(axis=0)],
            "roofline": {
                "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBps"], "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["achieved_GBps"] / roofline.HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": dom["bytes_per_launch"], "ms_per_launch": dom["ms_per_launch"], "launches_per_step": dom["launches"],
                "note": "rank 0's kernels; achieved = algorithmic bytes (SURVEY.md §8d) / HIP-event time; traffic = HBM bytes per launch from separate "
                        "rocprofv3 FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled per the gfx950 guide), measured offline, see profiles/",
            },
            "traversal": {"achieved": traversal_bytes / (traversal_ms * 1e-3) / 1e9 if traversal_ms > 0 else 0.0, "unit": "GB/s",
                          "frac": (traversal_bytes / (traversal_ms * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS) if traversal_ms > 0 else 0.0,
                          "kernels": "k_extend + k_shadow"},
            "stages_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "stages": rep,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = tuple(args.cpu_sample) if args.cpu_sample else (384, 384, 4)
            result["cpu_baseline"] = cpu_baseline(args.config, depth, args.seed, sample, args.integrator)
        print(json.dumps(result))
    if shared is not None:
        shared.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
