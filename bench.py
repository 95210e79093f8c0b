#!/usr/bin/env python3
"""bench.py — Msamples/s (+ Mrays/s) of the MI355X wavefront path tracer on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W          (N = 1: plain process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The timed workload (`value`, `config.workload`) is the one BASELINE.json's north star quotes its targets on: C4 =
configs[3], the procedural 1 048 576-triangle mesh, 1920x1080, 512 spp, path depth 8, at full size — it fits one GPU,
and it is the same workload for every N so that the N = 1 value of a scaling run agrees with the single-GPU bench.
A step is one whole frame: every rank renders its interleaved row bands of the frame on its own GPU (scene already
resident in HBM), copies them to the host and rank 0 assembles the frame (shared-memory frame, or a gloo gather; no RCCL:
pixels are independent).  Total work is fixed as N grows, so scaling is reported as "strong".
At N = 1 the same run also times C2 and C3 at full size and C5 on a stated 64-spp slice (fewer steps) and reports them under
`other_configs`.  After the timed loop — outside the timed region, like the `cpu_baseline` leg — the oracle renders a 64 x 8
window of the LAST TIMED FRAME at the workload's full strata and the line carries `parity_window` (bit-exact or not, max
per-pixel L2 error); a failed window makes the process exit non-zero after the line is printed.
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

DEFAULT_CONFIG = "c4"
CPU_SAMPLES = {"c1": (256, 256, 4), "c2": (512, 512, 4), "c3": (512, 512, 4), "c4": (960, 540, 4), "c5": (960, 540, 4), "c4xl": (960, 540, 4)}
# the window of the timed frame the oracle re-renders (x0, y0, w, h), chosen where each scene has its mixed materials / deep BLAS
PARITY_WINDOWS = {"c1": (96, 120, 64, 8), "c2": (480, 500, 64, 8), "c3": (300, 700, 64, 8), "c4": (900, 600, 64, 8), "c5": (1800, 1400, 64, 8),
                  "c4xl": (900, 600, 64, 8)}
# configs timed on a stated slice of their spp when they ride along in `other_configs` (samples are i.i.d. passes: SURVEY.md §8d)
ALSO_STRATA = {"c5": (8, 8)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=DEFAULT_CONFIG, help="c1..c5 (BASELINE.json configs[0..4]); default c4 = the north-star workload; c4xl = C4's generator at 16.8 M triangles (not a BASELINE config)")
    ap.add_argument("--also", default=None, help="comma-separated configs timed after the main one at N = 1 (default: c2,c3,c5 when --config is c4; c5 on a 64-spp slice)")
    ap.add_argument("--also-steps", type=int, default=2)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--strata", type=int, nargs=2, default=None, metavar=("SX", "SY"))
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--integrator", default="path", choices=["path", "direct"],
                    help="path = src/pathintegrator.rs (the BASELINE metric); direct = direct_lighting_integrator (src/directlighting.rs:14-47)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-window", action="store_true", help="skip the oracle's window of the timed frame (developer A/B runs)")
    ap.add_argument("--cpu-sample", type=int, nargs=3, default=None, metavar=("W", "H", "MSAA"), help="CPU baseline sample (default per config)")
    ap.add_argument("--allow-shared-gpus", action="store_true",
                    help="rehearsal only: let more ranks than visible GPUs share devices round-robin (the JSON then says so)")
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


def parity_window(config_name, frame, window, sx, sy, depth, seed, integrator="path"):
    """The oracle's render of `window` of the film at the workload's own strata against the same pixels of `frame` (a frame the
    GPU produced in the timed loop).  The oracle is the checker here, never the thing measured."""
    from oracle.binding import OracleScene
    from pbrs_amd import scenes
    H, W = frame.shape[:2]
    x0, y0, w, h = window
    x0, y0 = min(x0, max(W - w, 0)), min(y0, max(H - h, 0))
    w, h = min(w, W), min(h, H)
    sb, _ = scenes.build_config(config_name, width=W, height=H)
    t = time.perf_counter()
    ref, ost = OracleScene(sb).render(sx, sy, depth, seed, tile=(x0, y0, w, h), nthreads=host_cores(), integrator=integrator)
    dt = time.perf_counter() - t
    got = np.ascontiguousarray(frame[y0:y0 + h, x0:x0 + w])
    nan = np.isnan(ref)
    exact = bool((nan == np.isnan(got)).all() and (got.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all())
    l2 = np.sqrt(((np.nan_to_num(got).astype(np.float64) - np.nan_to_num(ref).astype(np.float64)) ** 2).sum(axis=-1))
    return {"rect": [x0, y0, w, h], "spp": sx * sy, "strata": [sx, sy], "samples": int(ost["samples"]), "bit_exact": exact,
            "max_l2": float(l2.max()), "tolerance_l2": 1e-4, "oracle_panics": int(ost["panics"]), "oracle_s": dt,
            "frame": "the last frame of the timed loop", "window_mean_radiance": [float(x) for x in ref.reshape(-1, 3).mean(axis=0)]}


def device_identity(torch, ordinal):
    p = torch.cuda.get_device_properties(ordinal)
    ident = {"ordinal": ordinal, "name": p.name}
    for k in ("pci_bus_id", "pci_device_id", "pci_domain_id", "uuid"):
        if hasattr(p, k):
            ident[k] = str(getattr(p, k))
    return ident


class Workload:
    """One BASELINE config resident on this rank's GPU, rendered frame by frame."""

    def __init__(self, args, config, ctx, torch, dist, dev, world, rank, strata=None):
        import pbrs_amd
        from pbrs_amd import scenes, tiling
        self.args, self.name, self.ctx, self.torch, self.dist, self.world, self.rank = args, config, ctx, torch, dist, world, rank
        self.tiling = tiling
        sb, cfg = scenes.build_config(config, width=args.width, height=args.height)
        if strata or args.strata:
            cfg["strata_x"], cfg["strata_y"] = strata or args.strata
        if args.depth:
            cfg["depth"] = args.depth
        self.W, self.H, self.sx, self.sy, self.depth = cfg["width"], cfg["height"], cfg["strata_x"], cfg["strata_y"], cfg["depth"]
        self.full_size = (self.W, self.H, self.sx, self.sy, self.depth) == tuple(scenes.CONFIGS[config][2:7]) and args.integrator == "path"
        self.hs = pbrs_amd.HostScene(sb)
        ctx.upload(self.hs)  # scene resident in HBM before any timed region
        self.my_rows = tiling.packed_height(self.H, world, rank)
        # two device buffers: a rank renders frame n + 1 while the rows of frame n travel to the host and into the frame
        self.out_dev = [torch.empty((max(self.my_rows, 1), self.W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        self.out_host = torch.empty((max(self.my_rows, 1), self.W, 3), dtype=torch.float32).pin_memory()
        self.bands = (tiling.BAND_ROWS, world, rank) if world > 1 else None
        # one node: the ranks write their rows into a frame in shared memory (no data through gloo); otherwise gloo gather
        self.shared = tiling.SharedFrame.create(self.W, self.H, world, rank) if world > 1 else None
        self.gather_s = 0.0

    def _launch(self, buf, timing, counters):
        """Queues one frame of this rank's rows on the context's stream (asynchronous)."""
        a = self.args
        if self.my_rows:
            self.ctx.render_device(self.out_dev[buf].data_ptr(), self.sx, self.sy, self.depth, a.seed, tile=(0, 0, self.W, self.my_rows),
                                   bands=self.bands, samples_per_pass=a.samples_per_pass, timing=timing, counters=counters, integrator=a.integrator)

    def _deliver(self, buf):
        """Rows of a finished frame: device -> pinned host -> the frame (shared memory, or gloo gather).  Rank 0 gets the frame."""
        t = time.perf_counter()
        self.out_host.copy_(self.out_dev[buf])  # torch's stream: does not wait for the next frame on the context's stream
        if self.shared is not None:
            frame = self.shared.publish(self.out_host.numpy()[:self.my_rows])
        else:
            frame = self.tiling.gather_frame(self.out_host.numpy()[:self.my_rows], self.W, self.H, self.world, self.rank)
        self.gather_s += time.perf_counter() - t
        return frame

    def frames(self, n, timing=False, counters=False):
        """n consecutive frames.  Frame k + 1 is queued on the GPU before frame k's rows are copied out and handed over, so
        the hand-over (D2H copy, row scatter, waiting for the slowest rank) hides behind rendering; the shared frame is
        double-buffered for exactly this one frame of lead.  Returns (last frame on rank 0, list of per-frame stats)."""
        stats, frame = [], None
        self._launch(0, timing, counters)
        for k in range(n):
            stats.append(self.ctx.collect_stats() if self.my_rows else None)  # waits for frame k on the context's stream
            if k + 1 < n:
                self._launch((k + 1) % 2, timing, counters)
            frame = self._deliver(k % 2)
        return frame, stats

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, steps, warmup):
        """W untimed frames, then exactly `steps` frames between barriers; returns rank 0's report (None elsewhere)."""
        from pbrs_amd import roofline
        torch, dist, world, rank = self.torch, self.dist, self.world, self.rank
        # the first frame allocates the path state (tens of GB): it is never a timed one, whatever --warmup says
        self.frames(max(warmup, 1))
        self.barrier()
        self.gather_s = 0.0
        t0 = time.perf_counter()
        stage_ms, launches = None, None
        timed_frame, per_frame = self.frames(steps, timing=True)

        for st in per_frame:
            if st is not None:
                if stage_ms is None:
                    stage_ms = {k: 0.0 for k in st if k.startswith("ms_")}
                    launches = {k: 0 for k in st if k.startswith("launches_")}
                for k in stage_ms:
                    stage_ms[k] += st[k]
                for k in launches:
                    launches[k] += st[k]
        own_wall = time.perf_counter() - t0
        self.barrier()
        elapsed = time.perf_counter() - t0
        if timed_frame is not None:
            timed_frame = np.array(timed_frame, copy=True)  # after the clock has stopped: the host buffers are reused by the frames that follow
        gather_ms = self.gather_s / steps * 1e3
        # what each rank did, so that a scaling curve explains itself: rows / bands it owns, its own wall time for the K frames
        # (its GPU work + hand-over, before the closing barrier) and the HIP-event time of its frames
        mine = {"rank": rank, "rows": int(self.my_rows), "bands": int(-(-self.my_rows // self.tiling.BAND_ROWS)),
                "gpu_ms_per_step": float(sum(st["ms_total"] for st in per_frame if st is not None) / steps),
                "own_wall_ms_per_step": float(own_wall / steps * 1e3), "handover_ms_per_step": float(gather_ms)}
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            t = torch.tensor([elapsed, gather_ms], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, gather_ms = float(t[0].item()), float(t[1].item())

        # Untimed: one frame with every pass on one stream, for per-stage milliseconds that are exclusive times (in the timed frames a pass's
        # late bounces run on a second stream beside the next pass's first bounces: their event brackets overlap and do not add up)
        self.ctx.set_pass_overlap(False)
        try:
            _, (sst,) = self.frames(1, timing=True)  # (every rank: the frame hand-over is collective)
        finally:
            self.ctx.set_pass_overlap(True)
        serial_times = {k: v for k, v in sst.items() if k.startswith("ms_")} if sst else None
        # Untimed: instrumented frame for ray / node / primitive counts (deterministic, equal to the timed work).
        frame, (cst,) = self.frames(1, counters=True)
        nb = 16
        counts = np.array(([cst["closest_rays"], cst["shadow_rays"], cst["samples"], cst["invalid_samples"], cst["shade_events"]] +
                           list(cst["paths_at_bounce"]) + list(cst["shadow_rays_at_bounce"])) if cst else [0] * (5 + 2 * nb), dtype=np.float64)
        if world > 1:
            tc = torch.from_numpy(counts)
            dist.all_reduce(tc, op=dist.ReduceOp.SUM)
            counts = tc.numpy()
        if rank != 0:
            return None
        W, H, sx, sy, depth, spp = self.W, self.H, self.sx, self.sy, self.depth, self.sx * self.sy
        # the instrumented frame is the same frame from other kernel instantiations: it must be the timed one bit for bit
        frames_agree = bool((np.isnan(frame) == np.isnan(timed_frame)).all() and
                            (frame.view(np.uint32)[~np.isnan(frame)] == timed_frame.view(np.uint32)[~np.isnan(timed_frame)]).all())

        samples_per_step = float(W) * H * spp
        assert counts[2] == samples_per_step, (counts, samples_per_step)
        rays_per_step = counts[0] + counts[1]
        times = {k: v / steps for k, v in stage_ms.items()}
        times.update({k: v // steps for k, v in launches.items()})
        # HBM bytes per launch cannot be collected in-process (rocprofv3 owns the counters): the last measured figures for
        # this workload are read from profiles/ (tools/round_artifacts.sh -> tools/traffic_from_pmc.py), full-size runs only.
        traffic_doc, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"latest_traffic_{self.name}.json")
        if os.path.exists(tpath) and world == 1 and self.args.integrator == "path" and not self.args.samples_per_pass:
            with open(tpath) as f:
                traffic_doc = json.load(f)
            traffic_src = os.path.relpath(tpath, ROOT)
            geo = traffic_doc.get("geometry", {})
            # per-launch figures only carry over when a launch is the same size: the same frame (pixels, spp — a full-size config, or
            # the stated slice C5 rides along on) cut into the same number of passes
            if (geo.get("pixels"), geo.get("spp")) != (W * H, spp):
                traffic_doc, traffic_src = None, f"{traffic_src} ignored: measured on {geo.get('pixels')} pixels x {geo.get('spp')} spp, this run has {W * H} x {spp}"
            elif geo.get("passes") not in (None, cst["passes"]):
                traffic_doc, traffic_src = None, f"{traffic_src} ignored: measured at {geo['passes']} passes per frame, this run has {cst['passes']}"
            # ... and when the counters were measured on the kernels being timed: the file carries a hash of the sources
            elif not roofline.counters_apply(traffic_doc)[0]:
                traffic_doc, traffic_src = None, (f"{traffic_src} ignored: measured on sources {traffic_doc.get('source_hash')} "
                                                  f"(git {traffic_doc.get('git_head')}), this run is built from {roofline.source_hash()}")
            elif traffic_doc.get("source_hash") != roofline.source_hash():
                traffic_src = f"{traffic_src} ({roofline.counters_apply(traffic_doc)[1]})"
        isa_doc = None
        ipath = os.path.join(ROOT, "profiles", "latest_isa_mix.json")
        if os.path.exists(ipath):
            with open(ipath) as f:
                isa_doc = json.load(f)
        rep = roofline.stage_report(cst, times, scene_nbytes=self.hs.nbytes, traffic_doc=traffic_doc, isa_doc=isa_doc, serial_times=serial_times)
        dom_name, dom = roofline.dominant(rep)
        trav = roofline.traversal(rep)
        # the queue/state bytes are a model: a fraction above 1 says the model is off, which the line reports instead of hiding
        inconsistent = [r["kernel"] if "kernel" in r else r["kernels"] for r in list(rep.values()) + [trav] if r["frac"] > 1.0]
        window = None
        if not self.args.no_parity_window:
            window = parity_window(self.name, timed_frame, PARITY_WINDOWS[self.name], sx, sy, depth, self.args.seed, self.args.integrator)
            window["timed_frame_equals_instrumented_frame"] = frames_agree
        n_b = [float(x) for x in counts[5:5 + nb]]
        s_b = [float(x) for x in counts[5 + nb:5 + 2 * nb]]
        last = max([i for i, x in enumerate(n_b) if x > 0] + [0])
        return {
            "value": samples_per_step * steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "steps": steps,
            "ms_per_step": elapsed / steps * 1e3,
            "config": {"workload": (f"BASELINE configs[{int(self.name[1]) - 1}] ({self.name})" if self.name in ("c1", "c2", "c3", "c4", "c5") else
                                    f"{self.name} (not a BASELINE config: C4's generator at 16.8 M triangles, a scene ten times the Infinity Cache)") +
                                   f": {W}x{H}, {spp} spp ({sx}x{sy} strata), "
                                   f"path depth {depth}, frame tiled over {world} GPU(s) in interleaved {self.tiling.BAND_ROWS}-row bands",
                       "scene": self.name, "width": W, "height": H, "spp": spp, "depth": depth, "seed": self.args.seed,
                       "integrator": self.args.integrator, "scene_bytes_in_hbm": self.hs.nbytes,
                       "full_size": self.full_size and self.name in ("c1", "c2", "c3", "c4", "c5")},
            "mrays_per_s": rays_per_step * steps / elapsed / 1e6,
            "rays_per_step": rays_per_step,
            "closest_rays_per_step": counts[0],
            "shadow_rays_per_step": counts[1],
            "shade_events_per_step": counts[4],
            # path vertices per camera sample (`scene.tlas.intersect` calls of src/pathintegrator.rs:16 per sample, misses included)
            "mean_path_length": counts[0] / samples_per_step,
            "rays_per_sample": rays_per_step / samples_per_step,
            "paths_at_bounce": n_b[:last + 1],          # k_extend's queue, bounce by bounce, summed over the frame's passes
            "shadow_rays_at_bounce": s_b[:last + 1],    # k_shadow's queue
            "invalid_samples": counts[3],
            "frame_mean_radiance": [float(x) for x in timed_frame.reshape(-1, 3).mean(axis=0)],
            "parity_window": window,
            "gather_ms": gather_ms,
            # which instantiations of the traversal kernels the timed frames ran (include/pbrs_gpu.h, pbrs_stats::kernel_features_*)
            "kernel_features": {"extend": int(per_frame[-1]["kernel_features_extend"]) if per_frame[-1] else None,
                                "shadow": int(per_frame[-1]["kernel_features_shadow"]) if per_frame[-1] else None,
                                "bits": "1 analytic shapes, 2 shading check, 4 scanned TLAS, 8 several node steps per round, 16 four-wide nodes, "
                                        "32 full further node steps (scene outside the guarded range of the division-free box test), 64 scene arrays in LDS, 128 TLAS in LDS, 256 (extend) the TLAS extent followed to the letter (a ParallelQuad next to a mesh)"},
            "per_rank": per_rank,
            "band_imbalance": max(r["rows"] for r in per_rank) / (sum(r["rows"] for r in per_rank) / len(per_rank)),
            "roofline_inconsistent": inconsistent or False,
            "roofline": {
                "bound": roofline.bound_of(dom), "kernel": dom["kernel"], "achieved": dom["achieved_GBps"], "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"], "frac_measured": dom["frac_measured"], "traffic": dom["traffic_bytes_per_launch"], "traffic_source": traffic_src,
                "bytes_per_launch": dom["queue_state_bytes_per_launch"] + dom["scene_miss_bytes_per_launch"],
                "queue_state_bytes_per_launch": dom["queue_state_bytes_per_launch"], "scene_miss_bytes_per_launch": dom["scene_miss_bytes_per_launch"],
                "scene_bytes_per_launch": dom["scene_bytes_per_launch"], "cache_work_rate_GBps": dom["cache_work_rate_GBps"],
                "ms_per_launch": dom["ms_per_launch"], "launches_per_step": dom["launches"],
                "ms_per_launch_serial": dom.get("ms_per_launch_serial"), "frac_serial": dom.get("frac_serial"),
                "bound_shares": roofline.bound_shares(dom),
                "valu_issue_frac": dom.get("valu_issue_frac"), "valu_issue_frac_min": dom.get("valu_issue_frac_min"),
                "valu_issue_upper_price": dom.get("valu_issue_upper_price"), "valu_32bit_encoding_share": dom.get("valu_32bit_encoding_share"),
                "salu_issue_frac": dom.get("salu_issue_frac"), "valu_lanes_active": dom.get("valu_lanes_active"),
                "l1_access_frac": dom.get("l1_access_frac"), "l1_accesses_per_launch": dom.get("l1_accesses_per_launch"),
                "ta_busy_share": dom.get("ta_busy_share"), "td_busy_share": dom.get("td_busy_share"),
                "note": "rank 0's kernels, the stage with the most exclusive time per frame (stages_ms_serial); ms_per_launch / achieved / frac are measured "
                        "over the timed frames, where a pass's late bounces share the chip with the next pass's first ones (stages_overlap); "
                        "ms_per_launch_serial / frac_serial are the same kernel with the overlap off, and the shares that divide offline per-launch "
                        "counters by a time (frac_measured, valu / salu_issue_frac, l1_access_frac) divide by that exclusive time; bound = the resource the kernel fills the largest share of "
                        "(bound_shares, every share <= 1): hbm (measured L2 -> fabric traffic where a counter file applies, else the model), valu_issue, "
                        "salu_issue, l1_access or td_busy (frac stays the HBM fraction); l1_access_frac = L1 accesses per launch (one per lane of "
                        "a load whose lanes name different lines, whatever its width: rocprofv3 TCP_TOTAL_CACHE_ACCESSES, offline) over 256 CUs x "
                        "one access per cycle x the live kernel time x 2.4 GHz; ta / td_busy_share = busy cycles of the texture address / data "
                        "units over the kernel's cycles (offline); achieved = (queue/state bytes that must "
                        "cross HBM: record sizes of the kernels' layout x units of one launch, pbrs_amd/roofline.py, + scene bytes that missed the "
                        "caches) / HIP-event time per launch; frac_measured = traffic / time / peak; scene misses and traffic = L2 -> fabric bytes "
                        "per launch from separate rocprofv3 TCC passes (Infinity-Cache hits included: an upper bound of HBM bytes), measured "
                        "offline on the same sources (source hash checked), see profiles/; "
                        "cache_work_rate prices every node / triangle visit at record size and is NOT an HBM figure; valu_issue_frac = "
                        "share of the chip's vector issue slots the kernel fills: wave-level VALU instructions per launch (SQ pass in profiles/) priced "
                        "at the kernel's static encoding mix (valu_32bit_encoding_share of them at the 2.7 cycles measured for 32-bit encodings, the "
                        "rest at 4.0: tools/isa_stats.py --json, tools/microbench/issue_rates.hip) over 1024 SIMDs x the live kernel time x 2.4 GHz; "
                        "valu_issue_frac_min = the same count with every instruction at 2.7 cycles, valu_issue_upper_price = with every instruction at 4.0 (an uncalibrated upper price kept for comparison with earlier rounds, NOT a share: it can pass 1); "
                        "salu_issue_frac = scalar instructions x 4.25 cycles over the same slots (they issue beside other waves' vector work)",
            },
            "traversal": trav,
            "stages_ms_per_step": times,
            "stages_overlap": "a render of several passes runs each pass's late bounces on a second stream beside the next pass's first bounces "
                              "(pbrs_set_pass_overlap): stages_ms_per_step are HIP-event brackets on the launching stream over the timed frames and "
                              "include the time a kernel shares the chip with the other stream's, so they do not add up to ms_total; "
                              "stages_ms_serial is one untimed frame with the overlap off (rank 0), where they do",
            "stages_ms_serial": serial_times,
            "stages": rep,
        }

    def close(self):
        if self.shared is not None:
            self.shared.close()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    import torch
    import torch.distributed as dist

    import pbrs_amd
    from pbrs_amd import scenes as scenes_mod

    gpus_visible = torch.cuda.device_count()
    if gpus_visible == 0:
        sys.exit("bench.py: no GPU visible; the product path has no CPU fallback")
    if local_world > gpus_visible and not args.allow_shared_gpus:
        # a scaling line must never claim more GPUs than it ran on
        sys.exit(f"bench.py: {local_world} ranks on this node but only {gpus_visible} GPU(s) visible; "
                 f"pass --allow-shared-gpus for a rehearsal (the JSON line then reports the sharing)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)  # host-side barrier/gather only
    ordinal = local_rank % gpus_visible
    torch.cuda.set_device(ordinal)
    dev = torch.device("cuda", ordinal)
    ident = device_identity(torch, ordinal)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    ctx = pbrs_amd.Context(ordinal)
    main_wl = Workload(args, args.config, ctx, torch, dist, dev, world, rank)
    result = main_wl.measure(args.steps, args.warmup)
    main_wl.close()

    others = {}
    also = args.also if args.also is not None else ("c2,c3,c5" if args.config == DEFAULT_CONFIG and not (args.width or args.height or args.strata or args.depth) else "")
    if world == 1:
        for name in [c for c in also.split(",") if c]:
            wl = Workload(args, name, ctx, torch, dist, dev, world, rank, strata=ALSO_STRATA.get(name))
            r = wl.measure(args.also_steps, 1)
            wl.close()
            others[name] = {k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "mrays_per_s", "mean_path_length", "config", "parity_window",
                                              "roofline", "traversal", "stages_ms_per_step", "stages_ms_serial", "kernel_features")}
            if name in ALSO_STRATA:
                full = scenes_mod.CONFIGS[name]
                others[name]["slice"] = {"spp_timed": ALSO_STRATA[name][0] * ALSO_STRATA[name][1], "spp_full": full[4] * full[5],
                                         "extrapolation": "linear in spp: samples are independent passes, Msamples/s carries over; a full frame takes "
                                                          f"{full[4] * full[5] // (ALSO_STRATA[name][0] * ALSO_STRATA[name][1])} x ms_per_step"}

    if rank == 0:
        distinct = {d.get("uuid") or d.get("pci_bus_id") or d["ordinal"] for d in idents}  # one node: ranks on one device share its identity
        line = {
            "metric": "Msamples/s",
            "value": result["value"],
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": result["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": result["config"],
            "gpus_visible": gpus_visible,
            "gpus_used": len(distinct),
            "shared_gpus": len(distinct) < world,
            "devices": idents,
        }
        for k in ("mrays_per_s", "rays_per_step", "closest_rays_per_step", "shadow_rays_per_step", "shade_events_per_step", "mean_path_length",
                  "rays_per_sample", "paths_at_bounce", "shadow_rays_at_bounce", "invalid_samples", "frame_mean_radiance", "parity_window", "gather_ms",
                  "per_rank", "band_imbalance", "kernel_features",
                  "roofline_inconsistent", "roofline", "traversal", "stages_ms_per_step", "stages_overlap", "stages_ms_serial", "stages"):
            line[k] = result[k]
        if others:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            sample = tuple(args.cpu_sample) if args.cpu_sample else CPU_SAMPLES[args.config]
            line["cpu_baseline"] = cpu_baseline(args.config, result["config"]["depth"], args.seed, sample, args.integrator)
        print(json.dumps(line), flush=True)
        windows = [("main", result["parity_window"])] + [(n, o["parity_window"]) for n, o in others.items()]
        failed = [n for n, w in windows if w is not None and not (w["bit_exact"] and w["timed_frame_equals_instrumented_frame"])]
    else:
        failed = []
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if failed:
        sys.exit(f"bench.py: the timed frame differs from the oracle in the parity window of: {', '.join(failed)}")


if __name__ == "__main__":
    main()
