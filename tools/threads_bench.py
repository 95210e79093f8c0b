"""Developer tool (GPU box): the in-process multi-GPU driver (pbrs_amd/threads.py: N host threads, N contexts, one frame buffer) on the
devices at hand — on a one-GPU box every context shares device 0, which rehearses the code path, not the speed-up.
    python tools/threads_bench.py [config [spp_x spp_y]]        prints one JSON line per thread count (1, 2, 4, 8)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pbrs_amd  # noqa: E402
from pbrs_amd import scenes, threads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
sb, cfg = scenes.build_config(name)
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 4)
hs = pbrs_amd.HostScene(sb)
n_dev = max(torch.cuda.device_count(), 1)
ref = None
for n in (1, 2, 4, 8):
    tf = threads.ThreadedFrame(hs, [k % n_dev for k in range(n)])
    tf.render(sx, sy, cfg["depth"], 1)  # allocates the path state
    t0 = time.perf_counter()
    frame, rep = tf.render(sx, sy, cfg["depth"], 1)
    dt = time.perf_counter() - t0
    tf.close()
    if ref is None:
        ref = frame
    same = bool((frame.view(np.uint32) == ref.view(np.uint32)).all())
    print(json.dumps({"config": name, "spp": sx * sy, "threads": n, "devices_visible": n_dev, "frame_ms": dt * 1e3, "frame_equals_one_thread_frame": same,
                      "band_imbalance": rep["band_imbalance"], "threads_report": rep["threads"]}), flush=True)
    assert same
