"""Where do the 2-4 ms between two streamed volumes come from (profiles/r05_cohort.md: steady state 0.93-0.99 of the resident rate, bi-stable)?  Round 6 found that
the runtime's D2H is an SDMA transfer in production -- the `__amd_rocclr_copyBuffer` kernel of round 5's traces only exists under rocprofv3 (scripts/d2h_log.py) --
so the copy kernel is not the cause.  This script runs CohortRunner (download mode, 24 volumes) with one thing changed at a time and reports the steady-state
interval and the GPU-side gap between consecutive volumes' kernels:
    default | torch.set_num_threads(1) (no OpenMP fan-out inside the host copy_) | one clone thread | no host copy at all (results = the pinned buffers) |
    sys.setswitchinterval(0.2 ms) (the GIL hand-over between the workers and the launch thread) | LAG 3
Reference loop this replaces: dask_processing.py:170-181."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import cohort
from oai_analysis_2_amd.cohort import CohortRunner
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline, VolumeResult
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

N = int(os.environ.get("N", "24"))
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
atlas = Image(make_volume(1000, shape), **meta)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
icon = IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192))
pipe = VolumePipeline(unet, icon, atlas)
base = [make_volume(i, shape) for i in range(4)]
vols = [Image(base[i % 4], **meta) for i in range(N)]
dev = torch.from_numpy(base[0]).cuda()
orig_run = pipe.run
stamps = []


def timed_run(v, m, check=True):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig_run(v, m, check=check)
    b.record()
    stamps.append((a, b))
    return r


def resident():
    for _ in range(2):
        orig_run(dev, vols[0], check=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        orig_run(dev, vols[0], check=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N * 1e3


def streamed(tag, make_runner=lambda: CohortRunner(pipe)):
    pipe.run = timed_run
    runner = make_runner()
    list(runner.run(vols[:3]))
    torch.cuda.synchronize()
    stamps.clear()
    t = time.perf_counter()
    st = []
    for _ in runner.run(vols):
        st.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    steady = (st[N - 3] - st[3]) / (N - 6) * 1e3
    dur = [a.elapsed_time(b) for a, b in stamps]
    gap = [stamps[i][1].elapsed_time(stamps[i + 1][0]) for i in range(len(stamps) - 1)]
    s = runner.stats
    print(f"{tag:58s} steady {steady:7.2f} ms/volume  GPU work {sum(dur[3:-3]) / len(dur[3:-3]):7.2f}  gap mean {sum(gap[3:-3]) / len(gap[3:-3]):6.3f} max {max(gap[3:-3]):6.3f} ms  "
          f"launch thread: upload wait {s['t_upload_wait']:.3f} queue {s['t_queue_compute']:.3f} d2h issue {s['t_issue_d2h']:.3f} result wait {s['t_result_wait']:.3f} s", flush=True)
    runner.close()
    pipe.run = orig_run
    return steady


if len(sys.argv) > 1 and sys.argv[1] == "queues":
    # child of the sweep below: five runner instances (each creates its own copy / download streams: the HIP runtime deals streams onto its hardware queues
    # round-robin, GPU_MAX_HW_QUEUES of them) under whatever GPU_MAX_HW_QUEUES the parent set before this process's first GPU call
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default 4)')}: resident loop {resident():.2f} ms per volume", flush=True)
    for i in range(6):
        streamed(f"runner instance {i}")
    streamed("keep_on_device (upload only)", lambda: CohortRunner(pipe, keep_on_device=True))
    print(f"resident loop {resident():.2f} ms per volume", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    raise SystemExit("the sweep is a parent without a GPU context: run scripts/cohort_gap_sweep.py")
print(f"resident loop {resident():.2f} ms per volume; host threads: torch {torch.get_num_threads()}, cores {os.cpu_count()}", flush=True)
for rep in range(2):
    streamed("default")
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)
    streamed("torch.set_num_threads(1)")
    torch.set_num_threads(n0)

    def one_clone():
        r = CohortRunner(pipe)
        r._clone.shutdown()
        from concurrent.futures import ThreadPoolExecutor
        r._clone = ThreadPoolExecutor(1)
        return r
    streamed("one clone thread", one_clone)

    def no_host_copy():
        r = CohortRunner(pipe)

        def collect(k, ev, keys, repeated, held=None):
            try:
                ev.synchronize()
                pins = r._pin_out[k]
                if len(keys) > 5 and int(pins[keys[5]][0]):
                    return None
                return VolumeResult(*[pins[key] for key in keys[:5]], repeated_f32=repeated)
            finally:
                r._free_out.put(k)
        r._collect = collect
        return r
    streamed("no host copy (results = the pinned buffers)", no_host_copy)
    old = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    streamed("sys.setswitchinterval(0.2 ms)")
    sys.setswitchinterval(old)
    old_lag = CohortRunner.LAG, CohortRunner.N_OUT_SETS
    CohortRunner.LAG, CohortRunner.N_OUT_SETS = 3, 4
    streamed("LAG 3")
    CohortRunner.LAG, CohortRunner.N_OUT_SETS = old_lag
    streamed("keep_on_device (upload only)", lambda: CohortRunner(pipe, keep_on_device=True))
print(f"resident loop {resident():.2f} ms per volume", flush=True)
