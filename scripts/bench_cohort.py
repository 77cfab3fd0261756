"""BASELINE config 4: fused segment -> register -> resample per volume, N synthetic 384x384x160 volumes streamed through one MI355X from
HOST memory -- which part of the streaming costs what (round 5): resident loop (pipe.run on one device tensor), results left on the device
(upload only), results copied back (upload + D2H + host copy on worker threads).  Steady state = the inter-result interval away from fill / drain."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.cohort import CohortRunner
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

N = int(os.environ.get("N", "16"))
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
atlas = Image(make_volume(1000, shape), **meta)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
icon = IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192))
pipe = VolumePipeline(unet, icon, atlas)
base = [make_volume(i, shape) for i in range(4)]
vols = [Image(base[i % 4], **meta) for i in range(N)]
dev = torch.from_numpy(base[0]).cuda()
for _ in range(2): pipe.run(dev, vols[0], check=False)
torch.cuda.synchronize(); t = time.time()
for i in range(N): r = pipe.run(dev, vols[0], check=False)
torch.cuda.synchronize(); res_ms = (time.time() - t) / N * 1e3
print(f"resident loop: {res_ms:.2f} ms per volume")
for keep in (True, False):
    runner = CohortRunner(pipe, keep_on_device=keep)
    list(runner.run(vols[:3]))                       # warm-up (workspace allocation, first-use costs)
    torch.cuda.synchronize(); t = time.time(); st = []
    for _ in runner.run(vols): st.append(time.time() - t)   # results are consumed (not accumulated) as a cohort driver would
    torch.cuda.synchronize(); dt = time.time() - t
    steady = (st[N - 3] - st[3]) / (N - 6) * 1e3
    print(f"{N} volumes from host memory, results {'left on the device' if keep else 'copied back (5 tensors, 0.6 GB per volume)'}: "
          f"{dt:.3f} s -> {N / dt:.2f} volumes/s; steady state {steady:.2f} ms per volume = {res_ms / steady:.3f} of resident; stats {runner.stats}")
    runner.close()
# ---- what the D2H alone costs (no host copy, no upload): the resident loop with the five result tensors of every volume copied to pinned memory on a side stream
side = torch.cuda.Stream()
pins = None
def loop(d2h):
    global pins
    torch.cuda.synchronize(); t = time.time()
    for i in range(N):
        r = pipe.run(dev, vols[0], check=False)
        if d2h:
            ev = torch.cuda.Event(); ev.record()
            side.wait_event(ev)
            ts = (r.fc, r.tc, r.phi, r.fc_atlas, r.tc_atlas)
            if pins is None: pins = [torch.empty(x.shape, dtype=x.dtype).pin_memory() for x in ts]
            with torch.cuda.stream(side):
                for p, x in zip(pins, ts): p.copy_(x, non_blocking=True)
            keep.append(r)
            if len(keep) > 2: keep.pop(0)
    torch.cuda.synchronize(); return (time.time() - t) / N * 1e3
keep = []
loop(True)
a, b, c = loop(False), loop(True), loop(False)
print(f"resident loop {a:.2f} / {c:.2f} ms per volume; with the D2H of its results on a side stream (no host copy, no upload): {b:.2f} ms")
# ---- ... and with the host copy out of the pinned buffers (on THIS thread, one volume behind): does the 566 MB host memcpy slow the GPU's kernels down?
import concurrent.futures as cf
pool = cf.ThreadPoolExecutor(5)
def loop2(parallel):
    evs = []
    torch.cuda.synchronize(); t = time.time()
    for i in range(N):
        r = pipe.run(dev, vols[0], check=False)
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        ts = (r.fc, r.tc, r.phi, r.fc_atlas, r.tc_atlas)
        with torch.cuda.stream(side):
            for p, x in zip(pins, ts): p.copy_(x, non_blocking=True)
            e2 = torch.cuda.Event(); e2.record(side)
        keep.append(r)
        if len(keep) > 2: keep.pop(0)
        evs.append(e2)
        if len(evs) > 1:
            evs[-2].synchronize()
            if parallel: outs = list(pool.map(lambda p: torch.empty(p.shape, dtype=p.dtype).copy_(p), pins))
            else: outs = [p.clone() for p in pins]
    torch.cuda.synchronize(); return (time.time() - t) / N * 1e3
print(f"... + host copy of the previous volume's results on the launch thread: serial clone {loop2(False):.2f} ms, five threads {loop2(True):.2f} ms per volume; resident again {loop(False):.2f}")
torch.set_num_threads(1)
print(f"... the same with torch.set_num_threads(1): serial clone {loop2(False):.2f} ms, five threads {loop2(True):.2f} ms per volume")
# ---- where the runner's extra time goes: GPU-side stamps around every pipe.run of a download-mode run (duration of a volume's work, gap to the next)
orig = pipe.run
stamps = []
def timed_run(v, m, check=True):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); r = orig(v, m, check=check); b.record(); stamps.append((a, b)); return r
pipe.run = timed_run
torch.set_num_threads(os.cpu_count() or 8)
for keep_ in (True, False):
    stamps.clear()
    runner = CohortRunner(pipe, keep_on_device=keep_)
    list(runner.run(vols[:3])); stamps.clear()
    for _ in runner.run(vols): pass
    torch.cuda.synchronize()
    dur = [a.elapsed_time(b) for a, b in stamps]
    gap = [stamps[i][1].elapsed_time(stamps[i + 1][0]) for i in range(len(stamps) - 1)]
    print(f"keep_on_device={keep_}: volume work on the GPU {sum(dur[3:-3]) / len(dur[3:-3]):.2f} ms (min {min(dur):.2f} max {max(dur):.2f}); gap to the next volume {sum(gap[3:-3]) / len(gap[3:-3]):.3f} ms (max {max(gap):.3f})")
    runner.close()
# ---- how long the D2H of one volume's results takes UNDER the next volume's kernels (side-stream events), and alone
def d2h_times(with_compute):
    out = []
    r = r0
    for i in range(6):
        if with_compute: r = pipe.run(dev, vols[0], check=False)
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        if with_compute: r2 = pipe.run(dev, vols[0], check=False)      # the work it runs underneath
        with torch.cuda.stream(side):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(side)
            for p, x in zip(pins, (r.fc, r.tc, r.phi, r.fc_atlas, r.tc_atlas)): p.copy_(x, non_blocking=True)
            b.record(side)
        torch.cuda.synchronize(); out.append(a.elapsed_time(b))
    return out
r0 = pipe.run(dev, vols[0], check=False); torch.cuda.synchronize()
print("D2H of 566 MB alone (ms):", [f"{t:.1f}" for t in d2h_times(False)], " underneath a volume's kernels:", [f"{t:.1f}" for t in d2h_times(True)])
# ---- the D2H runs as a copy KERNEL (__amd_rocclr_copyBuffer in the kernel trace) on the CUs, underneath the next volume's first kernels: does it yield to
# compute queued on a HIGH-priority stream?
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range())
hi = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
pipe.run = orig
for tag, ctx in (("default-priority compute stream", torch.cuda.stream(torch.cuda.current_stream())), ("high-priority compute stream", torch.cuda.stream(hi))):
    with ctx:
        runner = CohortRunner(pipe, keep_on_device=False)
        list(runner.run(vols[:4]))
        torch.cuda.synchronize(); t = time.time(); st = []
        for _ in runner.run(vols): st.append(time.time() - t)
        torch.cuda.synchronize()
        steady = (st[N - 3] - st[3]) / (N - 6) * 1e3
        print(f"{tag}: steady state {steady:.2f} ms per volume")
        runner.close()
