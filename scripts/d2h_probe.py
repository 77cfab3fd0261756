"""Why is the D2H of a volume's results a copy KERNEL on the compute units (profiles/r05_cohort.md) and not an SDMA transfer?  (VERDICT r5 #4a)

Parent (no GPU call of its own): for each runtime setting -- environment variables that must be set BEFORE the process's first GPU call -- starts
    rocprofv3 --kernel-trace --memory-copy-trace -- python3 scripts/d2h_probe.py child
with that environment and reads the traces: which engine moved the bytes (a `__amd_rocclr_copyBuffer*` kernel in the kernel trace = blit on the CUs;
a DEVICE_TO_HOST row in the memory-copy trace = the runtime's async-copy path, SDMA when enabled), how long the copy takes alone, and what it costs a
kernel stream that runs at the same time.  Reference loop this serves: dask_processing.py:170-181 (results come back to the host per volume).

Child: 566 MB in five device tensors -> pinned host tensors with tensor.copy_(non_blocking=True) on a side stream (what cohort.CohortRunner does),
(a) alone, (b) underneath a stream of matmul kernels that fill the CUs; plus one hipMemcpyAsync straight through the HIP runtime via ctypes
(no torch in between) to rule the framework out.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(160, 384, 384), (160, 384, 384), (3, 80, 192, 192), (160, 384, 384), (160, 384, 384)]       # fc, tc, phi, fc_atlas, tc_atlas


def child():
    import ctypes
    import torch
    torch.cuda.set_device(0)
    dev = [torch.rand(s, device="cuda") for s in SHAPES]
    pin = [torch.empty(s).pin_memory() for s in SHAPES]
    nbytes = sum(t.numel() * 4 for t in dev)
    side = torch.cuda.Stream()
    a = torch.rand(8192, 8192, device="cuda", dtype=torch.float16)
    b = torch.rand(8192, 8192, device="cuda", dtype=torch.float16)

    def copy_all(stream):
        with torch.cuda.stream(stream):
            for p, d in zip(pin, dev):
                p.copy_(d, non_blocking=True)

    def timed_copy(stream, with_compute: bool):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if with_compute:
            c0.record()
            for _ in range(40):
                a @ b                                              # ~40 x 0.6 ms of CU-filling kernels on the default stream
            c1.record()
        with torch.cuda.stream(stream):
            e0.record(stream)
        copy_all(stream)
        with torch.cuda.stream(stream):
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), (c0.elapsed_time(c1) if with_compute else None)

    for _ in range(2):
        timed_copy(side, False)
    alone = [timed_copy(side, False)[0] for _ in range(5)]
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(40):
        a @ b
    c1.record()
    torch.cuda.synchronize()
    compute_alone = c0.elapsed_time(c1)
    both = [timed_copy(side, True) for _ in range(5)]
    ok = all(torch.equal(p, d.cpu()) for p, d in zip(pin, dev))
    # the HIP runtime directly: hipMemcpyAsync(pinned, device, DeviceToHost) on a stream of its own
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    hip.hipStreamCreate(ctypes.byref(st))
    t0 = time.perf_counter()
    for p, d in zip(pin, dev):
        hip.hipMemcpyAsync(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(d.numel() * 4), 2, st)      # 2 = hipMemcpyDeviceToHost
    hip.hipStreamSynchronize(st)
    raw_ms = 1e3 * (time.perf_counter() - t0)
    print("D2HPROBE " + json.dumps({"bytes": nbytes, "alone_ms": alone, "GBps_alone": nbytes / (min(alone) * 1e-3) / 1e9, "compute_alone_ms": compute_alone,
                                    "under_compute": [{"copy_ms": c, "compute_ms": k} for c, k in both], "raw_hipMemcpyAsync_ms_host_clock": raw_ms, "bytes_ok": ok}), flush=True)


SETTINGS = [
    ("default", {}),
    ("HSA_ENABLE_SDMA=1", {"HSA_ENABLE_SDMA": "1"}),
    ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}),
    ("GPU_FORCE_BLIT_COPY_SIZE=0", {"GPU_FORCE_BLIT_COPY_SIZE": "0"}),
    ("GPU_BLIT_ENGINE_TYPE=2", {"GPU_BLIT_ENGINE_TYPE": "2"}),
    ("DEBUG_CLR_LIMIT_BLIT_WG=16", {"DEBUG_CLR_LIMIT_BLIT_WG": "16"}),
    ("HSA_ENABLE_SDMA=1 + GPU_FORCE_BLIT_COPY_SIZE=0", {"HSA_ENABLE_SDMA": "1", "GPU_FORCE_BLIT_COPY_SIZE": "0"}),
]


def read_traces(d):
    import csv
    import glob
    out = {"copy_kernels": {}, "memcpy_rows": {}}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "copyBuffer" in n or "Blit" in n or "blit" in n or "fillBuffer" in n:
                k = out["copy_kernels"].setdefault(n[:60], {"n": 0, "ms": 0.0})
                k["n"] += 1
                k["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = r.get("Direction", r.get("Name", "?"))
            k = out["memcpy_rows"].setdefault(key, {"n": 0, "ms": 0.0})
            k["n"] += 1
            k["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return out


def parent():
    out_root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "d2h_probe")
    os.makedirs(out_root, exist_ok=True)
    results = []
    for i, (name, env_add) in enumerate(SETTINGS):
        env = dict(os.environ)
        env.update(env_add)
        env["TMPDIR"] = "/tmp"
        d = os.path.join(out_root, f"s{i}")
        cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "-d", d, "-o", "t", "--output-format", "csv", "--",
               "python3", os.path.join(ROOT, "scripts", "d2h_probe.py"), "child"]
        try:
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("D2HPROBE ")]
            res = {"setting": name, "rc": r.returncode, "probe": json.loads(line[-1][9:]) if line else None, "traces": read_traces(d)}
            if not line:
                res["stderr_tail"] = r.stderr[-400:]
        except subprocess.TimeoutExpired:
            res = {"setting": name, "rc": "timeout"}
        results.append(res)
        print(json.dumps(res), flush=True)
    with open(os.path.join(out_root, "summary.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    child() if sys.argv[1:] == ["child"] else parent()
