"""Which path does the HIP runtime take for a pinned D2H?  Runs the copy in the situations the cohort runner creates and marks them in the runtime's own
log (AMD_LOG_LEVEL=4 python3 scripts/d2h_log.py 2> log): `HSA Copy copy_engine=...` = hsa_amd_memory_async_copy on an SDMA engine; a dispatch of
`__amd_rocclr_copyBuffer` = a blit kernel on the compute units.  (VERDICT r5 #4a; reference loop dask_processing.py:170-181)"""
import ctypes
import sys
import torch
torch.cuda.set_device(0)
d = torch.rand(64, 384, 384, device="cuda")
p = torch.empty(d.shape).pin_memory()
a = torch.rand(4096, 4096, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")


def mark(s):
    torch.cuda.synchronize()
    sys.stderr.write(f"=== {s}\n")
    sys.stderr.flush()


st = ctypes.c_void_p()
hip.hipStreamCreate(ctypes.byref(st))
mark("case 1: raw hipMemcpyAsync D2H on an idle stream of its own")
hip.hipMemcpyAsync(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(d.numel() * 4), 2, st)
hip.hipStreamSynchronize(st)
mark("case 2: torch copy_(non_blocking) on an idle side stream")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    p.copy_(d, non_blocking=True)
mark("case 3: the cohort runner's pattern -- kernels on the compute stream, an event, the side stream waits for it, then the copy")
for _ in range(4):
    a @ a
done = torch.cuda.Event()
done.record()
side.wait_event(done)
with torch.cuda.stream(side):
    p.copy_(d, non_blocking=True)
mark("case 4: the copy on the SAME stream as the kernels, right behind them")
for _ in range(4):
    a @ a
p.copy_(d, non_blocking=True)
mark("case 5: kernels still RUNNING on the compute stream while an independent side stream copies")
for _ in range(40):
    a @ a
with torch.cuda.stream(side):
    p.copy_(d, non_blocking=True)
mark("end")
