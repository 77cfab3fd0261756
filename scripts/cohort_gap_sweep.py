"""Parent of scripts/cohort_gap.py queues: one child per GPU_MAX_HW_QUEUES setting (the variable is read at the runtime's initialisation, so it is set in
the child's environment before its first GPU call; this process never touches the GPU)."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
for q in tuple(os.environ.get("QUEUES", "None,8,16,2").replace("None", "").split(",")):
    env = dict(os.environ)
    if not q:
        env.pop("GPU_MAX_HW_QUEUES", None)
    else:
        env["GPU_MAX_HW_QUEUES"] = q
    r = subprocess.run([sys.executable, os.path.join(here, "cohort_gap.py"), "queues"], env=env, capture_output=True, text=True, timeout=600)
    print("\n".join(ln for ln in r.stdout.splitlines() if "amdgpu.ids" not in ln), flush=True)
    if r.returncode:
        print("rc", r.returncode, r.stderr[-600:], flush=True)
