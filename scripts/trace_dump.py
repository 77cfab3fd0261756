"""Kernel durations of the LAST full-size segmentation pass in a rocprofv3 kernel trace, in launch order: python scripts/trace_dump.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def _first(name):                                       # conv3_igemm_sres<MREP, RX, RY, WY, WX, RING, FIRST, BLDS, M16> with FIRST = true
    if "conv3_igemm_sres<" not in name: return False
    a = [x.strip() for x in name.split("conv3_igemm_sres<", 1)[1].split(">", 1)[0].split(",")]
    return len(a) > 6 and a[6] == "true"
starts = [i for i, r in enumerate(rows) if _first(r["Kernel_Name"])]
rows = rows[starts[-1]:]
tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print(f"{r['Kernel_Name'][:100]:100s} {d:9.1f} us")
print(f"sum {tot / 1e3:.2f} ms")
