"""Kernel durations of the LAST full-size segmentation pass in a rocprofv3 kernel trace, in launch order: python scripts/trace_dump.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "conv3_igemm_sres<" in r["Kernel_Name"] and "false, true, false" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print(f"{r['Kernel_Name'][:100]:100s} {d:9.1f} us")
print(f"sum {tot / 1e3:.2f} ms")
