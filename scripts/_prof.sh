cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ic -o ic --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_icon.py > /tmp/ic.log 2>&1
tail -1 /tmp/ic.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ic/**/ic_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last direction: find last copy_f32 triple... just take the last 75 kernels
last = rows[-72:]
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{d:8.1f} us  {r['Kernel_Name'][:90]}  grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))}")
print('sum', tot)
PY
