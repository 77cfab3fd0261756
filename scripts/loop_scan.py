"""Loops of one disassembled kernel (llvm-objdump -d --disassemble-symbols=...): line range, MFMA count, scratch / SGPR-spill instructions inside.
usage: python scripts/loop_scan.py build/dis/kernel.s [min_lines]"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
addr = {}
for i, l in enumerate(L):
    m = re.search(r'// ([0-9A-F]{12}):', l)
    if m: addr[int(m.group(1), 16)] = i
for i, l in enumerate(L):
    m = re.match(r'\s+(s_cbranch_\w+|s_branch) (\d+)\s+// ([0-9A-F]{12}):', l)
    if not m: continue
    off = int(m.group(2))
    if off >= 32768: off -= 65536
    t = addr.get(int(m.group(3), 16) + 4 + off * 4)
    if t is not None and t < i and i - t >= minl:
        body = L[t:i]
        print(f"loop {t + 1}-{i + 1}: mfma {sum('mfma' in x for x in body)}  scratch {sum('scratch_' in x for x in body)}  readlane/writelane {sum('v_readlane' in x or 'v_writelane' in x for x in body)}"
              f"  barriers {sum('s_barrier' in x for x in body)}  global_load {sum('global_load' in x for x in body)}")
