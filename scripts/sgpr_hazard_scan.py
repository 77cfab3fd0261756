"""Scan the gfx950 code objects of liboai_hip.so for the hazard "VALU writes an SGPR -> VMEM reads that SGPR within 5 wait states".
The compiler's hazard recognizer pads its own instructions with s_nop; it cannot see inside inline assembly, so a `global_load ... s[b:b+1]`
issued from an asm statement right behind a v_readfirstlane / v_readlane (an SGPR spill reload, a uniform base) reads a stale SGPR: a wild
address (found the hard way: conv3_wino_sres faulted on some layers and not on others).  Prints every candidate; exit status 1 if any.
Round 4: the scan follows control flow -- a branch met inside the five-wait-state window is followed to its target (and, when
conditional, also fallen through), so a v_readlane at a loop tail in front of an asm load at the loop head is seen."""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.path.dirname(shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump")
VALU_SGPR = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
VALU_CMP = re.compile(r"^\s*v_cmp\w*_e64\s+s\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)\w+\s+(.*)$")
SREG = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")


BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\b")
ADDR = re.compile(r"//\s*([0-9A-Fa-f]{6,16}):")
TARGET = re.compile(r"<[^>+]+\+0x([0-9a-fA-F]+)>\s*$")


def scan(text):
    hits = []
    raw = text.split("\n")
    lines = [l.split("//")[0].rstrip() for l in raw]
    # address of every instruction line and the start address of its kernel: a branch's target is printed as <kernel+0xOFFSET>
    addr_of, index_at, kstart, kernel_of = {}, {}, {}, {}
    kernel, kbase = "?", None
    for i, l in enumerate(raw):
        m = re.match(r"^([0-9a-f]+) <(.+)>:", l)
        if m:
            kernel, kbase = m.group(2), int(m.group(1), 16)
            continue
        a = ADDR.search(l)
        if a:
            addr_of[i] = int(a.group(1), 16)
            index_at[addr_of[i]] = i
        kernel_of[i] = kernel
        kstart[i] = kbase

    def walk(i0, regs, ws0, origin, seen, depth):
        """instructions from line i0 on, `ws0` wait states behind the SGPR write at `origin`"""
        ws = ws0
        for j in range(i0, min(i0 + 16, len(lines))):
            t = lines[j].strip()
            if not t or t.endswith(":") or re.match(r"^[0-9a-f]+ <", t): continue
            if (j, ws) in seen: return
            seen.add((j, ws))
            v = VMEM.match(t)
            if v:
                used = set()
                for a, b, c in SREG.findall(v.group(2)):
                    used |= set(range(int(a), int(b) + 1)) if a else {int(c)}
                if used & regs and ws < 5:
                    hits.append((kernel_of.get(origin, "?"), lines[origin].strip(), t, ws))
                    return
            br = BRANCH.match(t)
            n = re.match(r"s_nop\s+(\d+)", t)
            ws += int(n.group(1)) + 1 if n else 1
            if br and depth < 3:
                tg = TARGET.search(raw[j])
                if tg and kstart.get(j) is not None:
                    k = index_at.get(kstart[j] + int(tg.group(1), 16))
                    if k is not None and ws < 5: walk(k, regs, ws, origin, seen, depth + 1)
                if br.group(1) == "s_branch": return                   # unconditional: no fall-through
            if ws >= 5: return

    for i, l in enumerate(lines):
        w = VALU_SGPR.match(l)
        regs = None
        if w: regs = {int(w.group(2))}
        else:
            c = VALU_CMP.match(l)
            if c: regs = set(range(int(c.group(1)), int(c.group(2)) + 1))
        if regs: walk(i + 1, regs, 0, i, set(), 0)
    return hits


def scan_library(lib):
    work = tempfile.mkdtemp()
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    out = []
    for co in sorted(glob.glob(os.path.join(work, "lib.so*gfx950*"))):
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout      # (with the raw encoding: the comment carries the address)
        for k, a, b, ws in scan(txt):
            out.append(f"{subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:90]}: `{a}` -> `{b}` after {ws} wait state(s)")
    shutil.rmtree(work)
    return out


def main():
    hits = scan_library(os.environ.get("OAI_LIB_PATH") or os.path.join(ROOT, "oai_analysis_2_amd", "liboai_hip.so"))
    print("\n".join(hits + [f"{len(hits)} candidate(s)"]))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
