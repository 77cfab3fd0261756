"""Scan the gfx950 code objects of liboai_hip.so for the hazard "VALU writes an SGPR -> VMEM reads that SGPR within 5 wait states".
The compiler's hazard recognizer pads its own instructions with s_nop; it cannot see inside inline assembly, so a `global_load ... s[b:b+1]`
issued from an asm statement right behind a v_readfirstlane / v_readlane (an SGPR spill reload, a uniform base) reads a stale SGPR: a wild
address (found the hard way: conv3_wino_sres faulted on some layers and not on others).  Prints every candidate; exit status 1 if any."""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
VALU_SGPR = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
VALU_CMP = re.compile(r"^\s*v_cmp\w*_e64\s+s\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)\w+\s+(.*)$")
SREG = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")


def scan(text):
    hits = []
    lines = [l.split("//")[0].rstrip() for l in text.split("\n")]
    kernel = "?"
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m: kernel = m.group(1); continue
        w = VALU_SGPR.match(l)
        regs = None
        if w: regs = {int(w.group(2))}
        else:
            c = VALU_CMP.match(l)
            if c: regs = set(range(int(c.group(1)), int(c.group(2)) + 1))
        if not regs: continue
        ws = 0
        for j in range(i + 1, min(i + 12, len(lines))):
            t = lines[j].strip()
            if not t or t.endswith(":"): continue
            v = VMEM.match(t)
            if v:
                used = set()
                for a, b, c in SREG.findall(v.group(2)):
                    used |= set(range(int(a), int(b) + 1)) if a else {int(c)}
                if used & regs and ws < 5:
                    hits.append((kernel, l.strip(), t, ws))
                    break
            n = re.match(r"s_nop\s+(\d+)", t)
            ws += int(n.group(1)) + 1 if n else 1
            if ws >= 5: break
    return hits


def scan_library(lib):
    work = tempfile.mkdtemp()
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    out = []
    for co in sorted(glob.glob(os.path.join(work, "lib.so*gfx950*"))):
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
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
