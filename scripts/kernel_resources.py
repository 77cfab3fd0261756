"""VGPR / SGPR / scratch / LDS of every kernel in liboai_hip.so (code-object metadata): python scripts/kernel_resources.py [filter]"""
import glob, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = os.environ.get("OAI_LIB_PATH") or os.path.join(ROOT, "oai_analysis_2_amd", "liboai_hip.so")
    work = tempfile.mkdtemp()
    shutil.copy(lib, os.path.join(work, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    for co in sorted(glob.glob(os.path.join(work, "lib.so*gfx950*"))):
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            if flt in name:
                print(f"vgpr {get('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {get('sgpr_count'):>4} scratch {get('private_segment_fixed_size'):>5} "
                      f"lds {get('group_segment_fixed_size'):>6}  {name[:150]}")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
