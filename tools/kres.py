"""Register / LDS use of every kernel of a built library: python tools/kres.py ocr_amd/libcrnn_hip.so [name-filter]"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
lib = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, os.path.basename(lib))
    shutil.copy(lib, tmp)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL)
    for co in sorted(f for f in os.listdir(td) if "amdgcn" in f):
        notes = subprocess.run([LLVM + "/llvm-readobj", "--notes", os.path.join(td, co)], capture_output=True, text=True).stdout
        cur = {}
        def flush():
            if cur.get("name") and flt in cur["name"]:
                nm = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
                print("%-60s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %s" % (nm[-60:], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"), cur.get("group_segment_fixed_size"), cur.get("private_segment_fixed_size")))
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m: continue
            k, v = m.group(1), m.group(2)
            if k == "agpr_count" and "agpr_count" in cur: flush(); cur = {}
            if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
                if k == "name" and "name" in cur and "vgpr_count" in cur: flush(); cur = {}
                cur[k] = v
        flush()
