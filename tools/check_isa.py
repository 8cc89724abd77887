#!/usr/bin/env python3
"""Build check: no packed-fp32 instruction with op_sel in the device code of a library.

v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel (the LOW result taking the HIGH dword of a source) were measured on
MI355X to give a wrong low result now and then while a wave of another kernel runs 16-bit MFMAs on the same CU
(tools/corun_test.hip).  The compiler chooses those forms by itself, so the Makefile runs this over the built
library: every gfx950 code object is extracted and disassembled, and any such instruction fails the build (move the kernel to
crnn_kernels_nopk.hip, which is compiled without packed fp32).

    python tools/check_isa.py ocr_amd/libcrnn_hip.so
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

# v_pk_mov_b32 moves fp32 lanes with the same op_sel source selection and goes through the same packed data path
PK = re.compile(r"\bv_pk_(fma_f32|mul_f32|add_f32|mov_b32)\b")
SEL = re.compile(r"op_sel:\[([01,]+)\]")


def llvm_bin():
    """llvm-objdump's directory: $LLVM_BIN, else beside the hipcc the Makefile used ($HIPCC: <rocm>/bin/hipcc ->
    <rocm>/lib/llvm/bin), else /opt/rocm."""
    cands = [os.environ.get("LLVM_BIN")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if c and os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    return None


def main():
    lib = sys.argv[1]
    LLVM = llvm_bin()
    if LLVM is None:
        print("check_isa: llvm-objdump not found (set LLVM_BIN to the ROCm LLVM bin directory); the library was NOT checked")
        return 2
    bad = []
    scratch = []
    agpr_of, vgpr_of = {}, {}
    n_pk = 0
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(lib))
        shutil.copy(lib, tmp)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL)
        cos = [f for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            print("check_isa: no device code object found in", lib)
            return 1
        for co in sorted(cos):
            out = subprocess.run([LLVM + "/llvm-objdump", "-d", os.path.join(td, co)], check=True, capture_output=True,
                                 text=True).stdout
            fn = "?"
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    fn = m.group(1)
                    continue
                if PK.search(line):
                    n_pk += 1
                    s = SEL.search(line)
                    if s and "1" in s.group(1):
                        bad.append((fn, line.strip().split("//")[0].strip()))
            # private (scratch) memory: the runtime allocates it on a queue at a kernel's first launch there, which is what
            # crnn_forward_device_noalloc promises not to cause -- no kernel of the library may need any
            notes = subprocess.run([LLVM + "/llvm-readobj", "--notes", os.path.join(td, co)], capture_output=True, text=True).stdout
            name = "?"
            pending_agpr = 0
            for line in notes.splitlines():
                m = re.match(r"\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
                if m and int(m.group(1)) > 0:
                    scratch.append((name, int(m.group(1))))
                # a kernel that keeps registers in AccVGPRs (more than 256 per lane) must claim the whole register file
                # of its SIMD (CRNN_WHOLE_REGISTER_FILE, crnn_kernels.h): beside a wave of another kernel its results
                # went wrong in the pipelined path
                # (.agpr_count is the first key of a kernel's record, before its .name: remembered until .vgpr_count, the last)
                m = re.match(r"\s*-?\s*\.agpr_count:\s+(\d+)", line)
                if m:
                    pending_agpr = int(m.group(1))
                m = re.match(r"\s*\.vgpr_count:\s+(\d+)", line)
                if m:
                    vgpr_of[name] = int(m.group(1))
                    agpr_of[name] = pending_agpr
                    pending_agpr = 0
    if scratch:
        print("check_isa: %d kernels use scratch memory (register spills or stack arrays):" % len(scratch))
        for k, n in scratch:
            print("  %s   %d bytes per lane" % (k, n))
        return 1
    shared = sorted(k for k, a in agpr_of.items() if a > 0 and vgpr_of.get(k, 0) < 512)
    if shared:
        print("check_isa: %d kernels use AccVGPRs without claiming the whole register file (CRNN_WHOLE_REGISTER_FILE):" % len(shared))
        for k in shared:
            print("  %s   %d registers, %d of them AccVGPRs" % (k, vgpr_of.get(k, 0), agpr_of[k]))
        return 1
    if bad:
        kernels = sorted(set(b[0] for b in bad))
        print("check_isa: %d packed-fp32 instructions with op_sel in %d kernels (move them to crnn_kernels_nopk.hip):" % (len(bad), len(kernels)))
        for k in kernels:
            ex = next(b[1] for b in bad if b[0] == k)
            print("  %s   e.g. %s" % (k, ex))
        return 1
    print("check_isa: ok (%d packed-fp32 instructions, none with op_sel; no kernel uses scratch memory; %d kernels with AccVGPRs, "
          "all with the whole register file)" % (n_pk, sum(1 for a in agpr_of.values() if a > 0)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
