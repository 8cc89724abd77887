"""Kernels of different streams sharing a CU (GPU): the product kernels must give the same bits beside an MFMA kernel.

Round 2 found that packed-fp32 instructions with op_sel return a wrong low result now and then while a wave of another
kernel executes 16-bit MFMAs on the same CU (ocr_amd/csrc/crnn_kernels_nopk.hip has the measurements).  The tests:
tools/corun_test (built by ocr_amd/csrc/Makefile) runs conv0_win4_h3_kernel beside the kernels that exposed it, and the
built library must not contain such an instruction anywhere (tools/check_isa.py; runs without a GPU).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "corun_test")
LIB = os.path.join(ROOT, "ocr_amd", "libcrnn_hip.so")


def test_library_has_no_packed_fp32_with_op_sel():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), LIB], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("aggressor", [1, 11, 14])   # the classifier filter GEMM of one crop; bf16 / fp16 MFMA on every CU
def test_conv0_win4_beside_mfma_kernels(aggressor):
    assert os.path.exists(TOOL), "tools/corun_test missing: make -C ocr_amd/csrc"
    r = subprocess.run([TOOL, str(aggressor), "3000"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and " 0 of 3000 iterations differ" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("aggressor", [1, 14])
def test_cls_select_beside_mfma_kernels(aggressor):
    """The other kernel that was moved to crnn_kernels_nopk.hip: the classifier's candidate selection and exact fp32 scoring
    (cls_select_kernel, fp16 logits) as the victim -- its arg-max indices on 2048 rows must not change beside the MFMA kernels."""
    assert os.path.exists(TOOL), "tools/corun_test missing: make -C ocr_amd/csrc"
    r = subprocess.run([TOOL, str(aggressor), "1500", "256", "2"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and " 0 of 1500 iterations differ" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("split_nt", ["auto", "1"])
def test_pipelined_batches_soak_equals_synchronous_results(split_nt):
    """Round 3 found a second way in which kernels of different lanes disturb each other: a kernel that keeps registers in
    AccVGPRs (more than 256 per lane: the chained Winograd transforms, the recurrences) returned a few wrong words when a wave
    of another kernel shared its SIMD -- one wrong line in 70 to 800 pipelined batches of 256 when small kernels of a third
    stream are on the device too (this soak's own comparison kernels are: with the comparison on the host the old build showed
    nothing in 60 000 batches, and the product-only 2M-line job 3 wrong lines in 33 M), never with one batch on the device at
    a time, so the 48-batch tests did not see it.  Such kernels now claim the whole register file
    (CRNN_WHOLE_REGISTER_FILE; tools/check_isa.py fails the build otherwise).  tools/soak_pipe.py queues batches back to back
    on two lanes and compares every batch's raw arg-max rows with a synchronous call's: 5000 batches here (before the fix
    20 to 200 of them were wrong, depending on the box); DESIGN.md has the 100 000-batch runs.  split_nt: the recurrence form
    of the pipelined path (auto = two row tiles per group from 128 lines) and the one-tile form."""
    env = dict(os.environ)
    if split_nt != "auto":
        env["CRNN_SPLIT_NT"] = split_nt
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_pipe.py"), "--batches", "5000"], capture_output=True,
                       text=True, timeout=170, env=env)
    assert r.returncode == 0 and '"bad_batches": 0' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert '"split_timeouts": 0' in r.stdout
