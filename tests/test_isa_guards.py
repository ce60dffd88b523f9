"""Guards on the COMPILED form of two kernels whose correctness rests on what hipcc emits, not only on the source
(ADVICE r4).  CPU-only: hipcc cross-compiles gfx950 without a device.  A toolchain upgrade that changes these shapes is
not necessarily wrong - it must be re-read by a human, and these tests make sure it is."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "deepsphere-weather_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")

pytestmark = pytest.mark.skipif(HIPCC is None, reason="hipcc not found")


def _device_asm(src, tmp_path):
    out = os.path.join(str(tmp_path), src.replace(".hip", ".s"))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(CSRC, src),
                    "-o", out], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def _functions(asm):
    for chunk in re.split(r"\n\s*\.globl\s+", asm)[1:]:
        name = chunk.split("\n", 1)[0].strip()
        if "s_endpgm" in chunk:
            yield name, chunk[:chunk.index("s_endpgm")]


def test_spmm1_dma_kernel_vmem_sequence(tmp_path):
    """dsw_spmm1s.hip counts its `s_waitcnt vmcnt(n)` by hand: n = this wave's LDS-DMA pieces of the next sample + its
    previous store.  The count is SAFE whenever the hardware sees at least the counted operations (extra VMEM operations
    only make the wait stricter) and UNSAFE if an operation the count includes is not issued.  Pin what the count assumes:
    one store instruction per sample, every DMA of the source present, no scratch traffic (a spill between the DMAs would
    still be safe, but means the register budget the kernel was tuned for is gone), one barrier per sample."""
    asm = _device_asm("dsw_spmm1s.hip", tmp_path)
    seen = 0
    for name, body in _functions(asm):
        m = re.search(r"spmm1_dma_kernelILb([01])ELi(\d)ELb([01])ELi(\d+)EE", name)
        if not m:
            continue
        seen += 1
        nst, hz = int(m.group(2)), int(m.group(3))
        n_dma = len(re.findall(r"global_load_lds_dwordx4", body))
        # static DMA sites: loop = NST pieces (+ 1 epilogue row); prologue = first sample (NST + HZ) and second sample (NST),
        # which the compiler may fold into a loop over the pieces - so at least the loop's sites plus one prologue group
        assert n_dma >= (nst + hz) + (nst + hz), (name, n_dma)
        assert len(re.findall(r"global_store_dwordx4", body)) == 1, name      # ONE store per sample (n_store)
        assert len(re.findall(r"scratch_(load|store)", body)) == 0, name
        assert len(re.findall(r"\ss_barrier", body)) == 1, name
        # the counted waits are there: vmcnt taken from a scalar register is emitted as a small jump table of s_waitcnt
        assert len(re.findall(r"s_waitcnt vmcnt\(\d+\)", body)) >= 3, name
    assert seen == 16, seen


def test_x3s_balanced_pieces_are_released_before_the_flag(tmp_path):
    """dsw_gemm_x3s.hip, balanced decomposition: a workgroup parks a partial tile with device-scope (sc1) buffer stores,
    then raises a flag another workgroup spins on.  Every wave must drain its own stores (`s_waitcnt vmcnt(0)`) BEFORE the
    workgroup barrier that precedes the flag store - the compiler does not do it on its own."""
    asm = _device_asm("dsw_gemm_x3s.hip", tmp_path)
    checked = 0
    for name, body in _functions(asm):
        if "ts_gemm_x3s_kernel" not in name:
            continue
        lines = body.split("\n")
        idx = [i for i, ln in enumerate(lines) if re.search(r"buffer_store_dword\s.*\bsc1\b", ln)]
        if not idx:
            continue
        # the last parked-piece store, then: vmcnt(0) ... s_barrier ... the flag store (global_store_dword ... sc1)
        tail = lines[idx[-1] + 1: idx[-1] + 80]
        i_wait = next((i for i, ln in enumerate(tail) if re.search(r"s_waitcnt\s+vmcnt\(0\)\s*$", ln)), None)
        i_bar = next((i for i, ln in enumerate(tail) if re.search(r"\ss_barrier", ln)), None)
        i_flag = next((i for i, ln in enumerate(tail) if re.search(r"global_store_dword\s.*\bsc1\b", ln)), None)
        assert i_wait is not None and i_bar is not None and i_flag is not None, name
        assert i_wait < i_bar < i_flag, (name, i_wait, i_bar, i_flag)
        checked += 1
    assert checked >= 4, checked
