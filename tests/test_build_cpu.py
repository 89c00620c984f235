"""Checks on the BUILT device code (no GPU needed: the gfx950 code objects are disassembled with the ROCm LLVM tools)."""
import os
import shutil
import subprocess
import tempfile

import pytest

from cases import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "sde4mbrl_px4_amd", "csrc")


def _m0_lines(obj):
    tmp = tempfile.mkdtemp()
    try:
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, os.path.join(tmp, "fatbin")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + os.path.join(tmp, "fatbin"), "--output=" + os.path.join(tmp, "co"), "--unbundle"])
        dis = subprocess.Popen([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, "co")], stdout=subprocess.PIPE)
        out = subprocess.run(["grep", "-w", "m0"], stdin=dis.stdout, capture_output=True, text=True).stdout
        dis.wait()
        return [ln.split("//")[0].strip() for ln in out.splitlines()]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("obj", ["sdempc_kernels_duo2.o", "sdempc_kernels_duo4.o"])
def test_only_the_noise_dma_block_writes_m0_in_the_duo_kernels(obj):
    """duo_noise_request (sdempc_duo.inc.h) sets M0 inside an asm block the compiler knows nothing about (declared through the builtin the
    compiler serialised the DMA against every LDS read). The block saves M0 on entry and restores it on exit, so whatever the compiler keeps
    there survives; as a second line of defence every M0 access of the built code must be one of the block's own — per request one save
    (s_mov_b32 sN, m0), the row address and the restore (two s_mov_b32 m0, sN) and five s_add_u32 m0, m0, 0x80."""
    path = os.path.join(CSRC, obj)
    if not (os.path.exists(path) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and shutil.which("objcopy")):
        pytest.skip("built objects or the LLVM tools are not here")
    lines = _m0_lines(path)
    import re
    saves = [ln for ln in lines if re.match(r"s_mov_b32 s\d+, m0$", ln)]
    movs = [ln for ln in lines if ln.startswith("s_mov_b32 m0, s")]
    adds = [ln for ln in lines if ln.startswith("s_add_u32 m0, m0, 0x80")]
    assert len(lines) > 0 and len(saves) + len(movs) + len(adds) == len(lines), [ln for ln in lines if ln not in movs and ln not in adds and ln not in saves][:5]
    assert len(movs) == 2 * len(saves) and len(adds) == 5 * len(saves)
