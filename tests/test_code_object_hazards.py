"""Static check of the gfx950 code objects inside libhipjpeg_ext.so (no GPU needed).

Round 1 hit a compiler hazard: hipcc folded a byte-array index into the BASE of a scalar load and emitted
`s_load_dwordx2 s[..], s[base+cc], soffset offset:0x20` -- an SMEM load whose base is not dword aligned returns the wrong
dwords on gfx950 (wild pointer, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; DESIGN.md "A compiler hazard worth recording").
The descriptors were reshaped so that every scalar load uses an aligned base plus an IMMEDIATE offset; this test keeps it
that way: any s_load / s_buffer_load whose offset operand is an SGPR fails the build check."""
import os
import re
import shutil
import subprocess

import pytest

from nvimagecodec_amd import _native as N

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def device_disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    d = tmp_path_factory.mktemp("codeobj")
    so = os.path.join(d, "lib.so")
    shutil.copy(N.LIB_PATH, so)
    subprocess.run([OBJDUMP, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    objs = sorted(f for f in os.listdir(d) if "gfx950" in f)
    assert objs, "no gfx950 code object found inside the library"
    text = []
    for o in objs:
        text.append(subprocess.run([OBJDUMP, "-d", os.path.join(d, o)], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def test_library_carries_only_gfx950_device_code(device_disassembly):
    assert "s_endpgm" in device_disassembly


def test_no_scalar_load_with_an_sgpr_offset(device_disassembly):
    """Kernels whose descriptors hold byte arrays (DecodeImage / HuffImage / EncodeImage ...) must not index them through the
    register-offset form at all.  The progressive-scan kernels (prog_*) do use it -- their descriptors (ProgImage / ProgScan,
    progressive_gpu_core.h) consist of 32- and 64-bit members only, so every such offset is a multiple of four by
    construction; test_progressive_descriptors_have_no_sub_dword_members pins that."""
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", device_disassembly)
    total, bad = 0, []
    for b in blocks:
        head = b.split("\n", 1)[0]
        for l in b.splitlines():
            if not re.search(r"\bs_(buffer_)?load_dword", l):
                continue
            l = l.split("//")[0].strip()
            total += 1
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            # operands: destination, base (SGPR pair / resource), offset.  The offset must be an immediate.
            if len(ops) != 3 or not re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", ops[2]):
                if "prog_" not in head:
                    bad.append(head[:100] + "  " + l)
    assert total > 100  # the kernels do read their descriptors with scalar loads
    assert not bad, "scalar loads with a register offset (unaligned-base hazard on gfx950):\n" + "\n".join(bad[:10])


def test_progressive_descriptors_have_no_sub_dword_members():
    """Textual check of progressive_gpu_core.h: between `struct ... ProgScan {` / `ProgImage {` and the closing brace no member
    is declared with an 8- or 16-bit type (pointers to such types are fine)."""
    src = open(os.path.join(os.path.dirname(N.LIB_PATH), "csrc", "progressive_gpu_core.h")).read()
    for name in ("ProgScan", "ProgImage"):
        m = re.search(r"struct (?:alignas\(16\) )?%s \{(.*?)\n\};" % name, src, re.S)
        assert m, name
        for line in m.group(1).splitlines():
            decl = line.split("//")[0].strip()
            if not decl:
                continue
            assert not re.match(r"(u?int8_t|u?int16_t|bool|char)\s+[a-z_]", decl), (name, decl)


def test_kernels_do_not_spill_in_the_everyday_configuration(device_disassembly):
    """No build of the luma/colour kernel (generic layout included, since round 3) nor the plane IDCT kernel may touch scratch."""
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", device_disassembly)
    checked = 0
    for b in blocks:
        head = b.split("\n", 1)[0]
        # luma_color_kernel<HS, VS, LAYOUT> and idct_plane_kernel
        if re.search(r"luma_color_kernelILi\dELi\dELi[012]EE", head) or "idct_plane_kernel" in head:
            checked += 1
            assert "scratch_" not in b, head
    assert checked >= 14
