"""The C-ABI library loads and exports every symbol include/yadcc_dispatch.h declares
(no compute here: this runs without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT, has_gpu
from yadcc_amd import binding

HEADER = os.path.join(ROOT, "include", "yadcc_dispatch.h")


@pytest.fixture(scope="module")
def libydc():
    if not os.path.exists(binding.LIB_PATH):
        subprocess.check_call(["make", "-s", "lib"], cwd=ROOT)
    return ctypes.CDLL(binding.LIB_PATH)


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"\b(ydc_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(binding.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(libydc):
    for name in declared_symbols():
        assert hasattr(libydc, name), name


def test_error_strings(libydc):
    libydc.ydc_strerror.restype = ctypes.c_char_p
    assert libydc.ydc_strerror(0) == b"ok"
    assert b"no CPU fallback" in libydc.ydc_strerror(-3)
    assert libydc.ydc_abi_version() == binding.ABI_VERSION


@pytest.mark.skipif(has_gpu(), reason="only meaningful without a GPU")
def test_fails_loudly_without_gpu():
    """No device => YDC_ERR_NO_DEVICE, never a silent CPU path."""
    with pytest.raises(binding.YdcError, match="no usable gfx950 device"):
        binding.Context()


def test_walk_groups_keeps_its_two_accumulation_registers(tmp_path):
    """k_walk_groups (wide_kernel.h) parks two loads in flight in a254 / a255 across separate asm
    statements; the clobber lists do not tell the compiler that their contents must survive in
    between. So the build is checked: in the kernel's gfx950 code nothing but those statements'
    own instructions names an accumulation register at all (no v_accvgpr_*, no spills to AGPRs)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "dev.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only",
                        "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "yadcc_amd", "csrc"),
                        os.path.join(ROOT, "yadcc_amd", "csrc", "ydc_api.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    inside, kernels, parked = False, 0, 0
    own = re.compile(r"^\s*(ds_write_b32 v\d+, a25[45]|global_load_dword a25[45], v\[\d+:\d+\], off)\s*$")
    for line in out.read_text().splitlines():
        if re.match(r"^_ZN3ydc13k_walk_groups.*:", line):
            inside, kernels = True, kernels + 1
        elif line.startswith(".Lfunc_end"):
            inside = False
        elif inside and not line.lstrip().startswith((";", ".")):
            code = line.split(";")[0]
            if re.search(r"\ba(\d+|\[\d+:\d+\])", code) or "accvgpr" in code:
                assert own.match(code), line
                parked += 1
    assert kernels == 2 and parked >= 10, (kernels, parked)
