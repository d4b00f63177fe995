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
