"""The C-ABI library without a GPU: it loads, exports every function include/amlike.h
declares, and its host-only entry points work; compute entry points are not called."""
import ctypes
import os
import re

import numpy as np
import pytest

from golden_util import load_am_gmm, load_vec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amlike.h")
LIB = os.path.join(ROOT, "kaldi_amd", "csrc", "libamlike.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(amlike_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    assert os.path.exists(LIB), "libamlike.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    import torch  # noqa: F401  (the HIP runtime must be the one torch loads)
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("amlike_init", "amlike_last_error", "amlike_gmm_create", "amlike_gmm_score",
                 "amlike_gmm_score_dev", "amlike_gmm_destroy", "amlike_gmm_compute_gconsts",
                 "amlike_gmm_gauss_loglikes", "amlike_nnet_create", "amlike_nnet_compute",
                 "amlike_nnet_compute_dev", "amlike_nnet_destroy"):
        assert must in names
    assert len(names) >= 25


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/amlike.h but not exported: %s" % missing


def test_no_torch_types_in_the_abi():
    text = open(HEADER).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    code = re.sub(r"//[^\n]*", "", code)
    assert "torch" not in code and "at::" not in code and "std::" not in code and "#include <hip" not in code


def test_host_gconsts_entry_point_matches_golden_bits(lib):
    m = load_am_gmm("gmm_a.mdl")
    want = load_vec("gmm_a_gconsts.vec")
    off = m["pdf_offsets"]
    fp = ctypes.POINTER(ctypes.c_float)
    lib.amlike_gmm_compute_gconsts.restype = ctypes.c_int
    got = []
    for a, b in zip(off[:-1], off[1:]):
        w = np.ascontiguousarray(m["weights"][a:b])
        miv = np.ascontiguousarray(m["means_invvars"][a:b])
        iv = np.ascontiguousarray(m["inv_vars"][a:b])
        gc = np.empty(b - a, np.float32)
        nbad = ctypes.c_int32(-7)
        rc = lib.amlike_gmm_compute_gconsts(w.ctypes.data_as(fp), miv.ctypes.data_as(fp),
                                            iv.ctypes.data_as(fp), ctypes.c_int32(b - a),
                                            ctypes.c_int32(m["dim"]), gc.ctypes.data_as(fp),
                                            ctypes.byref(nbad))
        assert rc == 0 and nbad.value == 0
        got.append(gc)
    np.testing.assert_array_equal(np.concatenate(got).view(np.uint32), want.view(np.uint32))


def test_compute_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib.amlike_init.restype = ctypes.c_int
    rc = lib.amlike_init(0)
    assert rc != 0
    lib.amlike_last_error.restype = ctypes.c_char_p
    assert len(lib.amlike_last_error()) > 0
    # and the Python mirror raises rather than falling back to anything
    from kaldi_amd import gmm
    m = load_am_gmm("gmm_a.mdl")
    with pytest.raises(Exception):
        gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
