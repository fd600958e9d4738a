"""Loaders for the committed golden vectors (tests/golden/)."""
import os

import numpy as np

from oracle import kaldi_io as kio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpath(name):
    return os.path.join(GOLDEN, name)


def load_am_gmm(name):
    """Returns dict(dim, weights, means_invvars, inv_vars, pdf_offsets) from a .mdl holding an AmDiagGmm."""
    r = kio.Reader(open(gpath(name), "rb").read())
    binary = r.read_binary_marker()
    dim, pdfs = r.am_diag_gmm(binary)
    off = np.concatenate([[0], np.cumsum([len(p["weights"]) for p in pdfs])]).astype(np.int32)
    return dict(dim=dim, pdf_offsets=off,
                weights=np.concatenate([p["weights"] for p in pdfs]),
                means_invvars=np.concatenate([p["means_invvars"] for p in pdfs]),
                inv_vars=np.concatenate([p["inv_vars"] for p in pdfs]))


def load_am_nnet(name):
    r = kio.Reader(open(gpath(name), "rb").read())
    return r.am_nnet(r.read_binary_marker())


def load_ark(name):
    return kio.read_ark(gpath(name))


def load_vec(name):
    return kio.read_vector_file(gpath(name))


def load_gselect(name):
    """Returns ({key: [list of index lists per frame]}, {key: total log-likelihood})."""
    sel, tot = {}, {}
    for line in open(gpath(name)):
        f = line.split()
        if f[1] == "TOTAL":
            tot[f[0]] = float(f[2])
        else:
            sel.setdefault(f[0], []).append([int(x) for x in f[2:]])
    return sel, tot
