import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x if dtype is None else x.to(dtype)


@pytest.fixture(scope="session")
def full_table_cache():
    """12.6 M-entry hash table regenerated from the fixture seed (portable mix32 stream)."""
    from oracle import hashgrid
    cache = {}

    def get(seed, scale):
        key = (int(seed), float(scale))
        if key not in cache:
            cache[key] = hashgrid.init_table(hashgrid.make_spec(), key[0], key[1], "mix32")
        return cache[key]
    return get


FIELD_KEYS = ("base.w0", "base.b0", "base.wo", "base.bo", "head.w0", "head.b0",
              "head.w1", "head.b1", "head.wo", "head.bo")


def field_params_from(g, table):
    p = {k: t(g[k]) for k in FIELD_KEYS}
    p["hash"] = table
    return p


def rel_err(a, b):
    """max |a - b| / max |b|: error relative to the tensor's scale"""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp(min=1e-30))


def elem_err(a, b, floor=1e-3):
    """max over elements of |a - b| / max(|b|, floor * max |b|): element-wise relative error, with elements below
    `floor` of the tensor's scale judged against that floor (an fp32 result has no relative accuracy at its zeros)"""
    a, b = torch.as_tensor(a).double().detach(), torch.as_tensor(b).double().detach()
    den = b.abs().clamp(min=floor * float(b.abs().max().clamp(min=1e-30)))
    return float(((a - b).abs() / den).max())
