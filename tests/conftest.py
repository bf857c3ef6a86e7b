import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_native_built():
    """The .so files are git-ignored: build them in-tree when a fresh checkout runs the tests."""
    import subprocess
    lib = os.path.join(ROOT, "ark_ec_vrfs_amd", "csrc", "libvrfhip.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.dirname(lib), "-j", "6"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session", autouse=True)
def native_built():
    _ensure_native_built()


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "bandersnatch_sha512_ell2_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx():
    """One libvrfhip context on cuda:0.  No skip and no fallback: on a GPU box a missing or
    broken HIP library must fail the test."""
    from ark_ec_vrfs_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def synth():
    """Deterministic synthetic items of SURVEY.md section 8d via the C oracle (sk_i, msg_i)."""
    from oracle import c_oracle as co, vrf_oracle as o

    def make(n, start=0):
        sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(start + i)), np.uint8) for i in range(n)])
        msg = np.stack([np.frombuffer(o.synth_msg(start + i), np.uint8) for i in range(n)])
        return sk, msg
    return make


def hx(s):
    return np.frombuffer(bytes.fromhex(s), dtype=np.uint8)
