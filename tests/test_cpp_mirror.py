"""The C++ host-side mirror of the reference API (include/vrfhip.hpp) -- the reference is compiled code and
there is no Rust toolchain here, so the mirror above the C ABI is C++.  CPU: it compiles and links against
libvrfhip.so.  GPU: tests/cpp/mirror_test drives from_seed -> Input::new -> output -> prove -> verify against the
golden vector, then batches (IETF and batched Pedersen) with tampered items."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
EXE = os.path.join(HERE, "mirror_test")


def _build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)


def test_cpp_mirror_builds_and_links(native_built):
    _build()
    assert os.path.exists(EXE)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libvrfhip.so" in out and "not found" not in out.split("libvrfhip.so")[1].splitlines()[0]


@pytest.mark.gpu
def test_cpp_mirror_kat_and_batches(kat):
    _build()
    v, p = kat["ietf"][0], kat["pedersen"][0]
    assert p["seed"] == v["seed"] and p["alpha"] == v["alpha"]
    args = [v["seed"], v["alpha"], v["ad"], v["pk"], v["h"], v["gamma"], v["beta"], v["c"], v["s"],
            p["ad"], p["blinding"], p["pk_com"], p["r"], p["ok"], p["s"], p["sb"]]
    # suites::secp256r1 through the same templates: RFC 9381 B.1 example 10 (tests/golden/rfc9381_p256_sha256_tai.json)
    import json
    w = json.load(open(os.path.join(os.path.dirname(HERE), "golden", "rfc9381_p256_sha256_tai.json")))["vectors"][0]
    args += [w["sk"], w["pk"], w["alpha"], w["h"], w["pi"], w["beta"]]
    # empty hex fields (e.g. ad = "") must survive the command line
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mirror_test ok" in r.stdout and "mirror_test p256 ok" in r.stdout
