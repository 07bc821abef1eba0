import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


class Golden:
    """One golden case: metadata + lazily loaded reference outputs."""

    def __init__(self, name, meta):
        self.name, self.meta = name, meta

    def __getitem__(self, key):
        return np.load(os.path.join(GOLDEN, self.meta["files"][key]))

    def __getattr__(self, key):
        try:
            return self.__dict__["meta"][key]
        except KeyError:
            raise AttributeError(key)


@pytest.fixture(scope="session")
def golden(manifest):
    def get(name):
        if name not in manifest["cases"]:
            pytest.skip("golden case %s not generated" % name)
        return Golden(name, manifest["cases"][name])
    return get


def golden_cases(op=None, prefix=None):
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        m = json.load(f)
    out = []
    for k, v in m["cases"].items():
        if op is not None and v.get("op") != op:
            continue
        if prefix is not None and not k.startswith(prefix):
            continue
        out.append(k)
    return sorted(out)
