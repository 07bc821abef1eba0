"""The production host: JS wrapper (nd4js_amd/js/index.js) over the N-API addon. Skipped where node
or its headers are missing (the addon is only built when /usr/include/node exists)."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "nd4js_amd", "js", "nd4hip_napi.node")
SCRIPT = os.path.join(ROOT, "tests", "js", "node_checks.js")
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the N-API addon not available")


def run(*args):
    r = subprocess.run([NODE, SCRIPT] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    return r.stdout


@needs_node
def test_js_wrapper_validation_and_loud_failure():
    out = run("cpu", GOLDEN)
    assert "node cpu checks ok" in out
    # the JS chain planner and the Python one choose the same parenthesisation for every golden chain
    import json
    from nd4js_amd import la
    plans = json.loads([ln for ln in out.splitlines() if ln.startswith("PLANS ")][0][6:])
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        cases = json.load(f)["cases"]
    assert len(plans) >= 8
    for name, cut in plans.items():
        assert cut == la.chain_plan(cases[name]["shapes"]), name


@needs_node
@pytest.mark.skipif(not os.path.exists("/root/reference/dist/nd.js"), reason="reference bundle only exists in the build container")
def test_install_patches_reference_module():
    assert "node install checks ok" in run("install", "/root/reference/dist/nd.js")


@needs_node
@pytest.mark.gpu
def test_js_wrapper_against_golden_on_gpu():
    assert "node gpu checks ok" in run("gpu", GOLDEN)
