"""The two places where a result of this repository depends on an optimiser decision (tools/repro/README.md) are
watched, not trusted: __graft_entry__.build() also builds each of them WITHOUT its guard (ab/canary_*.so), and this
file runs the tests that found the sensitivity against those builds in a child process.  The outcome is held against
tests/canaries.json — a canary that starts to pass (a compiler upgrade fixed it: the guard can go) or to fail (it
moved) fails the suite with a message that says which."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANARIES = {k: v for k, v in json.load(open(os.path.join(ROOT, "tests", "canaries.json"))).items() if not k.startswith("_")}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CANARIES))
def test_unguarded_build_behaves_as_recorded(name):
    spec = CANARIES[name]
    lib = os.path.join(ROOT, "ab", name + ".so")
    if not os.path.exists(lib):
        pytest.skip(f"{lib} was not built (build() builds it where hipcc exists)")
    env = dict(os.environ, LINS_IESKF_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + spec["tests"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = "\n".join(p.stdout.decode(errors="replace").splitlines()[-6:])
    assert p.returncode in (0, 1), f"the canary run itself broke (rc {p.returncode}):\n{tail}"
    got = "pass" if p.returncode == 0 else "fail"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "canaries.log"), "a") as f:
        f.write(f"{name}: {got} (recorded: {spec['expect']})\n")
    assert got == spec["expect"], (f"{name} ({spec['what']}) now {got}s the tests that found it — recorded: {spec['expect']}.  "
                                   f"The compiler's behaviour moved: tools/repro/README.md, tests/canaries.json.\n{tail}")
