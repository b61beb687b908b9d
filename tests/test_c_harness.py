"""The C ABI driven from plain C (tests/c/abi_harness.c): hand-built Arrow C Data Interface
structs, no Python in the data path.  On the GPU box the harness checks resident and
streaming execution, PARTIAL -> merge, device-resident state and the error convention;
without a GPU it must refuse loudly (there is no CPU fallback to fall into)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "c", "abi_harness")


def _run():
    if not os.path.exists(HARNESS):
        pytest.fail(f"{HARNESS} is missing: build it with `make -C tests/c` (or __graft_entry__.build())")
    return subprocess.run([HARNESS], capture_output=True, text=True, timeout=120)


@pytest.mark.gpu
def test_c_harness_on_the_gpu():
    r = _run()
    assert r.returncode == 0 and r.stdout.strip() == "OK", (r.returncode, r.stdout, r.stderr)


def test_c_harness_without_a_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run()
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_native_feeder_streams_every_shape_of_input():
    """tests/c/stream_bench, small: sixteen threads push 8192-row batches through the C ABI and the
    merged result is checked by the feeder itself (COUNT(*) adds up to the rows fed; C3: the ten
    leading ids in order).  The service column dictionary encoded and as plain Utf8 (coded on the
    device once the dictionary has its strings) must give the same groups."""
    import json

    exe = os.path.join(ROOT, "tests", "c", "stream_bench")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} is missing: build it with `make -C tests/c` (or __graft_entry__.build())")
    groups = {}
    for shape in ("c2", "c2plain", "c3"):
        r = subprocess.run([exe, "6000000", "16", "2", "8192", shape], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (shape, r.returncode, r.stdout, r.stderr)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["workload"] == shape and line["rows"] == 6000000
        groups[shape] = line["groups"]
    assert groups["c2"] == groups["c2plain"] > 0 and groups["c3"] > 0
