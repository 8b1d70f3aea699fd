"""bench.py's other workloads, run the way the driver runs the headline one, so that a driver-side record of
them exists: the JSON line must parse, its parity flags must hold and its numbers must be sane (loose bounds far
below what profiles/ records -- a sanity check, not a performance gate).  Full BASELINE sizes (4 GiB)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _line(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,alphabet,m", [("cfg3", "ACGT", 64), ("cfg3b", "printable-95", 64)])
def test_config3_line(ctx, workload, alphabet, m):
    """BASELINE config 3 (4 GiB, 64-byte pattern; ACGT: good-suffix-dominated for the reference's walker, the
    8-gram walker here) and its printable-95 twin."""
    line = _line("--workload", workload)
    assert line["unit"] == "GB/s" and line["n_gpus"] == 1 and line["dtype"] == "u8"
    assert line["config"]["alphabet"] == alphabet and line["config"]["pattern_bytes"] == m
    assert line["config"]["text_bytes_total"] == 4 << 30
    assert line["parity"]["planted_offsets_exact"] is True and line["config"]["matches"] > 4000
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["algorithmic_bytes_per_launch"] == 4 << 30
    assert 0.2 < roof["kernel_ms"] < 3.0 and 0.3 < roof["frac"] < 1.0
    assert abs(roof["achieved"] - (4 << 30) / (roof["kernel_ms"] * 1e-3) / 1e9) < 2.0
    assert roof["kernel_ms"] <= line["ms_per_step"] * 1.05  # the kernel cannot take longer than the step around it
    print(json.dumps(line))
