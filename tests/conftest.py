import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- evidence lines -----------------------------------------------------------------------------------------------------------------
# The GPU tests PRINT what they measured (perf floors, parity distances per stage, the config tables); `pytest -q` swallows the captured
# output of passing tests, so the driver's log showed pass / fail only.  Every such line is collected from the captured output of each
# test (passed or failed) and printed again at the end of the run -- the last lines of the log, where the GPU-test tail is cut from.
EVIDENCE_PREFIXES = ("PERF_FLOOR ", "PARITY ", "CONFIG1 ", "CONFIG3 ", "CONFIG4 ", "CONFIG5 ", "GEOM384 ", "ROUNDING ")
_EVIDENCE = []


def pytest_runtest_logreport(report):
    if report.when != "call":
        return
    for line in (getattr(report, "capstdout", "") or "").splitlines():
        if line.startswith(EVIDENCE_PREFIXES):
            _EVIDENCE.append(line.rstrip())


def _evidence_digest(lines):
    """every PERF_FLOOR / CONFIG / GEOM384 line; of the (many) PARITY and ROUNDING rows the worst ones only"""
    keep = [l for l in lines if l.startswith(("PERF_FLOOR ", "CONFIG", "GEOM384 "))]
    import re
    par = [l for l in lines if l.startswith("PARITY ")]
    scored = []
    for l in par:
        m = re.search(r"vs(?: |_bf16_)mirror ([0-9.e+-]+)", l)
        scored.append((float(m.group(1)) if m else -1.0, l))
    halves = [x for x in scored if re.search(r"(attention|MLP) block teacher-forced", x[1]) and x[0] >= 0]
    if halves:
        keep.append("PARITY worst half-layer row of %d: %s" % (len(halves), max(halves)[1][7:].strip()))
    stages = [x for x in scored if x not in halves and x[0] >= 0]
    if stages:
        keep.append("PARITY worst stage row of %d: %s" % (len(stages), max(stages)[1][7:].strip()))
    keep += [x[1] for x in scored if x[0] < 0][:12]                      # the rows without a mirror column (token agreement, next token, ...)
    rnd = [l for l in lines if l.startswith("ROUNDING ") and "worst" in l]
    if rnd:
        w = max(rnd, key=lambda l: float(re.search(r"worst ([0-9.]+) ulp", l).group(1)))
        keep.append("ROUNDING worst of %d kernel rows: %s" % (len(rnd), w[9:].strip()))
    return keep


def pytest_terminal_summary(terminalreporter):
    if not _EVIDENCE:
        return
    terminalreporter.section("measured by the tests of this run (tests/conftest.py)")
    try:
        lines = _evidence_digest(_EVIDENCE)
    except Exception as e:                      # a line in a format the digest does not know must not turn a green run red: print everything instead
        lines = [f"(digest failed: {e!r}; raw lines follow)"] + _EVIDENCE
    for line in lines:
        terminalreporter.write_line(line)


def load_golden(name):
    """Returns (arrays, weights) — weights are the 'w::'-prefixed entries as torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs, w = {}, {}
    for k in z.files:
        if k.startswith("w::"):
            w[k[3:]] = torch.from_numpy(z[k])
        else:
            arrs[k] = z[k]
    return arrs, w


def scaled_decoder(w, scale):
    """The state dict of oracle/gen_fixtures.py's second generate() pair: every 2-D `model.layers.*` matrix x scale."""
    return {k: (v * float(scale) if (k.startswith("model.layers.") and v.ndim == 2) else v) for k, v in w.items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.fixture(scope="session")
def golden():
    return load_golden
