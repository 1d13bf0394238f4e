import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("sfm-learner-chainer_amd")


@pytest.fixture(scope="session")
def ops():
    return importlib.import_module("sfm-learner-chainer_amd.ops")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("sfm-learner-chainer_amd.synth")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")



def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Prints what the tolerance-based tests actually measured (util.parity_note) and, on the GPU box, writes it to
    gpurun_out/parity_stats.txt so that the evidence travels back."""
    import util
    notes = util.PARITY_NOTES
    if not notes:
        return
    terminalreporter.section("parity statistics (%d notes; worst cases in gpurun_out/parity_stats.txt)" % len(notes))
    for line in notes[-25:]:
        terminalreporter.write_line(line)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_stats.txt"), "w") as f:
            f.write("\n".join(notes) + "\n")
        import json
        with open(os.path.join(out, "parity_rows.jsonl"), "w") as f:
            for row in util.PARITY_ROWS:
                f.write(json.dumps(row) + "\n")
    except OSError:
        pass
