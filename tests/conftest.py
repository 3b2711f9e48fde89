import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:                      # `from conftest import ...` in the test modules, also when one file is run alone
    sys.path.insert(0, HERE)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_env_is_reread():
    """librgnn caches its environment switches per call site (rgnn_env_reload): a test that sets one calls ops.reload_env() itself;
    this makes the library read them again once the test's monkeypatch has been undone (autouse fixtures are torn down last)."""
    yield
    ops = sys.modules.get("radargnn_amd.ops")
    if ops is not None:
        ops.reload_env()


def golden_files(prefix=""):
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def record_parity(name: str, **errors) -> None:
    """Print the measured error of an oracle comparison and keep it in gpurun_out/parity_margins.json (merged back from the GPU
    box; the summary committed under profiles/ is what bench.py's `parity_margin` quotes).  Errors are norm-wise per tensor:
    max|a - b| / max|b| against the float64 oracle; the bar is 1e-5."""
    import json
    path = os.path.join(REPO, "gpurun_out", "parity_margins.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        table = json.load(open(path))
    except Exception:
        table = {}
    table[name] = {k: float(v) for k, v in errors.items()}
    with open(path, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    print(f"[parity] {name}: " + ", ".join(f"{k} {float(v):.2e}" for k, v in errors.items()))
