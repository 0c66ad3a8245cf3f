import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(directory, target):
    subprocess.run(["make", "-C", directory, target], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def built():
    """Product library + CLI and the oracle, built in-tree if missing."""
    lib = os.path.join(ROOT, "rnamotif_amd", "librnamotif_amd.so")
    cli = os.path.join(ROOT, "rnamotif_amd", "bin", "rnamotif")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        _make(os.path.join(ROOT, "rnamotif_amd", "csrc"), "all")
    ora = os.path.join(ROOT, "oracle")
    if not (os.path.exists(os.path.join(ora, "liboracle.so")) and os.path.exists(os.path.join(ora, "rnamotif_oracle"))):
        _make(ora, "liboracle.so")
        _make(ora, "rnamotif_oracle")
    return {"lib": lib, "cli": cli, "oracle_cli": os.path.join(ora, "rnamotif_oracle"),
            "oracle_lib": os.path.join(ora, "liboracle.so")}


@pytest.fixture(scope="session")
def gbrna(tmp_path_factory):
    """The reference's test database (test/gbrna.111.0.fastn), unpacked."""
    import gzip
    import hashlib
    d = tmp_path_factory.mktemp("gbrna")
    path = str(d / "gbrna.111.0.fastn")
    with gzip.open(os.path.join(GOLDEN, "test", "gbrna.111.0.fastn.gz"), "rb") as f:
        data = f.read()
    assert hashlib.md5(data).hexdigest() == "0eca644050c53fdd243b10b5a1e73b5c"
    with open(path, "wb") as f:
        f.write(data)
    return path


@pytest.fixture(scope="session")
def workdir(tmp_path_factory, gbrna):
    """A directory holding the golden descriptors next to the database, so that
    `-descr NAME.descr` produces the same '#RM dfile' line as the pinned runs."""
    import shutil
    d = tmp_path_factory.mktemp("work")
    for f in os.listdir(os.path.join(GOLDEN, "test")):
        if f.endswith(".descr"):
            shutil.copy(os.path.join(GOLDEN, "test", f), str(d / f))
    shutil.copy(os.path.join(GOLDEN, "descr", "trna.descr"), str(d / "trna.efn.descr"))
    os.symlink(gbrna, str(d / "gbrna.111.0.fastn"))
    return str(d)
