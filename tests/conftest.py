import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def capi():
    from krepp_amd import capi as c

    c.load()
    return c


@pytest.fixture(scope="session")
def po():
    import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def synth():
    from krepp_amd import synth as s

    return s


@pytest.fixture(scope="session")
def toy_genomes(synth):
    return synth.evolve_genomes(open(os.path.join(GOLDEN, "tree_toy.nwk")).read(), 20000, seed=7)


def read_fastq_simple(path):
    names, seqs = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines) and lines[i].startswith(b"@"):
        names.append(lines[i][1:].decode())
        seqs.append(lines[i + 1])
        i += 4
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    return names, bases, offs


@pytest.fixture(scope="session")
def toy_reads():
    return read_fastq_simple(os.path.join(GOLDEN, "toy_reads.fq"))


@pytest.fixture(scope="session")
def toy_index_dir():
    return os.path.join(GOLDEN, "toy_index")


def rows_of_oracle(ref):
    r = ref["rows"]
    return sorted((int(a), int(b), float(c)) for a, b, c in zip(r["read"], r["se"], r["d_llh"]) if b != 0)


def assert_rows_close(got, want, tol=1e-6):
    """DIST tolerance of the north star: 1e-6 relative."""
    assert [g[:2] for g in got] == [w[:2] for w in want]
    for g, w in zip(got, want):
        assert abs(g[2] - w[2]) <= tol * abs(w[2]), (g, w)
