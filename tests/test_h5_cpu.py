"""HDF5 container of XDMFTensorOutput (mrl_h5_*, host code): the writer against an independent reader of the file format
(tests/h5_subset_reader.py) and, where the image has the HDF5 command-line tools, against libhdf5 itself (h5dump -- the same tool
tests/golden/make_fixtures.py reads the reference's gold files with)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from marlin_amd.api import H5File, MarlinHipError
from tests.conftest import load_golden
from tests.h5_subset_reader import read_h5

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def _h5dump(path, name, dtype, shape):
    """dataset `name` through libhdf5's own reader (binary dump, little endian), as make_fixtures.py does for the gold files"""
    out = str(path) + "." + name + ".raw"
    subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", out, str(path)], check=True, capture_output=True)
    return np.fromfile(out, dtype=dtype).reshape(shape)


def test_empty_file_is_valid(tmp_path):
    p = tmp_path / "empty.h5"
    H5File(p).close()
    assert read_h5(p) == {}
    if H5DUMP:
        txt = subprocess.run([H5DUMP, "-H", str(p)], check=True, capture_output=True, text=True).stdout
        assert 'GROUP "/"' in txt and "DATASET" not in txt


def test_many_datasets_all_types(tmp_path):
    """300 datasets (five symbol-table nodes), the four element types of XDMFTensorOutput.C:331-341, ranks 1-3, written in the
    order an output object produces them (c.0, mu.0, c.1, ...: NOT the sorted order the group B-tree stores)"""
    rng = np.random.default_rng(7)
    p = tmp_path / "many.h5"
    want = {}
    with H5File(p) as f:
        for frame in range(140):
            for name, shape in (("c", (6, 5)), ("mu", (3, 4, 5))):
                want[f"{name}.{frame}"] = rng.standard_normal(shape)
                f.write(f"{name}.{frame}", want[f"{name}.{frame}"])
        want["F_0.0"] = rng.standard_normal(9).astype(np.float32)
        want["ids.0"] = rng.integers(-2**31, 2**31 - 1, (4, 4), dtype=np.int32)
        want["big.0"] = rng.integers(-2**62, 2**62, (7,), dtype=np.int64)
        for k in ("F_0.0", "ids.0", "big.0"):
            f.write(k, want[k])
        for k in range(17):
            want[f"z{k}"] = np.full((2,), float(k))
            f.write(f"z{k}", want[f"z{k}"])
    got = read_h5(p)
    assert sorted(got) == sorted(want) and len(got) == 300
    for k, a in want.items():
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k
    if H5DUMP:
        for k in ("c.0", "c.139", "mu.77", "F_0.0", "ids.0", "big.0", "z16"):
            assert np.array_equal(_h5dump(p, k, want[k].dtype, want[k].shape), want[k]), k
        listing = subprocess.run([H5DUMP, "-n", str(p)], check=True, capture_output=True, text=True).stdout
        assert listing.count("\n dataset ") == 300


def test_file_is_valid_after_every_flush(tmp_path):
    """H5Fflush per output step (XDMFTensorOutput.C:244-246): a reader that opens the file between two frames sees every dataset
    written so far; later frames are appended behind the superseded metadata block"""
    p = tmp_path / "frames.h5"
    f = H5File(p)
    seen = {}
    for frame in range(5):
        a = np.arange(12.0).reshape(3, 4) + frame
        f.write(f"c.{frame}", a)
        seen[f"c.{frame}"] = a
        f.flush()
        got = read_h5(p)
        assert sorted(got) == sorted(seen) and all(np.array_equal(got[k], v) for k, v in seen.items())
        if H5DUMP:
            assert np.array_equal(_h5dump(p, f"c.{frame}", "<f8", (3, 4)), a)
    f.close()
    assert sorted(read_h5(p)) == sorted(seen)


def test_gold_datasets_round_trip(tmp_path):
    """the reference's gold fields (cahnhilliard.h5: c.k 21 x 21, mu.10 20 x 20) through the container, bit for bit"""
    g = load_golden("cahnhilliard_gold.npz")
    p = tmp_path / "gold.h5"
    with H5File(p) as f:
        for k in g.files:
            f.write(k, g[k])
    got = read_h5(p)
    assert sorted(got) == sorted(g.files)
    for k in g.files:
        assert np.array_equal(got[k], g[k])
        if H5DUMP:
            assert np.array_equal(_h5dump(p, k, "<f8", g[k].shape), g[k])


def test_errors(tmp_path):
    f = H5File(tmp_path / "e.h5")
    f.write("c.0", np.zeros((2, 2)))
    with pytest.raises(MarlinHipError, match="already exists"):     # XDMFTensorOutput.C:593-594
        f.write("c.0", np.zeros((2, 2)))
    with pytest.raises(MarlinHipError):
        f.write("a/b", np.zeros(2))
    with pytest.raises(MarlinHipError):
        f.write("r5", np.zeros((1, 1, 1, 1, 1)))
    with pytest.raises(ValueError):
        f.write("c16", np.zeros(2, dtype=np.complex128))
    f.close()
    assert sorted(read_h5(tmp_path / "e.h5")) == ["c.0"]
    with pytest.raises(MarlinHipError):
        H5File(tmp_path / "no_such_dir" / "x.h5")


def test_file_stays_valid_between_a_write_and_the_next_flush(tmp_path):
    """a dataset written after a flush used to overwrite the metadata block the on-disk superblock points at, so the
    file (and every frame flushed before) was unreadable until the next flush.  The reference flushes once per output step
    (XDMFTensorOutput.C:244-246) precisely so that a crash, or a reader, in mid-frame finds the earlier frames."""
    p = tmp_path / "midframe.h5"
    rng = np.random.default_rng(3)
    a0, a1, a2 = rng.standard_normal((4, 5)), rng.standard_normal((4, 5)), rng.standard_normal((300,))
    f = H5File(p)
    f.write("a.0", a0)
    f.flush()
    assert np.array_equal(read_h5(p)["a.0"], a0)
    f.write("a.1", a1)                         # no flush yet: the file on disk still describes exactly {a.0}
    got = read_h5(p)
    assert sorted(got) == ["a.0"] and np.array_equal(got["a.0"], a0)
    if H5DUMP:
        assert np.array_equal(_h5dump(p, "a.0", a0.dtype, a0.shape), a0)
    f.write("big.1", a2)
    got = read_h5(p)
    assert sorted(got) == ["a.0"]
    f.flush()
    got = read_h5(p)
    assert sorted(got) == ["a.0", "a.1", "big.1"] and np.array_equal(got["a.1"], a1) and np.array_equal(got["big.1"], a2)
    f.close()
    got = read_h5(p)
    assert sorted(got) == ["a.0", "a.1", "big.1"] and np.array_equal(got["a.0"], a0)
    if H5DUMP:
        for k, a in (("a.0", a0), ("a.1", a1), ("big.1", a2)):
            assert np.array_equal(_h5dump(p, k, a.dtype, a.shape), a), k
