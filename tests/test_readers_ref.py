"""mustache_amd.readers against the REFERENCE's own readers: tests/golden/readers_ref.npz holds what the reference's
read_cooler / read_mcooler / read_hic_file (mustache.py:300-592) and main()'s chromosome enumeration (:1019-1033) return
for the container of tests/readers_case.py (made by tests/golden/make_golden.py `readers`, which imports the reference and
serves it the same stand-in `cooler` / list-backed `hicstraw` objects used here).  Record SETS must be identical (the
reference's output order is Python set-iteration order and carries no meaning); values bit for bit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import readers_case as rc           # noqa: E402
from hic_writer import write_hic    # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "readers_ref.npz"))


def expect(key):
    return GOLD[key + "_x"].astype(np.int64), GOLD[key + "_y"].astype(np.int64), GOLD[key + "_v"]


def same(got, key):
    gx, gy, gv = rc.as_sorted(*got)
    ex, ey, ev = expect(key)
    assert len(gx) == len(ex), (key, len(gx), len(ex))
    assert np.array_equal(gx, ex) and np.array_equal(gy, ey) and np.array_equal(gv, ev), key


@pytest.fixture()
def standins(monkeypatch, tmp_path):
    monkeypatch.setitem(sys.modules, "cooler", rc.standin_cooler())
    fake = rc.fake_hicstraw(rc.RES)
    monkeypatch.setitem(sys.modules, "hicstraw", fake)
    cool, mcool = rc.write_cool_files(tmp_path)
    return cool, mcool, fake


def test_fixture_is_not_trivial():
    x, y, v = expect("cool_chrA")
    span = rc.DIST // rc.RES
    assert len(x) > 5000 and (y - x).max() == span and v.min() > 0 and not np.isnan(v).any()
    cx, cy, cv = next(c[2:] for c in rc.container(rc.RES) if c[0] == "chrA")
    assert np.isnan(cv).sum() > 100 and (cv == 0).sum() > 50 and (cv < 0).sum() > 20 and (cy - cx).max() > span
    # records on the seams of the reference's windows (bins 1600 / 2000 / 3200 / 3600 ...) are in, each once
    keys = x * 100000 + y
    assert len(np.unique(keys)) == len(keys)
    for b in (1600, 1999, 2000, 3200, 3599, 3600):
        assert ((x == b) | (y == b)).any(), b


def test_read_cooler_and_mcooler_equal_the_reference(standins):
    from mustache_amd.readers import read_cooler, read_mcooler
    cool, mcool, _ = standins
    for name, _ in rc.CHROMS:
        x, y, v, res = read_cooler(cool, rc.DIST, name, name, False)
        assert res == rc.RES
        same((x, y, v), "cool_" + name)
        same(read_mcooler(mcool, rc.DIST, name, name, 2 * rc.RES, False), "mcool_" + name)
    with pytest.raises(NameError):
        read_cooler(cool, rc.DIST, "chrZ", "chrZ", False)


def test_read_hic_file_straw_backend_equals_the_reference(standins, monkeypatch):
    from mustache_amd.readers import read_hic_file
    _, _, fake = standins
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "hicstraw")
    for name, _ in rc.CHROMS:
        same(read_hic_file("case.hic", False, False, rc.DIST, name, name, rc.RES), "hic_" + name)
    assert {c[0] for c in fake.calls} == {str(GOLD["hic_norm_default"])}
    del fake.calls[:]
    same(read_hic_file("case.hic", "VC", rc.CHROMS[0][1], rc.DIST, "chrA", "chrA", rc.RES), "hic_chrA_sized_VC")
    # the very windows the reference asked straw for
    assert [[c[2], c[3]] for c in fake.calls] == GOLD["hic_calls_chrA_sized_VC"].tolist()
    assert {c[0] for c in fake.calls} == {"VC"}


def write_case_hic(path, version):
    cont = rc.container(rc.RES)
    chroms = [("All", 1)] + [(n, s) for n, s, _, _, _ in cont]
    mats, norms = {}, {}
    for i, (name, size, x, y, v) in enumerate(cont, start=1):
        mats[i] = {rc.RES: (x, y, v)}
        norms[("KR", i, rc.RES)] = np.ones(-(-size // rc.RES) + 1)
    write_hic(path, chroms, mats, norms, version=version, block_bin_count=97, float_counts=True, short_coords=False)


@pytest.mark.parametrize("version", [8, 9])
def test_native_hic_reader_equals_the_reference_on_the_same_contacts(tmp_path, monkeypatch, version):
    """The native reader (one pass over the blocks near the diagonal) returns what the reference's windowed straw walk
    returns for the same contacts: the file is written from the container the reference was served from (values are
    multiples of 1/64 below 32, exact as the format's float32 counts; KR vector of ones)."""
    from mustache_amd.readers import read_hic_file, list_chromosomes
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "native")
    p = str(tmp_path / "case.hic")
    write_case_hic(p, version)
    for name, size in rc.CHROMS:
        same(read_hic_file(p, False, False, rc.DIST, name, name, rc.RES), "hic_" + name)
    same(read_hic_file(p, "KR", rc.CHROMS[0][1], rc.DIST, "chrA", "chrA", rc.RES), "hic_chrA_sized_VC")
    assert list_chromosomes(p, rc.RES) == [str(s) for s in GOLD["main_hic_chroms"]]


def test_whole_genome_chromosome_lists_equal_the_reference_main(standins, monkeypatch):
    """main() without -ch (mustache.py:1019-1033): `.cool` / `.mcool` keep the chromosomes above 1 Mb, `.hic` every one
    after the leading pseudo-chromosome, with the sizes it hands to read_hic_file."""
    from mustache_amd.readers import list_chromosomes, chromosome_sizes
    cool, mcool, _ = standins
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "hicstraw")
    assert list_chromosomes(cool, rc.RES) == [str(s) for s in GOLD["main_cool_chroms"]] == ["chrA", "chrB"]
    assert list_chromosomes(mcool, 2 * rc.RES) == [str(s) for s in GOLD["main_mcool_chroms"]]
    names = list_chromosomes("case.hic", rc.RES)
    assert names == [str(s) for s in GOLD["main_hic_chroms"]] == ["chrA", "chrB", "chrC"]
    sizes = chromosome_sizes("case.hic", rc.RES)
    assert [sizes[n] for n in names] == GOLD["main_hic_sizes"].tolist()
    assert GOLD["main_cool_sizes"].tolist() == [0, 0]                 # no size is passed for cooler files
    assert GOLD["main_cool_dist_res"].tolist() == [rc.DIST, rc.RES]


def test_whole_genome_gpu_test_container_equals_the_reference(standins, tmp_path):
    """The container tests/test_gpu_pipeline.py::test_whole_genome_cool_path_two_ranks_equals_oracle runs through the GPU
    pipeline: read_cooler returns the record sets the reference's read_cooler returned (count + sha256 in the fixture)."""
    from mustache_amd.readers import read_cooler
    gcool = str(tmp_path / "genome.cool")
    rc.standin_cooler().write_cool(gcool, rc.GENOME_RES, rc.genome_container())
    for name, _, _ in rc.GENOME:
        x, y, v, res = read_cooler(gcool, rc.GENOME_DPX * rc.GENOME_RES, name, name, False)
        assert len(x) == int(GOLD["genome_%s_count" % name]) > 100000
        assert rc.digest(x, y, v) == str(GOLD["genome_%s_sha256" % name])


@pytest.mark.parametrize("version", [8, 9])
def test_packed_hic_records_equal_the_reference(tmp_path, monkeypatch, version):
    """The packed form the GPU loader takes (mst_hic_read_intra_packed: int32 bin, int32 distance, float32 value) holds the
    record set the reference's read_hic_file returns, and n = max bin + 1 (mustache.py:894)."""
    from mustache_amd.readers import read_hic_packed
    p = str(tmp_path / "case.hic")
    write_case_hic(p, version)
    for name, size in rc.CHROMS:
        pc = read_hic_packed(p, False, False, rc.DIST, name, rc.RES)
        assert pc.x.dtype == np.int32 and pc.dist.dtype == np.int32 and pc.v.dtype == np.float32 and pc.res == rc.RES
        x, y, v = pc.coo()
        same((x, y, v), "hic_" + name)
        assert pc.n == int(expect("hic_" + name)[1].max()) + 1
    # a caller-supplied chromosome size cuts the last window like straw does: nothing at or past it
    pc = read_hic_packed(p, "KR", 20_000_000, rc.DIST, "chrA", rc.RES)
    x, y, v = pc.coo()
    ex, ey, ev = expect("hic_chrA")
    keep = ey * rc.RES < 20_000_000
    gx, gy, gv = rc.as_sorted(x, y, v)
    assert np.array_equal(gx, ex[keep]) and np.array_equal(gy, ey[keep]) and np.array_equal(gv, ev[keep])
    assert pc.n == int(ey[keep].max()) + 1
    with pytest.raises(NameError):
        read_hic_packed(p, False, False, rc.DIST, "chrQ", rc.RES)
