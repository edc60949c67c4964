"""Native text contact-map parser (libmustache_io.so: mst_text_read_contacts) pinned against pandas itself -- the
reference's read_pd() parses with pd.read_csv(f, sep=sep, header=None) + dropna() (reference mustache/mustache.py:254-258),
and pandas is importable here, so every returned double is compared bit for bit with what pandas yields."""
import os
import random

import numpy as np
import pandas as pd
import pytest


def _token(rng):
    k = rng.random()
    if k < 0.30:
        return str(rng.randint(0, 10 ** rng.randint(1, 12)))
    if k < 0.55:
        return "%d.%s" % (rng.randint(0, 10 ** rng.randint(0, 9)), "".join(rng.choice("0123456789") for _ in range(rng.randint(1, 22))))
    if k < 0.78:
        return repr(rng.uniform(0, 1) * 10 ** rng.randint(-8, 8))
    if k < 0.88:
        return "%.*e" % (rng.randint(0, 19), rng.uniform(0, 1) * 10 ** rng.randint(-320, 300))
    if k < 0.94:
        return rng.choice(["NaN", "nan", "NA", "", "null", "N/A", "inf", "-inf", "#N/A", "None", "<NA>"])
    return "%s%.*E" % (rng.choice(["", "-", "+"]), rng.randint(0, 6), rng.uniform(0, 1) * 10 ** rng.randint(-30, 30))


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


@pytest.mark.parametrize("sep,eol,final_newline,seed", [("\t", "\n", True, 1), (" ", "\n", False, 2), (",", "\r\n", True, 3),
                                                        ("\t", "\r\n", False, 4)])
def test_three_columns_equal_pandas_bitwise(tmp_path, sep, eol, final_newline, seed):
    from mustache_amd.hicfile import read_text_contacts
    rng = random.Random(seed)
    lines = [sep.join(["0", "5000", "1.5"])]
    for _ in range(60000):
        r = rng.random()
        if r < 0.01:
            lines.append("")                                    # blank line: skipped
        elif r < 0.02:
            lines.append(sep.join([_token(rng), _token(rng)]))  # short line: missing field is NaN, row dropped
        else:
            lines.append(sep.join([str(rng.randint(0, 250000) * 1000),
                                   _token(rng) if rng.random() < 0.3 else str(rng.randint(0, 250000) * 1000), _token(rng)]))
    p = str(tmp_path / "c.txt")
    with open(p, "w", newline="") as fh:
        fh.write(eol.join(lines) + (eol if final_newline else ""))
    df = pd.read_csv(p, sep=sep, header=None).dropna()
    for threads in (1, 5):
        nc, a, b, c = read_text_contacts(p, sep, threads=threads)
        assert nc == 3 and len(a) == len(df) > 50000
        for k, arr in zip((0, 1, 2), (a, b, c)):
            assert np.array_equal(_bits(df[k]), _bits(arr)), k


def test_five_columns_and_chromosome_filter(tmp_path):
    from mustache_amd.hicfile import read_text_contacts
    from mustache_amd.mustache import is_chr
    rng = random.Random(9)
    names = ["chr1", "1", "chr2", "chr11", "X", "chrX", "NA"]
    lines = []
    for _ in range(30000):
        lines.append("\t".join([rng.choice(names), str(rng.randint(0, 9000) * 5000), rng.choice(names),
                                str(rng.randint(0, 9000) * 5000), _token(rng)]))
    p = str(tmp_path / "c5.txt")
    open(p, "w").write("\n".join(lines) + "\n")
    df = pd.read_csv(p, sep="\t", header=None).dropna()
    for chrom in ("1", "chr1", "X", "chr11", "7"):
        sub = df[np.vectorize(is_chr)(df[0], chrom)]
        sub = sub[np.vectorize(is_chr)(sub[2], chrom)] if len(sub) else sub
        nc, a, b, c = read_text_contacts(p, "\t", chrom)
        assert nc == 5 and len(a) == len(sub)
        if len(sub):
            for k, arr in zip((1, 3, 4), (a, b, c)):
                assert np.array_equal(_bits(sub[k]), _bits(arr)), (chrom, k)
    assert len(read_text_contacts(p, "\t", "1")[1]) > 500


def test_uncovered_constructs_are_reported_not_guessed(tmp_path):
    from mustache_amd.hicfile import HicError, read_text_contacts
    cases = {"quoted": '0\t5000\t"3"\n', "extra_field": "0\t5000\t3\n1\t2\t3\t4\n", "word": "0\t5000\tthree\n",
             "four_columns": "0\t5000\t3\t9\n", "empty": ""}
    for name, text in cases.items():
        p = str(tmp_path / (name + ".txt"))
        open(p, "w").write(text)
        with pytest.raises(HicError) as e:
            read_text_contacts(p, "\t")
        assert e.value.code == -3, name
    with pytest.raises(HicError) as e:
        read_text_contacts(str(tmp_path / "missing.txt"), "\t")
    assert e.value.code == -2


def test_read_pd_native_equals_pandas_backend(tmp_path, monkeypatch):
    """read_pd() through the native parser == read_pd() through pandas (the reference's own code path), with a bias file
    (NaN and < 0.2 factors), the distance filter and the > 0 filter in between; an uncovered file falls back to pandas."""
    from mustache_amd.mustache import read_pd
    rng = np.random.default_rng(5)
    n, res, dist = 3000, 5000, 2_000_000
    x = rng.integers(0, n, 40000)
    y = np.minimum(x + rng.integers(0, 460, 40000), n - 1)
    v = np.round(rng.uniform(0, 300, 40000), 3)
    v[::50] = 0.0
    f = str(tmp_path / "c.RAWobserved")
    with open(f, "w") as fh:
        for a, b, c in zip(x, y, v):
            fh.write("%d\t%d\t%r\n" % (a * res, b * res, float(c)))
    bias = rng.uniform(0.1, 2.0, n)
    bias[::37] = np.nan
    bf = str(tmp_path / "c.KRnorm")
    open(bf, "w").write("\n".join("NaN" if np.isnan(t) else repr(float(t)) for t in bias) + "\n")
    out = {}
    for backend in ("native", "pandas"):
        monkeypatch.setenv("MUSTACHE_TEXT_BACKEND", backend)
        out[backend] = [np.asarray(a) for a in read_pd(f, dist, bf, "1", res)]
    assert len(out["native"][2]) > 20000
    for a, b in zip(out["native"], out["pandas"]):
        assert np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))
    monkeypatch.setenv("MUSTACHE_TEXT_BACKEND", "native")
    q = str(tmp_path / "q.txt")
    open(q, "w").write('0\t5000\t"3"\n5000\t10000\t4\n')          # quotes: the native parser declines, pandas reads it
    xq, yq, vq = read_pd(q, dist, False, "1", res)
    assert list(np.asarray(xq)) == [0, 1] and list(np.asarray(vq)) == [3, 4]
