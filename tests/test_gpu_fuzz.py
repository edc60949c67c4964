"""A seeded slice of the randomised sweeps (scripts/fuzz_parity.py, fuzz_pipeline.py, fuzz_genome.py, fuzz_diff.py,
large_geometry_check.py) under the driver's `pytest -m gpu`: the same case generators (tests/fuzz_cases.py), fixed seeds, sized
to stay within a few minutes (the CPU oracle is the slow side).  The pipeline cases run in BOTH tile-sharing modes (tiles of
overlapping blocks computed once / every tile once per block); all of them run with the tile list (skip_empty, the product
mode), the block cases additionally with every tile launched."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cases      # noqa: E402

pytestmark = pytest.mark.gpu


def _chromosome_cases(pipe, plan=None):
    rng = np.random.default_rng(20265)
    res = [fuzz_cases.pipeline_case(rng, pipe, max_n=3600, share_modes=(True, False), plan=plan) for _ in range(12)]
    # three cases at the 1 kb geometry in small: blocks overlap by half their edge, most tiles are shared
    res += [fuzz_cases.pipeline_case(rng, pipe, share_modes=(True, False), wide=True, plan=plan) for _ in range(3)]
    return res


def _genome_cases(pipe, plan=None):
    rng = np.random.default_rng(20266)
    return [fuzz_cases.genome_case(rng, pipe, max_n=3200, max_chroms=4, plan=plan) for _ in range(7)]


def _geometry_case(pipe, plan=None):
    return fuzz_cases.geometry_case(pipe, 7000, 3011, 1000, 300.0, share_modes=(True, False), plan=plan)


@pytest.fixture(scope="module")
def pipe():
    """the pipeline the cases share -- and the CPU oracle of every pipeline case of this module started AHEAD in worker
    processes (fuzz_cases.start_ahead: the cases are first drawn in plan mode, same seeds), largest job first: the oracle is the
    slow side (the module took 5 of the GPU suite's 10 minutes with the oracle inline)"""
    from mustache_amd.pipeline import ChromosomePipeline
    plan = []
    _geometry_case(None, plan)
    _chromosome_cases(None, plan)
    _genome_cases(None, plan)
    fuzz_cases.start_ahead(sorted(plan, key=lambda j: -(j[0] * max(j[1], 400))))
    yield ChromosomePipeline(fuzz_cases.OCT)
    fuzz_cases.stop_ahead()


def _report(results):
    bad = [d for ok, _, d in results if not ok]
    assert not bad, bad
    return sum(n for _, n, _ in results)


def test_fuzz_blocks_seeded_slice(pipe):
    rng = np.random.default_rng(20264)
    res = [fuzz_cases.parity_case(rng, pipe.engine) for _ in range(24)]
    loops = _report(res)
    # (small blocks call few loops; what these cases hold is the complete found set: pixels, levels, DoG values, loc, p)
    assert loops >= 5 and sum(d.get("found", 0) for _, _, d in res) > 2000


def test_fuzz_chromosomes_seeded_slice_both_share_modes(pipe):
    res = _chromosome_cases(pipe)
    loops = _report(res)
    assert loops > 100 and {d["branch"] for _, _, d in res} == {"A", "B"}


def test_fuzz_genomes_seeded_slice(pipe):
    loops = _report(_genome_cases(pipe))
    assert loops > 100


def test_fuzz_block_pairs_seeded_slice(pipe):
    rng = np.random.default_rng(20267)
    loops = _report([fuzz_cases.diff_case(rng, pipe.engine) for _ in range(14)])
    assert loops > 100


def test_large_geometry_odd_distance_limit(pipe):
    """dpx 3011 -> blocks of 6022 x 6022 whose overlap is not a multiple of anything (tile lattice phase differs per block),
    both share modes."""
    ok, loops, d = _geometry_case(pipe)
    assert ok and loops >= 2, d
