"""-m gpu: BASELINE config 5 at ITS size — one rank's real share of the 8-way partition of the papers100M-sized
graph (N = 111 M, 3.34 G directed edges incl. loops), built on one MI355X by the per-rank generator without a global
edge list and run through the product path as a dry partition (send lists and buffers exactly as in the 8-rank run,
nothing on the wire).  Checked through size-independent properties: edge share, 64-bit edge counts, halo / send-list
identities (the graph is symmetric: what a peer needs from me mirrors what I need from it), the aggregate against an
f64 column checksum and against f64 row evaluations.  ~25 s, 95 GB peak; skipped below 200 GB of HBM."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_papers100m_sized_rank_share_of_the_8_way_partition():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; the HIP path has no fallback")
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(dev).total_memory < 200 * 2**30:
        pytest.skip("needs > 200 GB of HBM")
    from gammagl_amd import engine
    from gammagl_amd.synth import DATASETS
    from share_checks import check_rank_share

    eng = engine()
    eng.clear_caches()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    n, e, _, _ = DATASETS["papers100M"]
    assert e + n > 2**31                                   # the global edge count does not fit 32 bits
    check_rank_share(eng, dev, n, e, P=8, r=3, min_buckets=8, mem_limit=200 * 2**30)
    torch.cuda.empty_cache()
