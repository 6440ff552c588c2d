"""oracle/track_oracle.c against the reference's own outputs for rr_hctg_track.py (canonical line order): CPU only."""
import pytest

from tests import golden_ovlp_util as G
from tests import oracle_lib


@pytest.mark.parametrize("name", G.track_cases())
def test_oracle_matches_reference(oracle, name):
    c = G.load_track(name)
    out = oracle_lib.track_reads(oracle, c["files"], c["phased_reads"], c["read_to_contig_map"], c["rawread_ids"], c["params"]["min_len"], c["params"]["bestn"])
    assert out == c["expected"]
    assert out.count(b"\n") > 50
