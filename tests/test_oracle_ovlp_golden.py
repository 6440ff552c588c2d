"""oracle/ovlp_oracle.c against the reference's own outputs (tests/golden_ovlp/): CPU only."""
import pytest

from tests import golden_ovlp_util as G
from tests import oracle_lib


@pytest.fixture(scope="module")
def orc():
    return oracle_lib.load()


@pytest.mark.parametrize("name", G.cases())
def test_oracle_matches_reference(orc, name):
    c = G.load(name)
    out, ignore, contained = oracle_lib.ovlp_filter(orc, c["files"], c["rid_map"], c["params"])
    assert sorted(ignore) == c["ignore"]
    assert sorted(contained) == c["contained"]
    assert out == c["expected"]


def test_oracle_reports_what_the_reference_would_raise(orc):
    c = G.load("o3_quirks")
    bad = c["files"][0] + b"000000001\n"           # one token: `q_id, t_id = l[:2]` raises
    with pytest.raises(oracle_lib.OracleError):
        oracle_lib.ovlp_filter(orc, [bad], c["rid_map"], c["params"])
    bad = b"000000001 000000002 -5000 abc 0 0 5000 9000 0 3000 8000 8000 overlap\n"   # float('abc')
    with pytest.raises(oracle_lib.OracleError):
        oracle_lib.ovlp_filter(orc, [bad], c["rid_map"], c["params"])
    ok = b"000000099 000000002 -5000 abc\n"          # never parsed: q is not in the map
    out, _, _ = oracle_lib.ovlp_filter(orc, [ok], c["rid_map"], c["params"])
    assert out == b""
