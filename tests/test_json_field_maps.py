"""The web JSON column lists (names and order) the engine emits are the reference's own: parsed here from
/root/reference/common/gy_json_field_maps.h (json_db_svcstate_arr :1102, json_db_svcsumm_arr :1396, json_db_clusterstate_arr :2162)
and compared with (i) the lists the GPU JSON test expects and (ii) the field names gyeeta_amd/csrc/gys_json.hpp writes, in order.
Runs where the reference tree is mounted (this container); skipped on the GPU box."""
import os
import re

import pytest

REF = "/root/reference/common/gy_json_field_maps.h"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref_columns(array_name):
    src = open(REF).read()
    m = re.search(r"static constexpr JSON_DB_MAPPING\s+" + array_name + r"\[\]\s*=\s*\{(.*?)\n\};", src, re.S)
    assert m, array_name
    return re.findall(r'^\{\s*"([^"]+)"\s*,', m.group(1), re.M)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted")
def test_column_lists_equal_the_reference_field_maps():
    from tests import test_gpu_json as tj
    assert _ref_columns("json_db_svcstate_arr") == tj.SVCSTATE_COLS
    assert _ref_columns("json_db_svcsumm_arr") == tj.SVCSUMM_COLS
    assert _ref_columns("json_db_clusterstate_arr") == tj.CLUSTER_COLS


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted")
def test_emitter_source_names_every_reference_column():
    src = open(os.path.join(ROOT, "gyeeta_amd", "csrc", "gys_json.hpp")).read()
    for arr in ("json_db_svcstate_arr", "json_db_svcsumm_arr", "json_db_clusterstate_arr"):
        cols = _ref_columns(arr)
        for c in cols:  # every column name is a key literal of the emitter (the order is checked on the emitted JSON by tests/test_gpu_json.py)
            assert '"' + c + '"' in src, (arr, c)
        assert len(cols) > 5
