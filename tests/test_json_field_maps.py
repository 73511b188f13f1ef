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


def _ref_enum(path, name):
    """members of `enum <name> ... { ... }` in a reference header, in order, with their values (plain enumerators count up from the last
    explicit value)"""
    src = open(path).read()
    m = re.search(r"enum\s+" + name + r"\b[^{]*\{(.*?)\};", src, re.S)
    assert m, name
    out, nxt = {}, 0
    for line in m.group(1).splitlines():
        line = line.split("//")[0].strip().rstrip(",")
        if not line:
            continue
        mm = re.match(r"(\w+)\s*(?:=\s*(\w+))?$", line)
        if not mm:
            continue
        if mm.group(2) is not None:
            nxt = int(mm.group(2), 0)
        out[mm.group(1)] = nxt
        nxt += 1
    return out


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted")
def test_filter_query_tables_equal_the_reference():
    """the filtered multi-host query (gys_query_svcstate_scan / _aggr): its column list is the numeric / bool / state part of
    json_db_svcstate_arr in the reference's order, its comparator and aggregation-operator numbers are COMPARATORS_E / AGGR_OPER_E"""
    from gyeeta_amd import capi
    src = open(REF).read()
    m = re.search(r"static constexpr JSON_DB_MAPPING\s+json_db_svcstate_arr\[\]\s*=\s*\{(.*?)\n\};", src, re.S)
    rows = re.findall(r'^\{\s*"([^"]*)"\s*,\s*"([^"]+)"\s*,[^,]+,\s*(\w+)\s*,\s*\w+\s*,\s*(\w+)\s*,\s*(\w+)\s*,', m.group(1), re.M)
    want = [name for name, db, crc, jtype, ntype in rows if name and (jtype in ("JSON_NUMBER", "JSON_BOOL") or name == "state")]
    # the header lists state / issue / ishttp in this order after the NUM_INT32 block; the engine's enum keeps the reference's order
    assert want == capi.SVC_COLS, (want, capi.SVC_COLS)
    hdr = open(os.path.join(ROOT, "include", "gysketch.h")).read()
    enum_cols = re.search(r"enum \{ GYS_SVC_COL_QPS5S = 0,(.*?)GYS_SVC_NCOLS \};", hdr, re.S).group(0)
    names = re.findall(r"GYS_SVC_COL_(\w+)", enum_cols)
    assert [n.lower() for n in names] == capi.SVC_COLS
    comp = _ref_enum("/root/reference/common/gy_query_criteria.h", "COMPARATORS_E")
    ref_comp = {"=": "COMP_EQ", "!=": "COMP_NEQ", "<": "COMP_LT", "<=": "COMP_LE", ">": "COMP_GT", ">=": "COMP_GE", "bit2": "COMP_BIT2", "bit3": "COMP_BIT3",
                "substr": "COMP_SUBSTR", "notsubstr": "COMP_NOTSUBSTR", "like": "COMP_LIKE", "notlike": "COMP_NOTLIKE", "in": "COMP_IN", "notin": "COMP_NOTIN"}
    for k, v in capi.COMP.items():
        assert comp[ref_comp[k]] == v, k
    for cname, v in comp.items():  # ... and every GYS_COMP_* of the header carries the reference's number
        if cname == "COMP_MAX":
            continue
        mm = re.search(r"GYS_" + cname + r"\b(?:\s*=\s*(\d+))?", hdr)
        assert mm, cname
        if mm.group(1) is not None:
            assert int(mm.group(1)) == v, cname
    aop = _ref_enum(REF, "AGGR_OPER_E")
    ref_aop = {"sum": "AOPER_SUM", "avg": "AOPER_AVG", "max": "AOPER_MAX", "min": "AOPER_MIN", "count": "AOPER_COUNT", "bool_or": "AOPER_BOOL_OR",
               "bool_and": "AOPER_BOOL_AND"}
    for k, v in capi.AOPER.items():
        assert aop[ref_aop[k]] == v, k
    for k, v in capi.AOPER.items():  # ... and the header's GYS_AOPER_* carry the same numbers
        mm = re.search(r"GYS_AOPER_" + k.upper() + r"(?:\s*=\s*(\d+))?", hdr)
        assert mm, k


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted")
def test_summary_query_columns_equal_the_reference():
    """gys_json_svcsumm_multihost: GYS_SUMM_COL_* are the numeric columns of json_db_svcsumm_arr in the reference's order"""
    from gyeeta_amd import capi
    assert _ref_columns("json_db_svcsumm_arr") == ["time"] + capi.SUMM_COLS
    hdr = open(os.path.join(ROOT, "include", "gysketch.h")).read()
    names = re.findall(r"GYS_SUMM_COL_(\w+)", re.search(r"enum \{ GYS_SUMM_COL_NIDLE = 0,(.*?)GYS_SUMM_NCOLS \};", hdr, re.S).group(0))
    assert [n.lower() for n in names] == capi.SUMM_COLS
