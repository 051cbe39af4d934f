"""The selection-language subset (groan_rs_amd/select.py) against the reference's own known answers on example.gro
(System::group_create tests, src/system/groups.rs:1095-2110): atom counts and member indices of keyword queries, macros,
open-ended ranges, regular expressions, group references, and the error variants of malformed queries.  The names / numbers
of example.gro are the fixture tests/golden/example_names.npz (tests/golden/make_names_fixture.py)."""
import os
import types

import numpy as np
import pytest

from groan_rs_amd.select import SelectError, parse_query, select

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ex():
    d = np.load(os.path.join(HERE, "golden", "example_names.npz"))
    s = types.SimpleNamespace(n_atoms=int(d["resid"].size), resid=d["resid"], atomid=d["atomid"],
                              resname=[x.decode() for x in d["resname"]], atomname=[x.decode() for x in d["atomname"]])
    g = np.load(os.path.join(HERE, "golden", "example.npz"))
    groups = {}
    for k in g.files:
        if k.startswith("blocks_"):
            groups[k[7:]] = np.concatenate([np.zeros(0, np.int64)] + [np.arange(a, b + 1) for a, b in g[k]]).astype(np.int64)
    # further groups of the reference's index.ndx that the regular-expression tests match (same atoms as in that file)
    groups["POPC"] = np.arange(61, 6205); groups["Protein_Membrane"] = np.arange(0, 6205); groups["Protein-H"] = np.arange(0, 61)
    return s, groups


def rng(a, b):
    return np.arange(a, b + 1, dtype=np.uint64)


@pytest.mark.parametrize("query,count,members", [
    ("resname POPC", 6144, rng(61, 6204)),                       # groups.rs:1100-1107
    ("serial 1 to 61", 61, rng(0, 60)),                          # :1109-1116
    ("@protein", 61, rng(0, 60)),                                # :1370-1378
    ("@membrane", 6144, rng(61, 6204)),                          # :1384-1392
    ("@water", 10399, rng(6205, 16603)),                         # :1412-1420
    ("@ion", 240, rng(16604, 16843)),                            # :1438-1446
    ("resid < 380", 4261, rng(0, 4260)),                         # :1931-1938
    ("resid <= 380", 4273, rng(0, 4272)),                        # :1940-1946
    ("serial > 9143", 7701, rng(9143, 16843)),                   # :1948-1954
    ("serial >= 9143", 7702, rng(9142, 16843)),                  # :1956-1962
    ("serial <= 10000 10005-10010", 10006, np.concatenate([rng(0, 9999), rng(10004, 10009)])),   # :1964-1978
])
def test_reference_counts_and_members(ex, query, count, members):
    s, groups = ex
    got = select(s, query, groups)
    assert got.size == count
    assert np.array_equal(got, members)


def test_regular_expressions(ex):
    s, groups = ex
    got = select(s, "resname r'^[LA].*'", groups)                                        # :1984-1991
    assert got.size == 36 and 1 in got and 58 in got
    got = select(s, "resname POPC and name r'^[CD][124][AB]'", groups)                   # :1993-2001
    assert got.size == 3072 and 65 in got and 6204 in got
    got = select(s, "resname r'^..PC' r'L'", groups)                                     # :2003-2011
    assert got.size == 6203 and 0 in got and 6204 in got
    got = select(s, "resname POPC and (name r'C[1234]A|C[1234]B' or name D2A)", groups)  # :2016-2035
    assert got.size == 4096 and all(i in got for i in (78, 79, 80, 81))


def test_group_references(ex):
    s, groups = ex
    assert np.array_equal(select(s, "Protein", groups), rng(0, 60))
    got = select(s, "r'^Transmembrane'", groups)                                         # :2054-2059
    assert got.size == 61 and 0 in got and 60 in got and 61 not in got
    got = select(s, "r'^Transmembrane$'", groups)                                        # :2062-2067
    assert got.size == 29 and 0 in got and 59 in got and 60 not in got
    for q in ("group r'^P' ION", "group r'^P' r'^X' ION"):                               # :2070-2089
        got = select(s, q, groups)
        assert got.size == 6445 and 0 in got and 16843 in got and 16603 not in got and 6205 not in got
    with pytest.raises(SelectError) as e:                                                # :2092-2095
        select(s, "group r'X'", groups)
    assert e.value.variant == "NoRegexMatch" and e.value.detail == "X"
    with pytest.raises(SelectError) as e:                                                # :1152-1156
        select(s, "Protein", {})
    assert e.value.variant == "GroupNotFound"
    got = select(s, "Membrane or (Protein and not serial 1 to 10)", groups)
    assert got.size == 6144 + 51
    assert np.array_equal(select(s, "!Membrane && !W && !ION", groups), rng(0, 60))


def test_operators_associate_to_the_left_with_equal_precedence(ex):
    s, groups = ex
    a = select(s, "Protein or Membrane and resname LYS", groups)          # (Protein or Membrane) and resname LYS
    b = select(s, "(Protein or Membrane) and resname LYS", groups)
    c = select(s, "Protein or (Membrane and resname LYS)", groups)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(select(s, "atomnum 1 to 61", groups), select(s, "serial 1 - 61", groups))
    assert np.array_equal(select(s, "name 'BB' \"SC1\" and resid 1 to 3", groups), select(s, "atomname BB SC1 && resnum 1-3", groups))


@pytest.mark.parametrize("query,variant", [
    ("", "EmptyQuery"), ("   ", "EmptyQuery"),
    ("resname POPC &&", "MissingArgument"),                               # :1134-1139 "missing argument"
    ("(resname POPC && resname POPE))", "InvalidParentheses"),            # :1144-1149 "unmatching parentheses"
    ("resname POPC & name P", "InvalidOperator"),
    ("resname", "EmptyArgument"), ("serial", "EmptyArgument"),
    ("resid 5 to x", "InvalidNumber"), ("serial 1 2 - ", "InvalidNumber"),
    ("resname 'POPC", "InvalidQuotes"),
    ("resname r'['", "InvalidRegex"),
    ("atomid 5", "DeprecatedKeyword"),
    ("chain A", "Unsupported"), ("molecule with serial 5", "Unsupported"),
])
def test_malformed_queries(ex, query, variant):
    s, groups = ex
    with pytest.raises(SelectError) as e:
        select(s, query, groups)
    assert e.value.variant == variant
