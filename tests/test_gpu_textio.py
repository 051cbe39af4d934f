"""System::from_file(.gro) + System::read_ndx on the device mirror (groan_rs_amd.textio over gr_gro_read / gr_ndx_install):
the group bookkeeping and the warnings of src/io/ndx_io.rs:31-98 with the known answers of its tests (:546-700), and a first
analysis straight from files."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
T = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textio")
f = lambda name: os.path.join(T, name)


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def fresh(G):
    return G.system_from_gro(f("example_novelocities.gro"))


def test_from_file_and_read_ndx(G):
    s = fresh(G)
    assert s.get_n_atoms() == 50 and s.group_exists("all") and s.group_exists("All")     # structure.rs:546-558
    G.system_read_ndx(s, f("index_small.ndx"))                                             # ndx_io.rs:334-363
    assert s.group_get_n_atoms("System") == 50 and s.group_get_n_atoms("Protein") == 50
    s.close()
    s = fresh(G); G.system_read_ndx(s, f("index_empty.ndx"))                               # :428-437
    assert not s.group_exists("System") and s.group_exists("all")
    s.close()


@pytest.mark.parametrize("name,dups,sizes", [                                                # ndx_io.rs:546-621
    ("index_duplicate_groups.ndx", {"Protein"}, {"System": 50, "Protein": 32}),
    ("index_duplicate_groups2.ndx", {"Protein"}, {"System": 50, "Protein": 15}),
    ("index_group_exists.ndx", {"All"}, {"System": 50, "Protein": 50, "All": 35}),
    ("index_groups_exist.ndx", {"All", "Protein"}, {"System": 50, "Protein": 15, "All": 35}),
])
def test_duplicate_group_warnings(G, name, dups, sizes):
    s = fresh(G)
    with pytest.raises(G.ParseNdxError) as e:
        G.system_read_ndx(s, f(name))
    assert e.value.variant == "DuplicateGroupsWarning" and e.value.detail == dups
    for g, n in sizes.items():
        assert s.group_get_n_atoms(g) == n
    s.close()


def test_invalid_names_warning_and_failed_reads_leave_the_system_alone(G):
    s = fresh(G)
    with pytest.raises(G.ParseNdxError) as e:                                               # :623-645
        G.system_read_ndx(s, f("index_invalid_names.ndx"))
    assert e.value.variant == "InvalidNamesWarning" and e.value.detail == {"inval@id", "&also_invalid", "(parentheses are invalid)"}
    assert s.group_get_n_atoms("System") == 50 and s.group_exists("Valid Name") and not s.group_exists("inval@id")
    s.close()
    for bad in ("index_invalid_name.ndx", "index_unfinished_name.ndx", "index_invalid_line.ndx", "index_invalid_index1.ndx", "index_invalid_index2.ndx", "nonexistent.ndx"):
        s = fresh(G)                                                                        # read_ndx_fails! :491-506
        with pytest.raises(G.ParseNdxError):
            G.system_read_ndx(s, f(bad))
        assert not s.group_exists("System") and not s.group_exists("Protein") and s.group_exists("all") and s.group_exists("All")
        s.close()


def test_analysis_straight_from_files(G):
    s = G.system_from_gro(f("triclinic.gro"))
    st = s.structure
    idx = st.indices_where(name="BB")
    s.group_create_from_indices("Backbone", idx)
    np.testing.assert_allclose(s.group_estimate_center("Backbone"), O.estimate_center(st.positions, idx, st.box9), atol=1e-5, rtol=0)
    np.testing.assert_allclose(s.group_get_center("all"), O.get_center(st.positions, np.arange(st.n_atoms), st.box9), atol=1e-5, rtol=0)
    s.close()


def test_group_create_from_a_selection_query(G):
    """System::group_create (groups.rs:36-92) on the device mirror: query -> indices (groan_rs_amd/select.py) -> device group;
    existing groups can be referenced, an overwritten group answers True (the reference's AlreadyExistsWarning)"""
    s = fresh(G)
    st = G.Structure(f("example_novelocities.gro"))
    G.system_read_ndx(s, f("index_small.ndx"))
    assert s.group_create("Sel", "serial 1 to 10 or resid 3", st) is False
    want = np.union1d(st.indices_where(serial=(1, 10)), st.indices_where(resid=3))
    assert np.array_equal(np.array(list(s.group_container("Sel")), np.uint64), want) and want.size >= 10
    assert s.group_create("Sel", "Protein and not Sel", st) is True                    # references the OLD Sel, then replaces it
    assert s.group_get_n_atoms("Sel") == 50 - want.size
    names = sorted(set(st.atomname))[:2]
    s.group_create("ByName", "name %s %s and System" % tuple(names), st)
    assert np.array_equal(np.array(list(s.group_container("ByName")), np.uint64), st.indices_where(name=names))
    assert "Sel" in s.group_names() and "ByName" in s.group_names() and "all" in s.group_names()
    np.testing.assert_allclose(s.group_get_center_naive("ByName"), st.positions[st.indices_where(name=names).astype(int)].mean(0), atol=1e-5)
    with pytest.raises(G.SelectError) as e:
        s.group_create("Bad", "resname", st)
    assert e.value.variant == "EmptyArgument" and not s.group_exists("Bad")
    with pytest.raises(G.SelectError) as e:
        s.group_create("Bad", "NoSuchGroup", st)
    assert e.value.variant == "GroupNotFound"
    s.close()
