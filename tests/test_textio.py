"""gro / ndx readers (groan_rs_amd/csrc/gr_textio.h through gr_gro_read / gr_ndx_read; host code, no GPU) against the known
answers of the reference's own reader tests -- src/io/gro_io/structure.rs:240-582 and src/io/ndx_io.rs:238-700 -- on the
reference's small data files (tests/golden/textio/)."""
import os

import numpy as np
import pytest

import groan_rs_amd as G

T = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textio")
f = lambda name: os.path.join(T, name)


def test_read_novelocities():                                   # structure.rs:328-385
    s = G.Structure(f("example_novelocities.gro"))
    assert s.title == "Buforin II peptide P11L" and s.n_atoms == 50
    assert np.allclose(s.box9[:3], 6.08608) and np.all(s.box9[3:] == 0.0)
    first, mid, last = 0, 24, 49
    assert (s.resid[first], s.resname[first], s.atomname[first], s.atomid[first]) == (1, "THR", "BB", 1)
    assert np.allclose(s.positions[first], [1.660, 2.061, 3.153])
    assert (s.resid[mid], s.resname[mid], s.atomname[mid], s.atomid[mid]) == (11, "LEU", "SC1", 25)
    assert np.allclose(s.positions[mid], [3.161, 2.868, 2.797])
    assert (s.resid[last], s.resname[last], s.atomname[last], s.atomid[last]) == (21, "LYS", "SC2", 50)
    assert np.allclose(s.positions[last], [4.706, 4.447, 2.813])
    assert np.isnan(s.velocities).all() and not s.has_velocities()


def test_read_box9_and_zero_box_and_velocities():               # :391-413
    b = G.Structure(f("example_box9.gro")).box9
    assert np.allclose(b, [6.08608, 6.08608, 6.08608, 0.0, 0.0, 2.2, 0.0, 1.4, 3.856])
    assert G.Structure(f("example_box_zero.gro")).box9 is None
    t = G.Structure(f("triclinic.gro"))                          # a 50-atom frame with velocities and a 9-value box
    assert t.n_atoms == 50 and t.box9 is not None and t.box9[5] != 0.0


@pytest.mark.parametrize("name,variant,payload", [                # structure.rs:434-580
    ("example_incomplete_line.gro", "ParseAtomLineErr", "   16HIS    SC1   35   3.458   3.653   "),
    ("example_empty.gro", "LineNotFound", None),
    ("example_only_title.gro", "LineNotFound", None),
    ("example_missing_natoms.gro", "ParseLineErr", ""),
    ("example_unparsable_natoms.gro", "ParseLineErr", "6A5F"),
    ("example_missing_atom.gro", "ParseAtomLineErr", "   6.08608   6.08608   6.08608"),
    ("example_invalid_resid.gro", "ParseAtomLineErr", "   1APHE    SC2   22   2.519   3.025   3.387"),
    ("example_invalid_atomid.gro", "ParseAtomLineErr", "   21LYS     BB        4.362   4.008   3.161"),
    ("example_invalid_position.gro", "ParseAtomLineErr", "    2ARG    SC1    4   1.877   1. 73   3.023"),
    ("example_invalid_velocity.gro", "ParseAtomLineErr", "   15LEU    SC1   31   9.638   2.052   5.595  0.0685  O.0634  0.1453"),
    ("example_shifted_line.gro", "ParseAtomLineErr", "    20ARG     BB   45   4.265   3.832   2.925"),
    ("example_empty_box_line.gro", "ParseBoxLineErr", ""),
    ("example_short_box.gro", "ParseBoxLineErr", "   6.08608   6.08608"),
    ("example_long_box.gro", "ParseBoxLineErr", "   6.08608   6.08608   6.08608   6.08608   6.08608"),
    ("example_unparsable_box.gro", "ParseBoxLineErr", "   6.08608   6.08608   6,08608"),
    ("example_unsupported_box.gro", "UnsupportedBox", "   6.08608   6.08608   6.08608   0.00000   0.60000   2.20000   0.00000   1.40000   3.85600"),
    ("nan_error.gro", "InvalidFloat", "   19ALA    SC1   39     nan   2.496   5.027  0.0733 -0.2227 -0.2563"),
    ("nan_error_velocity.gro", "InvalidFloat", "    6VAL     BB   12   9.947   2.258   6.831 -0.2096     NaN  0.0665"),
    ("nonexistent.gro", "FileNotFound", None),
])
def test_read_gro_fails(name, variant, payload):
    with pytest.raises(G.ParseGroError) as e:
        G.Structure(f(name))
    assert e.value.variant == variant
    if payload is not None:
        assert e.value.detail == payload
    else:
        assert e.value.detail.endswith(name)                      # the path


def group_sizes(path, n=50):
    return [(name, G.AtomContainer.from_indices(idx.tolist(), n).get_n_atoms(), idx) for name, idx in G.read_ndx_groups(f(path), n)]


def test_read_ndx_small_shuffled_duplicate_multiword_empty():   # ndx_io.rs:334-489
    for name in ("index_small.ndx", "index_shuffled.ndx", "index_duplicate.ndx", "index_empty_lines.ndx"):
        g = group_sizes(name)
        assert [(a, b) for a, b, _ in g] == [("System", 50), ("Protein", 50)], name
        for _, _, idx in g:
            assert set(idx.tolist()) == set(range(50))
    assert [(a, b) for a, b, _ in group_sizes("index_multiword_group.ndx")] == [("System", 50), ("Protein Named Buforin II P11L", 50)]
    assert group_sizes("index_empty.ndx") == []


@pytest.mark.parametrize("name,variant,payload", [                # ndx_io.rs:508-544
    ("nonexistent.ndx", "FileNotFound", None),
    ("index_invalid_name.ndx", "ParseGroupNameErr", "[   ] "),
    ("index_unfinished_name.ndx", "ParseLineErr", "[ Protein "),
    ("index_invalid_line.ndx", "ParseLineErr", "  16   17   18   19   20   21   -22   23   24   25   26   27   28   29   30"),
    ("index_invalid_index1.ndx", "InvalidAtomIndex", 0),
    ("index_invalid_index2.ndx", "InvalidAtomIndex", 51),
])
def test_read_ndx_fails(name, variant, payload):
    with pytest.raises(G.ParseNdxError) as e:
        G.read_ndx_groups(f(name), 50)
    assert e.value.variant == variant
    if payload is not None:
        assert e.value.detail == payload


def test_repeated_and_invalid_names_as_written():               # :546-640: what install turns into warnings
    g = group_sizes("index_duplicate_groups.ndx")
    assert [a for a, _, _ in g].count("Protein") == 2 and g[-1][1] == 32        # the later definition wins (group of 32)
    g = group_sizes("index_duplicate_groups2.ndx")
    assert [b for a, b, _ in g if a == "Protein"][-1] == 15
    names = [a for a, _, _ in group_sizes("index_invalid_names.ndx")]
    assert {"inval@id", "&also_invalid", "(parentheses are invalid)", "System", "Valid Name"} <= set(names)


def test_structure_filters():
    s = G.Structure(f("example_novelocities.gro"))
    assert s.indices_where(resname="LEU").size == sum(r == "LEU" for r in s.resname) > 0
    assert np.array_equal(s.indices_where(serial=(10, 19)), np.arange(9, 19, dtype=np.uint64))
    bb = s.indices_where(name="BB", resid=(1, 5))
    assert all(s.atomname[int(i)] == "BB" and 1 <= s.resid[int(i)] <= 5 for i in bb) and bb.size == 5
