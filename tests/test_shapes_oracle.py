"""The oracle's geometry-selection restatement (oracle/groan_oracle.c, shape.rs:110-505, group.rs:119-175) against the
reference's own known answers: the 37 single-point assertions of `mod tests_sphere / tests_rectangular / tests_cylinder /
tests_triprism` (tests/golden/shape_cases.json, extracted by tests/golden/make_shape_cases.py) and the four
`group_create_from_geometry_*` counts on example.gro (src/system/groups.rs:1578-1668)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "shape_cases.json")))


@pytest.mark.parametrize("case", CASES["points"], ids=lambda c: "%s_%s_L%d" % (c["shape"]["kind"], c["test"], c["line"]))
def test_reference_point_cases(case):
    if "inside" in case:
        assert O.shape_inside(case["shape"], case["point"], case["box"]) == case["inside"]
    if "inside_naive" in case:
        assert O.shape_inside_naive(case["shape"], case["point"]) == case["inside_naive"]


@pytest.mark.parametrize("case", CASES["groups"], ids=lambda c: "%s_L%d" % (c["source"], c["line"]))
def test_reference_group_counts(example, case):
    idx = O.container_expand(example["blocks_" + case["source"]])
    got = O.group_from_geometries(example["pos"], idx, example["box9"], case["shapes"])
    assert got.size == case["count"]
    assert np.all(np.diff(got.astype(np.int64)) > 0)            # selection order preserved


def test_prism_constructor_rejections():
    # shape.rs:948-968 (should_panic): base not in a coordinate plane / degenerate base
    with pytest.raises(O.OracleError) as e:
        O.shape({"kind": "prism", "base1": [3.0, 4.0, 2.0], "base2": [7.0, 5.0, 1.8], "base3": [4.0, 3.0, 2.0], "height": 4.3})
    assert e.value.status == 101
    with pytest.raises(O.OracleError) as e:
        O.shape({"kind": "prism", "base1": [3.0, 4.0, 2.0], "base2": [7.0, 4.0, 2.0], "base3": [4.0, 4.0, 2.0], "height": 4.3})
    assert e.value.status == 102
    s = O.shape({"kind": "prism", "base1": [5.0, 7.0, 3.0], "base2": [5.0, 0.0, 2.0], "base3": [5.0, 4.0, 5.0], "height": 4.3})
    assert (s.orientation, s.plane) == (O.DIM["x"], O.DIM["yz"])    # :937-946


def test_several_geometries_intersect(example):
    # apply_geometries (group.rs:149-175): an atom must be inside every shape
    idx = O.container_expand(example["blocks_W"])
    sphere = {"kind": "sphere", "position": [0.5, 4.5, 3.5], "radius": 4.6}
    prism = {"kind": "prism", "base1": [8.0, 8.0, 8.0], "base2": [15.0, 12.0, 8.0], "base3": [9.5, 7.3, 8.0], "height": 5.4}
    a = set(O.group_from_geometries(example["pos"], idx, example["box9"], [sphere]).tolist())
    b = set(O.group_from_geometries(example["pos"], idx, example["box9"], [prism]).tolist())
    both = O.group_from_geometries(example["pos"], idx, example["box9"], [sphere, prism])
    assert set(both.tolist()) == (a & b)
