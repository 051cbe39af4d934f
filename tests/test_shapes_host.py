"""The product's host-side shape predicates (gr_shape_* of include/groan_hip.h through groan_rs_amd.shapes) against the
reference's own known answers (src/structures/shape.rs `mod tests_*`, tests/golden/shape_cases.json) and constructor
rules.  Host functions only: runs without a GPU."""
import json
import os

import pytest

import groan_rs_amd as G

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "shape_cases.json")))


def build(spec):
    k = spec["kind"]
    if k == "sphere": return G.Sphere(spec["position"], spec["radius"])
    if k == "rectangular": return G.Rectangular(spec["position"], *spec["size"])
    if k == "cylinder": return G.Cylinder(spec["position"], spec["radius"], spec["height"], G.Dimension[spec["orientation"]])
    return G.TriangularPrism(spec["base1"], spec["base2"], spec["base3"], spec["height"])


@pytest.mark.parametrize("case", CASES["points"], ids=lambda c: "%s_%s_L%d" % (c["shape"]["kind"], c["test"], c["line"]))
def test_reference_point_cases(case):
    s = build(case["shape"])
    if "inside" in case:
        assert s.inside(case["point"], case["box"]) == case["inside"]
    if "inside_naive" in case:
        assert s.inside_naive(case["point"]) == case["inside_naive"]


def test_constructors():
    c = G.Cylinder([1.0, 2.0, 3.0], 0.8, 4.3, G.Dimension.Y)                       # shape.rs:729-738
    assert (c.get_radius(), c.get_height(), c.get_orientation()) == (pytest.approx(0.8), pytest.approx(4.3), G.Dimension.Y)
    with pytest.raises(ValueError):                                                  # Cylinder::new panics for XY (:215-218)
        G.Cylinder([1.0, 2.0, 3.0], 0.8, 4.3, G.Dimension.XY)
    for (b1, b2, b3), orient, plane in ((([3, 4, 2], [7, 5, 2], [4, 3, 2]), G.Dimension.Z, G.Dimension.XY),       # :908-946
                                        (([3, 3, 3], [7, 3, 2], [4, 3, 5]), G.Dimension.Y, G.Dimension.XZ),
                                        (([5, 7, 3], [5, 0, 2], [5, 4, 5]), G.Dimension.X, G.Dimension.YZ)):
        p = G.TriangularPrism(b1, b2, b3, 4.3)
        assert (p.get_orientation(), p.get_plane()) == (orient, plane)
    with pytest.raises(ValueError):                                                  # :948-957
        G.TriangularPrism([3.0, 4.0, 2.0], [7.0, 5.0, 1.8], [4.0, 3.0, 2.0], 4.3)
    with pytest.raises(ValueError):                                                  # :959-968
        G.TriangularPrism([3.0, 4.0, 2.0], [7.0, 4.0, 2.0], [4.0, 4.0, 2.0], 4.3)
    with pytest.raises(TypeError):                                                   # no NaiveShape for the prism (:466-505)
        G.TriangularPrism([3, 4, 2], [7, 5, 2], [4, 3, 2], 4.3).inside_naive([1, 1, 1])
