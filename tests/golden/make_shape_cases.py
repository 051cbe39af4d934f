#!/usr/bin/env python3
"""Extract the known answers of the reference's shape tests (src/structures/shape.rs, `mod tests_*`) into
tests/golden/shape_cases.json: for every test that builds one shape, one point and one box and asserts
`inside` / `inside_naive`, the numbers and the expected booleans (+ the reference line of the test).
Data only: shape parameters, the point, the box and the asserted truth values.  Build container only."""
import json
import os
import re

REF = os.environ.get("GROAN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"(-?\d+(?:\.\d+)?)"
V3 = r"\[?\(?\s*" + NUM + r",\s*" + NUM + r",\s*" + NUM + r"\s*\)?\]?"


def main():
    lines = open(os.path.join(REF, "src", "structures", "shape.rs")).read().split("\n")
    cases, i = [], 0
    while i < len(lines):
        m = re.match(r"\s*fn (\w+)\(\) \{", lines[i])
        if not m:
            i += 1; continue
        name, start = m.group(1), i + 1
        body = []
        i += 1
        while i < len(lines) and not re.match(r"    \}", lines[i]):
            body.append(lines[i]); i += 1
        text = "\n".join(body)
        shape = None
        m = re.search(r"Sphere::new\(" + V3 + r"\.into\(\),\s*" + NUM + r"\)", text)
        if m: shape = {"kind": "sphere", "position": [float(x) for x in m.groups()[:3]], "radius": float(m.group(4))}
        m = re.search(r"Rectangular::new\(" + V3 + r"\.into\(\),\s*" + NUM + r",\s*" + NUM + r",\s*" + NUM + r"\)", text)
        if m: shape = {"kind": "rectangular", "position": [float(x) for x in m.groups()[:3]], "size": [float(x) for x in m.groups()[3:6]]}
        m = re.search(r"Cylinder::new\(" + V3 + r"\.into\(\),\s*" + NUM + r",\s*" + NUM + r",\s*Dimension::(\w)\)", text)
        if m: shape = {"kind": "cylinder", "position": [float(x) for x in m.groups()[:3]], "radius": float(m.group(4)), "height": float(m.group(5)), "orientation": m.group(6)}
        bases = re.findall(r"let base(\d) = Vector3D::new\(" + NUM + r",\s*" + NUM + r",\s*" + NUM + r"\);", text)
        mh = re.search(r"TriangularPrism::new\(base1, base2, base3,\s*" + NUM + r"\)", text)
        if len(bases) == 3 and mh:
            shape = {"kind": "prism", "height": float(mh.group(1))}
            for k, x, y, z in bases: shape["base" + k] = [float(x), float(y), float(z)]
        mp = re.search(r"let point = Vector3D::new\(" + NUM + r",\s*" + NUM + r",\s*" + NUM + r"\);", text)
        mb = re.search(r"SimBox::from\(\[" + NUM + r",\s*" + NUM + r",\s*" + NUM + r"\]\)", text)
        if shape is None or mp is None:
            continue
        case = {"test": name, "line": start, "shape": shape, "point": [float(x) for x in mp.groups()]}
        if mb: case["box"] = [float(x) for x in mb.groups()]
        for neg, var, fn in re.findall(r"assert!\((!?)(\w+)\.(inside|inside_naive)\(&point", text):
            case[fn] = (neg == "")
        if "inside" in case or "inside_naive" in case:
            cases.append(case)
    # group-level known answers on example.gro (src/system/groups.rs:1578-1668)
    groups = [
        {"line": 1578, "source": "Membrane", "count": 206, "shapes": [{"kind": "cylinder", "position": [5.0, 8.0, 3.0], "radius": 2.0, "height": 6.0, "orientation": "Y"}]},
        {"line": 1598, "source": "W", "count": 1881, "shapes": [{"kind": "sphere", "position": [0.5, 4.5, 3.5], "radius": 4.6}]},
        {"line": 1618, "source": "Protein", "count": 25, "shapes": [{"kind": "rectangular", "position": [5.0, 0.0, 2.0], "size": [5.0, 4.0, 4.3]}]},
        {"line": 1645, "source": "W", "count": 213, "shapes": [{"kind": "prism", "base1": [8.0, 8.0, 8.0], "base2": [15.0, 12.0, 8.0], "base3": [9.5, 7.3, 8.0], "height": 5.4}]},
    ]
    json.dump({"points": cases, "groups": groups}, open(os.path.join(HERE, "shape_cases.json"), "w"), indent=1)
    print(len(cases), "point cases;", len(groups), "group cases")


if __name__ == "__main__":
    main()
