"""Geometry-selection shapes: the mirror of src/structures/shape.rs (Sphere, Rectangular, Cylinder, TriangularPrism with
`inside` / `inside_naive`) over the C ABI's `gr_shape` (include/groan_hip.h).  Constructors raise ValueError where the
reference panics (unsupported cylinder orientation, a prism base that is not in a coordinate plane or is degenerate)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OK


class GrShape(C.Structure):
    _fields_ = [("kind", C.c_int), ("position", C.c_float * 3), ("size", C.c_float * 3), ("base2", C.c_float * 3),
                ("base3", C.c_float * 3), ("orientation", C.c_int), ("plane", C.c_int)]


def _v3(v):
    a = np.ascontiguousarray(v, dtype=np.float32).ravel()
    if a.size != 3:
        raise ValueError("expected three coordinates")
    return a


class Shape:
    """Shape::inside (shape.rs:70-74) / NaiveShape::inside_naive (:464-467)"""
    has_naive = True

    def __init__(self):
        self._c = GrShape()

    def _check(self, st, what):
        if st != OK:
            raise ValueError("FATAL GROAN ERROR | %s" % what)

    def inside(self, point, simbox):
        """`simbox`: 3 or 9 floats in gro order (orthogonal, like the reference)"""
        p = _v3(point)
        b = np.zeros(9, np.float32); sb = np.asarray(simbox, np.float32).ravel(); b[: sb.size] = sb
        out = C.c_int(0)
        st = _lib.load().gr_shape_inside(C.byref(self._c), p.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 0, C.byref(out))
        if st != OK:
            raise ValueError("gr_shape_inside: status %d" % st)
        return bool(out.value)

    def inside_naive(self, point):
        if not self.has_naive:
            raise TypeError("%s does not implement NaiveShape" % type(self).__name__)
        p = _v3(point)
        out = C.c_int(0)
        st = _lib.load().gr_shape_inside(C.byref(self._c), p.ctypes.data_as(C.c_void_p), None, 1, C.byref(out))
        if st != OK:
            raise ValueError("gr_shape_inside: status %d" % st)
        return bool(out.value)


class Sphere(Shape):
    def __init__(self, position, radius):
        super().__init__()
        self.position, self.radius = _v3(position), float(radius)
        self._check(_lib.load().gr_shape_sphere(C.byref(self._c), self.position.ctypes.data_as(C.c_void_p), C.c_float(radius)), "Sphere::new")

    def get_position(self): return self.position
    def get_radius(self): return self.radius


class Rectangular(Shape):
    def __init__(self, position, x, y, z):
        super().__init__()
        self.position, self.x, self.y, self.z = _v3(position), float(x), float(y), float(z)
        self._check(_lib.load().gr_shape_rectangular(C.byref(self._c), self.position.ctypes.data_as(C.c_void_p), C.c_float(x), C.c_float(y), C.c_float(z)),
                    "Rectangular::new")

    def get_position(self): return self.position
    def get_x(self): return self.x
    def get_y(self): return self.y
    def get_z(self): return self.z


class Cylinder(Shape):
    def __init__(self, position, radius, height, orientation):
        super().__init__()
        self.position, self.radius, self.height, self.orientation = _v3(position), float(radius), float(height), orientation
        self._check(_lib.load().gr_shape_cylinder(C.byref(self._c), self.position.ctypes.data_as(C.c_void_p), C.c_float(radius), C.c_float(height), int(orientation)),
                    "Cylinder::new | Unsupported orientation dimension '%s'." % (getattr(orientation, "name", orientation),))

    def get_position(self): return self.position
    def get_radius(self): return self.radius
    def get_height(self): return self.height
    def get_orientation(self): return self.orientation


class TriangularPrism(Shape):
    has_naive = False

    def __init__(self, base1, base2, base3, height):
        super().__init__()
        self.base1, self.base2, self.base3, self.height = _v3(base1), _v3(base2), _v3(base3), float(height)
        self._check(_lib.load().gr_shape_triangular_prism(C.byref(self._c), self.base1.ctypes.data_as(C.c_void_p), self.base2.ctypes.data_as(C.c_void_p),
                                                           self.base3.ctypes.data_as(C.c_void_p), C.c_float(height)),
                    "TriangularPrism::new | Base of the requested TriangularPrism does not lie in xy, xz, nor yz plane / can not be constructed.")

    def get_base1(self): return self.base1
    def get_base2(self): return self.base2
    def get_base3(self): return self.base3
    def get_height(self): return self.height
    def get_orientation(self): return int(self._c.orientation)
    def get_plane(self): return int(self._c.plane)


def pack(shapes):
    """-> ctypes array of gr_shape for gr_group_create_from_geometries"""
    shapes = list(shapes)
    return (GrShape * max(len(shapes), 1))(*[s._c for s in shapes]), len(shapes)
