"""groan_rs_amd -- MI355X-native per-frame geometry engine behind the groan_rs System API.

The product is groan_rs_amd/libgroan_hip.so (hand-written HIP kernels for gfx950 + the C ABI of
include/groan_hip.h).  This package is the thin host-side mirror of the reference interface used by
the tests and the benchmark; importing it does not load the library (so CPU-only tooling can import
it), but every operation does and fails loudly when the library is missing.
"""
from . import _lib
from .system import (AtomContainer, AtomError, AtomIterator, DeviceError, Dimension, GroanError, GroupError, RMSDError,
                     RMSDPlan, SimBoxError, System, pinned_array, pinned_free)
from .traj import (FrameAnalyze, FrameConvert, FrameConvertAnalyze, RMSDConverterAnalyzer, TrajAnalyzer,
                   TrajAnalysisError, TrajConverter, TrajConverterAnalyzer, TrajReader)
from .xtc import XtcError, XtcFile, XtcWriter
from .trr import TrrFile, TrrWriter
from .textio import ParseGroError, ParseNdxError, Structure, read_ndx_groups, system_from_gro, system_read_ndx
from .select import SelectError, parse_query, select
from .shapes import Cylinder, Rectangular, Shape, Sphere, TriangularPrism
from .parallel import AbortedByOtherRank, Comm, ParallelTrajData, Pool, gather_per_frame, interleave, shard_frames, traj_iter_map_reduce

__all__ = [
    "AtomContainer", "AtomError", "AtomIterator", "DeviceError", "Dimension", "GroanError", "GroupError", "RMSDError", "RMSDPlan",
    "SimBoxError", "System", "pinned_array", "pinned_free", "FrameAnalyze", "FrameConvert", "FrameConvertAnalyze", "RMSDConverterAnalyzer",
    "TrajAnalyzer", "TrajAnalysisError", "TrajConverter", "TrajConverterAnalyzer", "TrajReader",
    "XtcError", "XtcFile", "XtcWriter", "ParseGroError", "ParseNdxError", "Structure", "read_ndx_groups", "system_from_gro", "system_read_ndx", "Cylinder", "Rectangular", "Shape", "Sphere", "TriangularPrism", "AbortedByOtherRank", "Comm", "Pool", "ParallelTrajData", "gather_per_frame", "interleave", "shard_frames", "traj_iter_map_reduce",
]
