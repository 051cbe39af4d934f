"""Trajectory adapters: the reference's plug-in boundary for per-frame work.

Mirrors src/structures/traj_convert.rs: FrameConvert / FrameAnalyze / FrameConvertAnalyze (one method
each) and the iterator wrappers TrajConverter / TrajAnalyzer / TrajConverterAnalyzer that call them once
per frame, plus RMSDConverterAnalyzer and the RMSDTrajRead helpers (src/system/rmsd.rs:170-398).

A "reader" here is any iterable yielding frames as (positions[n,3] float32, box9 or None[, step, time])
-- what an xtc decoder hands to TrajRead::update_system.  `TrajReader` uploads each frame into slot 0 of
the System (update_system, src/io/traj_read.rs:160-186) and yields the System, like the reference's
`xtc_iter` yields `&mut System`.
"""
from .system import RMSDPlan


class TrajReadError(Exception):
    pass


class TrajAnalysisError(Exception):      # errors.rs:691-729
    def __init__(self, variant, inner):
        super().__init__("%s(%r)" % (variant, inner))
        self.variant, self.inner = variant, inner


class FrameConvert:                      # traj_convert.rs:30-36
    def convert(self, system):
        raise NotImplementedError


class FrameAnalyze:                      # traj_convert.rs:76-83
    def analyze(self, system):
        raise NotImplementedError


class FrameConvertAnalyze:               # traj_convert.rs:125-132
    def convert_analyze(self, system):
        raise NotImplementedError


class TrajReader:
    """TrajReader::next (traj_read.rs:160-186): update the System with each frame and yield it."""

    def __init__(self, system, frames, slot=0):
        self.system, self._it, self.slot = system, iter(frames), slot

    def __iter__(self):
        return self

    def __next__(self):
        fr = next(self._it)
        pos, box = fr[0], fr[1]
        step = fr[2] if len(fr) > 2 else None
        time = fr[3] if len(fr) > 3 else None
        try:
            self.system.set_frame(pos, box, slot=self.slot, step=step, time=time)
        except ValueError as e:
            raise TrajReadError(str(e))
        return self.system

    # ConvertableTrajRead (traj_convert.rs:161-203)
    def convert(self, converter): return TrajConverter(self, converter)
    def analyze(self, analyzer): return TrajAnalyzer(self, analyzer)
    def convert_and_analyze(self, ca): return TrajConverterAnalyzer(self, ca)

    # RMSDTrajRead (rmsd.rs:258-398)
    def calc_rmsd(self, reference, group):
        return TrajAnalyzer(self, RMSDConverterAnalyzer(reference, self.system, group))

    def calc_rmsd_and_fit(self, reference, group):
        return TrajConverterAnalyzer(self, RMSDConverterAnalyzer(reference, self.system, group))


class TrajConverter:                     # traj_convert.rs:14-57
    def __init__(self, reader, converter):
        self.reader, self.converter = reader, converter

    def __iter__(self):
        return self

    def __next__(self):
        frame = next(self.reader)
        try:
            self.converter.convert(frame)
        except Exception as e:
            raise TrajAnalysisError("ConversionError", e)
        return frame


class TrajAnalyzer:                      # traj_convert.rs:59-105
    def __init__(self, reader, analyzer):
        self.reader, self.analyzer = reader, analyzer

    def __iter__(self):
        return self

    def __next__(self):
        frame = next(self.reader)
        try:
            return frame, self.analyzer.analyze(frame)
        except StopIteration:
            raise
        except Exception as e:
            raise TrajAnalysisError("AnalysisError", e)


class TrajConverterAnalyzer:             # traj_convert.rs:107-157
    def __init__(self, reader, ca):
        self.reader, self.ca = reader, ca

    def __iter__(self):
        return self

    def __next__(self):
        frame = next(self.reader)
        try:
            return frame, self.ca.convert_analyze(frame)
        except Exception as e:
            raise TrajAnalysisError("ConversionAnalysisError", e)


class RMSDConverterAnalyzer(FrameAnalyze, FrameConvertAnalyze):
    """src/system/rmsd.rs:170-251: reference side cached once, per-frame analyze / convert_analyze."""

    def __init__(self, reference, target, group, ref_slot=0, slot=0):
        self.plan = RMSDPlan(reference, target, group, ref_slot)
        self.slot = slot

    def analyze(self, system):
        r, _ = self.plan.rmsd(self.slot, 1)
        return float(r[0])

    def convert_analyze(self, system):
        r, _ = self.plan.rmsd_fit(self.slot, 1)
        return float(r[0])
