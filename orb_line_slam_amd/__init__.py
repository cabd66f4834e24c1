"""orb_line_slam_amd -- MI355X (gfx950) implementation of the per-frame feature path of
ORB_Line_SLAM: ORBextractor, Lineextractor, ORBmatcher / LineMatcher Hamming searches and the
Frame stereo matchers, as hand-written HIP kernels behind a C ABI (include/orbline.h).

The Python classes here mirror the reference's C++ class surfaces (same names, argument meaning and
error behaviour) on top of that C ABI via ctypes.  There is no CPU fallback: constructing any
compute object without the built HIP library or without a GPU raises.
"""
from ._lib import (KEYPOINT_DTYPE, KEYLINE_DTYPE, OlfError, OlfParams, default_params, device_count,
                   lib, last_error)
from .extractor import ORBextractor, Lineextractor
from .matcher import ORBmatcher, FrameView, KeyFrameView, MapPointView, MapPointGeom
from .vocabulary import ORBVocabulary, LineVocabulary
from . import matcher as LineMatcher
from .frame import StereoFrontEnd, StereoFrames
from . import synth

__all__ = ["ORBextractor", "Lineextractor", "ORBmatcher", "FrameView", "KeyFrameView", "MapPointView", "MapPointGeom", "ORBVocabulary", "LineVocabulary", "LineMatcher", "StereoFrontEnd", "StereoFrames", "KEYPOINT_DTYPE", "KEYLINE_DTYPE", "OlfError", "OlfParams", "default_params",
           "device_count", "lib", "last_error", "synth"]
