"""MI355X-native EKF-SLAM core: host-side Python plumbing around libekfslam_hip.so.

The directory name is not a Python identifier; load it with `__graft_entry__.load_package()`
(importlib, registered as module `ekfslam_amd`).
"""
from . import ekfslam, features, montecarlo, scenarios  # noqa: F401
from .ekfslam import FilterBatch, KalmanFilter, EkfError, load  # noqa: F401
from .features import FeatureExtractor  # noqa: F401
