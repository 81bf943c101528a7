"""htd_amd: MI355X-native HTD detection hot path behind the mmdet registry/config surface."""
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, ROI_EXTRACTORS, Config, ConfigDict,  # noqa: F401
                       build_detector, build_from_cfg)

__version__ = '0.1.0'
