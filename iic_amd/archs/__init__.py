"""Architecture registry mirroring /root/reference/code/archs/__init__.py: the reference
builds a net with ``archs.__dict__[config.arch](config)`` (cluster_sobel.py:140)."""
from .cluster import ClusterNet5g, ClusterNet5gTwoHead  # noqa: F401
from .seg import SegmentationNet10a, SegmentationNet10aTwoHead  # noqa: F401
from .vgg import ClusterNet6c, ClusterNet6cTwoHead  # noqa: F401

__all__ = ["ClusterNet5g", "ClusterNet5gTwoHead", "ClusterNet6c", "ClusterNet6cTwoHead",
           "SegmentationNet10a", "SegmentationNet10aTwoHead"]
