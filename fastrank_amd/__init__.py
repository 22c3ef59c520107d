"""fastrank_amd: MI355X-native drop-in for the fastrank learning-to-rank hot path.

Same names as the reference package (fastrank/__init__.py:2-7): CQRel, CDataset, CModel,
query_json, TrainRequest.  All computation happens in libfastrank_amd.so (HIP, gfx950).
"""
from .clib import CDataset, CModel, CQRel, query_json
from .training import CoordinateAscentParams, RandomForestParams, TrainRequest

VERSION_TUPLE = (0, 7, 0)
__version__ = "{}.{}.{}".format(*VERSION_TUPLE)

__all__ = ["clib", "training", "native", "CQRel", "CDataset", "CModel", "query_json", "TrainRequest",
           "CoordinateAscentParams", "RandomForestParams"]
