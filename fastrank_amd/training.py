"""Training requests for the C ABI's `train_model` (JSON wire form: src/json_api.rs:13-34).

The public names and fields are those of the reference's Python package so that user code keeps
working (`TrainRequest`, `CoordinateAscentParams`, `RandomForestParams`; fastrank/training.py), but
the implementation is table-driven: every parameter class registers its serde variant name in
`_VARIANTS`, and `TrainRequest` (de)serialises through that registry.
"""
import dataclasses
import random
from typing import Any, ClassVar, Dict, Optional, Type

from .clib import CQRel, query_json

_VARIANTS: Dict[str, Type["_LearnerParams"]] = {}
# one seed per process, like the reference module (its dataclass default is evaluated at import)
_PROCESS_SEED = random.getrandbits(64)


class _LearnerParams:
    """Behaviour shared by the learner parameter dataclasses."""

    VARIANT: ClassVar[str] = ""

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if cls.VARIANT:
            _VARIANTS[cls.VARIANT] = cls

    def name(self) -> str:
        return self.VARIANT

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, params: Dict[str, Any]):
        return cls(**params)


@dataclasses.dataclass
class CoordinateAscentParams(_LearnerParams):
    """src/coordinate_ascent.rs:11-41.  The wire form requires all ten keys."""

    VARIANT: ClassVar[str] = "CoordinateAscent"

    num_restarts: int = 5
    num_max_iterations: int = 25
    step_base: float = 0.05
    step_scale: float = 2.0
    tolerance: float = 0.001
    normalize: bool = True
    init_random: bool = True
    output_ensemble: bool = False
    seed: int = _PROCESS_SEED
    quiet: bool = False


@dataclasses.dataclass
class RandomForestParams(_LearnerParams):
    """src/random_forest.rs:127-157; trained on the device (csrc/rf_train.hpp, kernels_rf.inc).  As in the reference
    (fastrank/training.py:36-60) the dataclass default of `split_method` is a bare string, which the request parser --
    serde there, its restatement here -- rejects: requests made by `TrainRequest.random_forest()` carry the wire
    form {"SquaredError": []} (one of SquaredError, BinaryGiniImpurity, InformationGain, TrueVarianceReduction)."""

    VARIANT: ClassVar[str] = "RandomForest"

    num_trees: int = 100
    weight_trees: bool = True
    split_method: Any = "SquaredError"
    instance_sampling_rate: float = 0.5
    feature_sampling_rate: float = 0.25
    min_leaf_support: int = 10
    split_candidates: int = 3
    max_depth: int = 8
    seed: int = _PROCESS_SEED
    quiet: bool = False


@dataclasses.dataclass
class TrainRequest:
    """What to optimise (`measure`: "ndcg", "ndcg@10", "map", "mrr", ...), how (`params`) and,
    optionally, the judgments that define ideal gains / relevant counts."""

    measure: str = "ndcg"
    params: _LearnerParams = dataclasses.field(default_factory=CoordinateAscentParams)
    judgments: Optional[CQRel] = None

    def to_dict(self) -> Dict[str, Any]:
        wire: Dict[str, Any] = {"measure": self.measure, "params": {self.params.name(): self.params.to_dict()}}
        wire["judgments"] = self.judgments.to_dict() if self.judgments is not None else None
        return wire

    @staticmethod
    def from_dict(params: Dict[str, Any]) -> "TrainRequest":
        variants = params["params"]
        if len(variants) != 1:
            raise ValueError("What do I do with this?: {}".format(variants))
        (variant, fields), = variants.items()
        if variant not in _VARIANTS:
            raise ValueError("Python doesn't know about model-params: {}".format(variants))
        qrel = params.get("judgments")
        return TrainRequest(
            measure=params["measure"],
            params=_VARIANTS[variant].from_dict(fields),
            judgments=CQRel.from_dict(qrel) if qrel is not None else None,
        )

    def clone(self) -> "TrainRequest":
        """An independent copy (through the wire form)."""
        return TrainRequest.from_dict(self.to_dict())

    @staticmethod
    def _defaults(which: str) -> "TrainRequest":
        # defaults come from the native side, like the reference (src/ffi.rs:215-236)
        return TrainRequest.from_dict(query_json(which))

    @staticmethod
    def coordinate_ascent() -> "TrainRequest":
        return TrainRequest._defaults("coordinate_ascent_defaults")

    @staticmethod
    def random_forest() -> "TrainRequest":
        return TrainRequest._defaults("random_forest_defaults")
