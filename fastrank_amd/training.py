"""Training-request dataclasses, JSON-compatible with the reference's TrainRequest wire form
(src/json_api.rs:13-34; Python mirror fastrank/training.py:7-134)."""
import random
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, Optional, Union

from .clib import CQRel, query_json


@dataclass
class CoordinateAscentParams:
    """src/coordinate_ascent.rs:11-23 (all ten keys are required on the wire)."""

    num_restarts: int = 5
    num_max_iterations: int = 25
    step_base: float = 0.05
    step_scale: float = 2.0
    tolerance: float = 0.001
    normalize: bool = True
    init_random: bool = True
    output_ensemble: bool = False
    seed: int = random.randint(0, (1 << 64) - 1)
    quiet: bool = False

    def name(self):
        return "CoordinateAscent"

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_dict(params) -> "CoordinateAscentParams":
        return CoordinateAscentParams(**params)


@dataclass
class RandomForestParams:
    """src/random_forest.rs:127-139.  Forest *training* is not part of the MI355X hot path; the
    dataclass exists so requests round-trip and forests can be scored."""

    num_trees: int = 100
    weight_trees: bool = True
    split_method: Any = "SquaredError"
    instance_sampling_rate: float = 0.5
    feature_sampling_rate: float = 0.25
    min_leaf_support: int = 10
    split_candidates: int = 3
    max_depth: int = 8
    seed: int = random.randint(0, (1 << 64) - 1)
    quiet: bool = False

    def name(self):
        return "RandomForest"

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_dict(params) -> "RandomForestParams":
        return RandomForestParams(**params)


@dataclass
class TrainRequest:
    measure: str = "ndcg"
    params: Union[CoordinateAscentParams, RandomForestParams] = field(default_factory=CoordinateAscentParams)
    judgments: Optional[CQRel] = None

    def to_dict(self) -> Dict[str, Any]:
        judgments = None if self.judgments is None else self.judgments.to_dict()
        return {"measure": self.measure, "params": {self.params.name(): self.params.to_dict()}, "judgments": judgments}

    def clone(self) -> "TrainRequest":
        return TrainRequest.from_dict(self.to_dict())

    @staticmethod
    def coordinate_ascent() -> "TrainRequest":
        return TrainRequest.from_dict(query_json("coordinate_ascent_defaults"))

    @staticmethod
    def random_forest() -> "TrainRequest":
        return TrainRequest.from_dict(query_json("random_forest_defaults"))

    @staticmethod
    def from_dict(params) -> "TrainRequest":
        measure = params["measure"]
        judgments = None
        if params["judgments"] is not None:
            judgments = CQRel.from_dict(params["judgments"])
        params_dict = params["params"]
        if len(params_dict) != 1:
            raise ValueError("What do I do with this?: {}".format(params_dict))
        if "RandomForest" in params_dict:
            parsed = RandomForestParams.from_dict(params_dict["RandomForest"])
        elif "CoordinateAscent" in params_dict:
            parsed = CoordinateAscentParams.from_dict(params_dict["CoordinateAscent"])
        else:
            raise ValueError("Python doesn't know about model-params: {}".format(params_dict))
        return TrainRequest(measure, parsed, judgments)
