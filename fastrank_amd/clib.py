"""ctypes binding of libfastrank_amd.so that mirrors the reference's Python surface.

Same class / method names, argument meaning and error behaviour as the reference's
fastrank/clib.py (CQRel :62-121, CModel :124-209, CDataset :212-488, query_json :491-503), so a
user script only changes its import.  The reference binds a Rust cdylib through cffi; cffi is
not part of this image, and the C ABI is identical (include/fastrank.h), so this module uses
ctypes.  Nothing here computes: every call crosses the C ABI into the HIP library.
"""
import ctypes as C
import json
import os
from typing import Dict, List, Optional, Set

from . import _build

# model.rs:10-16 ModelEnum variants (clib.py:6 keeps the same list)
_MODEL_TYPES = ["SingleFeature", "Linear", "DecisionTree", "Ensemble"]


# int (*fr_allreduce_sum_fn)(void *ctx, double *values, size_t n)  (include/fastrank.h)
ALLREDUCE_SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


class _CResult(C.Structure):
    _fields_ = [("error_message", C.c_void_p), ("success", C.c_void_p)]


_lib = None


def _load():
    """dlopen the in-tree library (building it first if hipcc is around and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if _build.needs_build():
        try:
            _build.build()
        except Exception as exc:  # keep a stale-but-present library usable on boxes without hipcc
            if not os.path.exists(path):
                raise ImportError(
                    "libfastrank_amd.so is not built and could not be built: {}".format(exc)
                ) from exc
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    res = C.POINTER(_CResult)
    sigs = {
        "free_str": (None, [vp]),
        "free_c_result": (None, [res]),
        "free_dataset": (None, [vp]),
        "free_model": (None, [vp]),
        "free_cqrel": (None, [vp]),
        "load_cqrel": (res, [C.c_char_p]),
        "cqrel_from_json": (res, [C.c_char_p]),
        "cqrel_query_json": (vp, [vp, C.c_char_p]),
        "load_ranksvm_format": (res, [C.c_char_p, C.c_char_p]),
        "dataset_query_sampling": (res, [vp, C.c_char_p]),
        "dataset_feature_sampling": (res, [vp, C.c_char_p]),
        "dataset_query_json": (vp, [vp, C.c_char_p]),
        "query_json": (vp, [C.c_char_p]),
        "make_dense_dataset_f32_f64_i64": (res, [sz, sz, vp, vp, vp]),
        "train_model": (res, [C.c_char_p, vp]),
        "model_from_json": (res, [C.c_char_p]),
        "model_query_json": (vp, [vp, C.c_char_p]),
        "evaluate_by_query": (vp, [vp, vp, vp, C.c_char_p]),
        "predict_scores": (vp, [vp, vp]),
        "predict_to_trecrun": (vp, [vp, vp, C.c_char_p, C.c_char_p, sz]),
        # extensions
        "fr_device_count": (C.c_int, []),
        "fr_set_device": (C.c_int, [C.c_int]),
        "fr_version": (C.c_char_p, []),
        "fr_train_model_shard": (vp, [C.c_char_p, vp, C.c_uint32, C.c_uint32]),
        "fr_select_model": (res, [C.c_char_p, C.c_int]),
        "fr_ca_begin": (vp, [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "fr_ca_begin_query_shard": (vp, [C.c_char_p, vp, C.c_uint64, ALLREDUCE_SUM_FN, vp, C.POINTER(vp)]),
        "fr_ca_step": (vp, [vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
        "fr_ca_state": (vp, [vp]),
        "fr_ca_free": (None, [vp]),
        "fr_last_train_stats": (vp, []),
        "fr_predict_scores_dense": (vp, [vp, vp, vp, sz]),
        "fr_evaluate_dense": (vp, [vp, vp, vp, C.c_char_p, vp, sz, C.POINTER(vp)]),
        "fr_rank_order": (vp, [vp, vp, vp, sz, vp, sz]),
        "fr_dataset_num_queries": (sz, [vp]),
        "fr_dataset_num_instances": (sz, [vp]),
        "fr_dataset_device_info": (vp, [vp]),
        "fr_evaluate_candidates": (vp, [vp, vp, C.c_char_p, sz, vp, vp, vp, vp, vp, vp]),
        "fr_profile_enable": (None, [C.c_int]),
        "fr_profile_reset": (None, []),
        "fr_profile_json": (vp, []),
        "fr_debug_resident_bound": (C.c_double, [C.c_int, C.c_double, C.c_double, C.c_double]),
        "fr_synchronize": (C.c_int, []),
        "fr_debug_device_plan": (vp, [C.c_char_p, C.c_int, C.c_uint32, C.c_int]),
        "fr_debug_fullrank_class": (C.c_uint32, [C.c_uint32]),
        "fr_debug_restart_queue": (vp, [C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32]),
        "fr_dataset_release_replicas": (sz, [vp]),
        "fr_debug_peer_copy": (vp, [C.c_int, C.c_int, sz]),
        "fr_pack_restart_records": (vp, [C.c_char_p, sz, sz, vp]),
        "fr_unpack_restart_records": (vp, [vp, sz, sz]),
        "fr_rccl_allgather": (vp, [vp, sz, vp, sz, vp]),
        "fr_debug_rccl_selftest": (vp, [C.c_int]),
        "fr_debug_walk_tiles": (vp, [vp, vp, vp, sz, vp, vp, sz, sz]),
    }
    for name, (restype, argtypes) in sigs.items():
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


EXPORTED_SYMBOLS = None  # filled lazily by exported_symbols()


def exported_symbols() -> List[str]:
    """Names this binding expects the shared object to export (used by the ABI test)."""
    _load()
    return [
        "free_str", "free_c_result", "free_dataset", "free_model", "free_cqrel", "load_cqrel",
        "cqrel_from_json", "cqrel_query_json", "load_ranksvm_format", "dataset_query_sampling",
        "dataset_feature_sampling", "dataset_query_json", "query_json",
        "make_dense_dataset_f32_f64_i64", "train_model", "model_from_json", "model_query_json",
        "evaluate_by_query", "predict_scores", "predict_to_trecrun",
    ]


def _take_str(ptr) -> Optional[str]:
    """Copy a library-allocated C string into Python and release it (free_str)."""
    if not ptr:
        return None
    try:
        return C.cast(ptr, C.c_char_p).value.decode("utf-8")
    finally:
        _load().free_str(ptr)


def _raise_if_error_str(text: Optional[str]):
    if text is None:
        return
    if "{" in text:
        payload = json.loads(text)
        if "error" in payload and "context" in payload:
            raise Exception("{0}: {1}".format(payload["error"], payload["context"]))
    else:
        raise Exception(text)


def _raise_if_error_json(payload):
    if isinstance(payload, dict) and "error" in payload and "context" in payload:
        raise Exception("{0}: {1}".format(payload["error"], payload["context"]))


def _unwrap(result_ptr):
    """CResult{error_message, success} -> success pointer, or raise (src/ffi.rs:57-74)."""
    if not result_ptr:
        raise ValueError("CResult should not be NULL")
    res = result_ptr.contents
    err_ptr, ok_ptr = res.error_message, res.success
    _load().free_c_result(result_ptr)
    _raise_if_error_str(_take_str(err_ptr))
    return ok_ptr


def _status(ptr):
    """Extension calls return NULL on success or an error-envelope string."""
    _raise_if_error_str(_take_str(ptr))


def _json_reply(ptr):
    payload = json.loads(_take_str(ptr))
    _raise_if_error_json(payload)
    return payload


class CQRel:
    """A loaded set of TREC relevance judgments (needed for MAP's relevant counts and NDCG's
    ideal gains when a dataset holds only part of the judged pool)."""

    def __init__(self, pointer=None):
        self.pointer = pointer
        self._queries = None

    def __del__(self):
        if getattr(self, "pointer", None) is not None and _lib is not None:
            _lib.free_cqrel(self.pointer)
            self.pointer = None

    @staticmethod
    def load_file(path: str) -> "CQRel":
        return CQRel(_unwrap(_load().load_cqrel(path.encode("utf-8"))))

    @staticmethod
    def from_dict(dictionaries: Dict[str, Dict[str, float]]) -> "CQRel":
        return CQRel(_unwrap(_load().cqrel_from_json(json.dumps(dictionaries).encode("utf-8"))))

    def _require_init(self):
        if self.pointer is None:
            raise ValueError("CQRel is null!")

    def _query_json(self, message="queries"):
        self._require_init()
        return _json_reply(_load().cqrel_query_json(self.pointer, message.encode("utf-8")))

    def to_dict(self) -> Dict[str, Dict[str, float]]:
        return self._query_json("to_json")

    def queries(self) -> Set[str]:
        if self._queries is None:
            self._queries = set(self._query_json("queries"))
        return self._queries

    def query_judgments(self, qid: str) -> Dict[str, float]:
        if qid in self.queries():
            return self._query_json(qid)
        raise ValueError("No qid={0} in cqrel: {1}".format(qid, self.queries()))


class CModel:
    """A trained or deserialised ranking model living behind the C ABI."""

    def __init__(self, pointer, params=None):
        self.pointer = pointer
        self.params = params

    @staticmethod
    def _check_model_json(model_json: Dict):
        [single_key] = list(model_json.keys())
        assert single_key in _MODEL_TYPES

    @staticmethod
    def from_dict(model_json: Dict) -> "CModel":
        CModel._check_model_json(model_json)
        return CModel(_unwrap(_load().model_from_json(json.dumps(model_json).encode("utf-8"))))

    def predict_dense_scores(self, dataset: "CDataset", missing: float = float("nan")) -> List[float]:
        """Scores as a list indexed by instance id, 0 .. the largest id the dataset view holds; ids it does not hold
        (a sampled view) read `missing` (same result as fastrank/clib.py:159-168)."""
        by_id = self.predict_scores(dataset)
        if not by_id:
            return []
        dense = [missing] * (max(by_id) + 1)
        for index, score in by_id.items():
            dense[index] = score
        return dense

    def predict_scores(self, dataset: "CDataset") -> Dict[int, float]:
        self._require_init()
        dataset._require_init()
        response = _json_reply(_load().predict_scores(self.pointer, dataset.pointer))
        return dict((int(k), v) for k, v in response.items())

    def __del__(self):
        if getattr(self, "pointer", None) is not None and _lib is not None:
            _lib.free_model(self.pointer)
            self.pointer = None

    def _require_init(self):
        if self.pointer is None:
            raise ValueError("CModel is null!")

    def _query_json(self, message="to_json"):
        self._require_init()
        return _json_reply(_load().model_query_json(self.pointer, message.encode("utf-8")))

    def to_dict(self):
        return self._query_json("to_json")

    def __str__(self):
        return str(self.to_dict())


class CDataset:
    """A ranking dataset behind the C ABI: open_ranksvm() for files, from_numpy() for arrays.
    The first compute call uploads it to HBM (column-major features, query CSR)."""

    def __init__(self, pointer=None):
        self.pointer = pointer
        self.numpy_arrays_to_keep = []  # the library borrows these buffers (src/lib.rs:232-234)

    def __del__(self):
        if getattr(self, "pointer", None) is not None and _lib is not None:
            _lib.free_dataset(self.pointer)
            self.pointer = None
        self.numpy_arrays_to_keep = []

    @staticmethod
    def open_ranksvm(data_path, feature_names_path=None) -> "CDataset":
        names = feature_names_path.encode("utf-8") if feature_names_path is not None else None
        return CDataset(_unwrap(_load().load_ranksvm_format(data_path.encode("utf-8"), names)))

    @staticmethod
    def from_numpy(X, y, qid) -> "CDataset":
        import numpy as np

        (N, D) = X.shape
        assert N > 0
        assert D > 0
        assert len(y) == N
        assert len(qid) == N
        assert X.dtype == "float32"
        assert y.dtype == "float64"
        assert qid.dtype == "int64"
        # the ABI reads a C-contiguous row-major matrix; np.matrix (.todense()) and strided views
        # are normalised here instead of being mis-read
        X = np.ascontiguousarray(np.asarray(X))
        y = np.ascontiguousarray(np.asarray(y).reshape(-1))
        qid = np.ascontiguousarray(np.asarray(qid).reshape(-1))
        keep = [X, y, qid]
        dataset = CDataset(
            _unwrap(_load().make_dense_dataset_f32_f64_i64(N, D, X.ctypes.data, y.ctypes.data, qid.ctypes.data))
        )
        dataset.numpy_arrays_to_keep = keep
        return dataset

    def _require_init(self):
        if self.pointer is None:
            raise ValueError("Forgot to call open_* or from_numpy on CDataset!")

    def subsample_queries(self, queries: List[str]) -> "CDataset":
        self._require_init()
        actual_queries = self.queries()
        for q in queries:
            if q not in actual_queries:
                raise ValueError(
                    "Asked for query that does not exist in subsample: {0} not in {1}".format(q, actual_queries)
                )
        child = CDataset()
        child.numpy_arrays_to_keep = self.numpy_arrays_to_keep
        child.pointer = _unwrap(_load().dataset_query_sampling(self.pointer, json.dumps(queries).encode("utf-8")))
        return child

    def subsample_feature_names(self, features: List[str]) -> "CDataset":
        name_to_id = dict(zip(self._query_json("feature_names"), self._query_json("feature_ids")))
        fnums = sorted(set(name_to_id[f] for f in features))
        child = CDataset(_unwrap(_load().dataset_feature_sampling(self.pointer, json.dumps(fnums).encode("utf-8"))))
        child.numpy_arrays_to_keep = self.numpy_arrays_to_keep
        return child

    def train_model(self, train_req: "TrainRequest") -> CModel:  # noqa: F821
        self._require_init()
        request = json.dumps(train_req.to_dict()).encode("utf-8")
        return CModel(_unwrap(_load().train_model(request, self.pointer)), train_req)

    def _query_json(self, message="num_features"):
        self._require_init()
        return _json_reply(_load().dataset_query_json(self.pointer, message.encode("utf-8")))

    def is_sampled(self) -> bool:
        return self._query_json("is_sampled")

    def num_features(self) -> int:
        return self._query_json("num_features")

    def feature_ids(self) -> Set[int]:
        return set(self._query_json("feature_ids"))

    def feature_names(self) -> Set[str]:
        return set(self._query_json("feature_names"))

    def feature_index_to_name(self) -> Dict[int, str]:
        return dict(zip(self._query_json("feature_ids"), self._query_json("feature_names")))

    def feature_name_to_index(self) -> Dict[str, int]:
        return dict(zip(self._query_json("feature_names"), self._query_json("feature_ids")))

    def num_instances(self) -> int:
        return self._query_json("num_instances")

    def queries(self) -> Set[str]:
        return set(self._query_json("queries"))

    def instances_by_query(self) -> Dict[str, List[int]]:
        return self._query_json("instances_by_query")

    def evaluate(self, model: CModel, evaluator: str, qrel: CQRel = None) -> Dict[str, float]:
        self._require_init()
        model._require_init()
        qrel_pointer = None
        if qrel is not None:
            qrel._require_init()
            qrel_pointer = qrel.pointer
        return _json_reply(
            _load().evaluate_by_query(model.pointer, self.pointer, qrel_pointer, evaluator.encode("utf-8"))
        )

    def predict_scores(self, model: CModel) -> Dict[int, float]:
        return model.predict_scores(self)

    def predict_trecrun(self, model: CModel, output_path: str, system_name: str = "fastrank", quiet=True,
                        depth=0) -> int:
        self._require_init()
        model._require_init()
        response = _json_reply(
            _load().predict_to_trecrun(
                model.pointer, self.pointer, output_path.encode("utf-8"), system_name.encode("utf-8"), depth
            )
        )
        if not quiet:
            print("Wrote {} records to {} as {}.".format(response, output_path, system_name))
        return response


def query_json(message: str):
    """Global JSON query: 'coordinate_ascent_defaults' | 'random_forest_defaults'."""
    return _json_reply(_load().query_json(message.encode("utf-8")))
