"""ctypes front-end for oracle/liboracle.so (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under fastrank_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

NDCG, AP, RR = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "fastrank_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class RFParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("num_trees", C.c_uint32),
        ("weight_trees", C.c_int32),
        ("split_method", C.c_int32),
        ("instance_sampling_rate", C.c_double),
        ("feature_sampling_rate", C.c_double),
        ("min_leaf_support", C.c_uint32),
        ("split_candidates", C.c_uint32),
        ("max_depth", C.c_uint32),
    ]


SPLIT_METHODS = {"SquaredError": 0, "BinaryGiniImpurity": 1, "InformationGain": 2, "TrueVarianceReduction": 3}


class CAParams(C.Structure):
    _fields_ = [
        ("num_restarts", C.c_uint32),
        ("num_max_iterations", C.c_uint32),
        ("step_base", C.c_double),
        ("step_scale", C.c_double),
        ("tolerance", C.c_double),
        ("seed", C.c_uint64),
        ("normalize", C.c_int32),
        ("init_random", C.c_int32),
        ("max_evals_per_restart", C.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, sz, i64, u32, u64, dbl = C.c_void_p, C.c_size_t, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    L.oracle_dataset_new.restype = vp
    L.oracle_dataset_new.argtypes = [sz, sz, vp, vp, vp]
    L.oracle_dataset_free.argtypes = [vp]
    L.oracle_dataset_set_presence.argtypes = [vp, vp]
    L.oracle_num_queries.restype = sz
    L.oracle_num_queries.argtypes = [vp]
    L.oracle_query_ids.argtypes = [vp, vp]
    L.oracle_query_offsets.argtypes = [vp, vp]
    L.oracle_query_docs.argtypes = [vp, vp]
    L.oracle_score_linear.argtypes = [vp, vp, sz, vp]
    L.oracle_score_ensemble.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.oracle_score_single_feature.argtypes = [vp, u32, dbl, vp]
    L.oracle_rank_order.argtypes = [sz, vp, vp, vp, vp]
    L.oracle_compute_dcg.restype = dbl
    L.oracle_compute_dcg.argtypes = [vp, sz, i64, C.c_int]
    L.oracle_ideal_dcg.restype = dbl
    L.oracle_ideal_dcg.argtypes = [vp, sz, i64]
    L.oracle_default_norms.argtypes = [vp, C.c_int, i64, vp]
    L.oracle_metric_from_scores.restype = C.c_int
    L.oracle_metric_from_scores.argtypes = [vp, C.c_int, i64, vp, vp, vp, vp]
    L.oracle_evaluate_mean_linear.restype = dbl
    L.oracle_evaluate_mean_linear.argtypes = [vp, C.c_int, i64, vp, vp, sz]
    L.oracle_ca_learn.restype = C.c_int
    L.oracle_ca_learn.argtypes = [vp, C.POINTER(CAParams), C.c_int, i64, vp, vp, sz, sz, u32, u32, vp, vp, vp]
    L.oracle_select_best.restype = u32
    L.oracle_select_best.argtypes = [vp, u32, u32]
    L.oracle_ca_candidates.restype = sz
    L.oracle_ca_candidates.argtypes = [dbl, dbl, dbl, u32, vp]
    L.oracle_set_mean_segment.argtypes = [sz]
    L.oracle_set_mean_shards.argtypes = [C.c_void_p, sz]
    L.oracle_set_mean_shards.restype = None
    L.oracle_get_mean_segment.restype = sz
    L.oracle_mean.restype = dbl
    L.oracle_mean.argtypes = [vp, sz]
    L.oracle_check_div_identity.restype = C.c_long
    L.oracle_check_div_identity.argtypes = [C.c_int]
    L.oracle_rand64_stream.argtypes = [u64, sz, vp]
    L.oracle_shuffle_with_seed.argtypes = [u64, vp, sz]
    L.oracle_rf_learn.restype = i64
    L.oracle_rf_learn.argtypes = [vp, C.POINTER(RFParams), C.c_int, i64, vp, vp, sz, vp, vp, vp, vp, sz, vp, vp, vp]
    L.oracle_resident_update.restype = None
    L.oracle_resident_update.argtypes = [vp, vp, sz, dbl, dbl, dbl]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def parse_measure(name: str):
    """src/evaluators.rs:132-155: 'ndcg', 'ndcg@K', 'ap'|'map', 'rr'|'mrr' (case-insensitive)."""
    depth = -1
    base = name
    if "@" in name:
        base, rhs = name.split("@", 1)
        depth = int(rhs)
    base = base.lower()
    kind = {"ndcg": NDCG, "ap": AP, "map": AP, "rr": RR, "mrr": RR}[base]
    return kind, depth


class Dataset:
    """DenseDataset restatement: borrowed row-major float32 X, float64 y, int64 qid."""

    def __init__(self, X, y, qid):
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float64)
        qid = np.ascontiguousarray(qid, dtype=np.int64)
        assert X.ndim == 2 and len(y) == X.shape[0] and len(qid) == X.shape[0]
        self.X, self.y, self.qid = X, y, qid
        self.n, self.d = X.shape
        self.ptr = lib().oracle_dataset_new(self.n, self.d, _p(X), _p(y), _p(qid))
        if not self.ptr:
            raise ValueError("qid out of u32 range")
        self.nq = lib().oracle_num_queries(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().oracle_dataset_free(self.ptr)
            self.ptr = None

    def set_presence(self, present):
        """File-loaded datasets (src/instance.rs:64-74): present[i, f] = False where instance i does not HOLD feature f
        (it reads 0.0 where a value is needed, and FeatureStats skips it: src/normalizers.rs:24-29).  None = all held."""
        if present is None:
            self._present = None
            lib().oracle_dataset_set_presence(self.ptr, None)
            return
        pm = np.ascontiguousarray(present, dtype=np.uint8)
        assert pm.shape == (self.n, self.d)
        self._present = pm  # (borrowed by the C side)
        lib().oracle_dataset_set_presence(self.ptr, _p(pm))

    def query_ids(self):
        out = np.zeros(self.nq, dtype=np.uint32)
        lib().oracle_query_ids(self.ptr, _p(out))
        return out

    def query_offsets(self):
        out = np.zeros(self.nq + 1, dtype=np.uint64)
        lib().oracle_query_offsets(self.ptr, _p(out))
        return out

    def query_docs(self):
        out = np.zeros(self.n, dtype=np.uint32)
        lib().oracle_query_docs(self.ptr, _p(out))
        return out

    def score_linear(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.zeros(self.n, dtype=np.float64)
        lib().oracle_score_linear(self.ptr, _p(w), len(w), _p(out))
        return out

    def score_single_feature(self, fid, direction):
        out = np.zeros(self.n, dtype=np.float64)
        lib().oracle_score_single_feature(self.ptr, fid, direction, _p(out))
        return out

    def score_ensemble(self, trees, weights):
        """trees: list of nested dicts in the reference's JSON shape
        ({"FeatureSplit": {...}} | {"LeafNode": v}); weights: per-tree f64."""
        fid, split, lhs, rhs, roots = [], [], [], [], []

        def add(node):
            idx = len(fid)
            fid.append(0), split.append(0.0), lhs.append(-1), rhs.append(-1)
            if "LeafNode" in node:
                fid[idx] = -1
                split[idx] = float(node["LeafNode"])
            else:
                fs = node["FeatureSplit"]
                fid[idx] = int(fs["fid"])
                split[idx] = float(fs["split"])
                lhs[idx] = add(fs["lhs"])
                rhs[idx] = add(fs["rhs"])
            return idx

        for t in trees:
            roots.append(add(t))
        roots = np.asarray(roots, dtype=np.int32)
        tw = np.ascontiguousarray(weights, dtype=np.float64)
        fid = np.asarray(fid, dtype=np.int32)
        split = np.asarray(split, dtype=np.float64)
        lhs = np.asarray(lhs, dtype=np.int32)
        rhs = np.asarray(rhs, dtype=np.int32)
        out = np.zeros(self.n, dtype=np.float64)
        lib().oracle_score_ensemble(self.ptr, len(roots), _p(roots), _p(tw), _p(fid), _p(split), _p(lhs), _p(rhs), _p(out))
        return out

    def default_norms(self, measure: str):
        kind, depth = parse_measure(measure)
        out = np.zeros(self.nq, dtype=np.float64)
        lib().oracle_default_norms(self.ptr, kind, depth, _p(out))
        return out

    def qrel_norms(self, measure: str, qrel: dict):
        """Norms when judgments are supplied (src/evaluators.rs:310-333, 397-413;
        src/qrel.rs:18-39): NDCG ideal from the qrel's positive gains, AP num_relevant from
        the qrel's positive count (0 -> dataset fallback), for queries the qrel knows."""
        kind, depth = parse_measure(measure)
        norms = self.default_norms(measure)
        for k, q in enumerate(self.query_ids()):
            judged = qrel.get(str(int(q)))
            if judged is None:
                continue
            pos = np.asarray([g for g in judged.values() if np.float32(g) > 0], dtype=np.float32)
            if kind == NDCG:
                norms[k] = lib().oracle_ideal_dcg(_p(pos), len(pos), depth) if len(pos) else float("nan")
            elif kind == AP:
                if len(pos) > 0:
                    norms[k] = float(len(pos))
        return norms

    def metric_from_scores(self, measure: str, scores, norms=None, want_rank=False):
        kind, depth = parse_measure(measure)
        if norms is None:
            norms = self.default_norms(measure)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        norms = np.ascontiguousarray(norms, dtype=np.float64)
        out = np.zeros(self.nq, dtype=np.float64)
        rank = np.zeros(self.n, dtype=np.uint32) if want_rank else None
        err = lib().oracle_metric_from_scores(self.ptr, kind, depth, _p(scores), _p(norms), _p(out),
                                              _p(rank) if want_rank else None)
        if want_rank:
            return out, rank, err
        return out, err

    def evaluate_by_query(self, measure: str, w, norms=None):
        vals, err = self.metric_from_scores(measure, self.score_linear(w), norms)
        return dict(zip((str(int(q)) for q in self.query_ids()), vals.tolist()))

    def evaluate_mean(self, measure: str, w, norms=None):
        kind, depth = parse_measure(measure)
        if norms is None:
            norms = self.default_norms(measure)
        w = np.ascontiguousarray(w, dtype=np.float64)
        norms = np.ascontiguousarray(norms, dtype=np.float64)
        return lib().oracle_evaluate_mean_linear(self.ptr, kind, depth, _p(norms), _p(w), len(w))

    def rf_learn(self, measure: str, params: dict, fids=None, norms=None):
        """src/random_forest.rs:288-342.  params: the reference's RandomForestParams as a dict (split_method: the
        variant's name, or the serde form {"SquaredError": []}).  Returns (trees as nested dicts in the reference's
        JSON shape, weights, per-tree (n_features, n_instances) of the sample)."""
        kind, depth = parse_measure(measure)
        if norms is None:
            norms = self.default_norms(measure)
        norms = np.ascontiguousarray(norms, dtype=np.float64)
        if fids is None:
            fids = np.arange(self.d, dtype=np.uint32)
        fids = np.ascontiguousarray(fids, dtype=np.uint32)
        sm = params.get("split_method", "SquaredError")
        if isinstance(sm, dict):
            sm = next(iter(sm))
        p = RFParams(int(params["seed"]), int(params["num_trees"]), int(bool(params.get("weight_trees", False))),
                     SPLIT_METHODS[sm], float(params["instance_sampling_rate"]), float(params["feature_sampling_rate"]),
                     int(params["min_leaf_support"]), int(params["split_candidates"]), int(params["max_depth"]))
        nt = p.num_trees
        cap = max(16, nt * (2 * self.n + 1))
        cap = min(cap, nt * (2 ** min(int(p.max_depth), 20) + 1))
        fid = np.zeros(cap, dtype=np.int32)
        split = np.zeros(cap, dtype=np.float64)
        lhs = np.zeros(cap, dtype=np.int32)
        rhs = np.zeros(cap, dtype=np.int32)
        roots = np.zeros(max(1, nt), dtype=np.int32)
        weights = np.zeros(max(1, nt), dtype=np.float64)
        sample = np.zeros((max(1, nt), 2), dtype=np.uint32)
        n = lib().oracle_rf_learn(self.ptr, C.byref(p), kind, depth, _p(norms), _p(fids), len(fids), _p(fid), _p(split),
                                  _p(lhs), _p(rhs), cap, _p(roots), _p(weights), _p(sample))
        if n < 0:
            raise RuntimeError("oracle_rf_learn: error %d (1 = node capacity, 2 = the reference would have panicked)" % -n)

        def build(k):
            if fid[k] < 0:
                return {"LeafNode": float(split[k])}
            return {"FeatureSplit": {"fid": int(fid[k]), "split": float(split[k]), "lhs": build(int(lhs[k])), "rhs": build(int(rhs[k]))}}

        return [build(int(r)) for r in roots[:nt]], weights[:nt].copy(), sample[:nt].copy()

    def ca_learn(self, measure: str, params: dict, fids=None, threads=1, norms=None,
                 restart_range=None, max_evals_per_restart=0):
        """Returns (scores[R], weights[R, dim], evals[R], err)."""
        kind, depth = parse_measure(measure)
        if norms is None:
            norms = self.default_norms(measure)
        norms = np.ascontiguousarray(norms, dtype=np.float64)
        if fids is None:
            fids = np.arange(self.d, dtype=np.uint32)
        fids = np.ascontiguousarray(fids, dtype=np.uint32)
        dim = int(fids.max()) + 1
        p = CAParams(
            int(params["num_restarts"]), int(params["num_max_iterations"]), float(params["step_base"]),
            float(params["step_scale"]), float(params["tolerance"]), int(params["seed"]),
            int(bool(params["normalize"])), int(bool(params["init_random"])), int(max_evals_per_restart),
        )
        R = p.num_restarts
        b, e = restart_range if restart_range is not None else (0, R)
        scores = np.full(R, np.nan, dtype=np.float64)
        weights = np.zeros((R, dim), dtype=np.float64)
        evals = np.zeros(R, dtype=np.uint64)
        err = lib().oracle_ca_learn(self.ptr, C.byref(p), kind, depth, _p(norms), _p(fids), len(fids),
                                    threads, b, e, _p(scores), _p(weights), _p(evals))
        return scores, weights, evals, err


DEVICE_MEAN_SEGMENT = 256  # fastrank_amd/csrc/device.hip MEAN_SEG


def set_mean_shards(starts=None) -> None:
    """Query-sharded mean shape: starts[r] = index (in dataset query order) of shard r's first query.
    None / [] turns it off."""
    arr = np.ascontiguousarray(starts if starts is not None else [], dtype=np.uint64)
    lib().oracle_set_mean_shards(arr.ctypes.data_as(C.c_void_p) if arr.size else None, int(arr.size))


def set_mean_segment(seg: int) -> None:
    """0 = one sequential pass (default); 256 = the HIP path's two-level summation shape."""
    lib().oracle_set_mean_segment(int(seg))


def mean(values) -> float:
    values = np.ascontiguousarray(values, dtype=np.float64)
    return lib().oracle_mean(_p(values), len(values))


def rank_order(scores, gains, ids):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    gains = np.ascontiguousarray(gains, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(len(ids), dtype=np.uint32)
    lib().oracle_rank_order(len(ids), _p(scores), _p(gains), _p(ids), _p(out))
    return out


def compute_dcg(gains, depth=None, ideal=False):
    gains = np.ascontiguousarray(gains, dtype=np.float32)
    return lib().oracle_compute_dcg(_p(gains), len(gains), -1 if depth is None else depth, int(ideal))


def select_best(scores):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    return int(lib().oracle_select_best(_p(scores), 0, len(scores)))


def ca_candidates(orig, step_base, step_scale, iters):
    out = np.zeros(1 + 2 * iters, dtype=np.float64)
    n = lib().oracle_ca_candidates(orig, step_base, step_scale, iters, _p(out))
    return out[:n]


def rand64_stream(seed, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().oracle_rand64_stream(seed, n, _p(out))
    return out


def shuffle_with_seed(seed, n):
    v = np.arange(n, dtype=np.uint32)
    lib().oracle_shuffle_with_seed(seed, _p(v), n)
    return v


def check_div_identity(maxb: int) -> int:
    """Pairs 1 <= a <= b <= maxb where the 2-FMA quotient of kernels_fullverify.inc differs from a / b."""
    return int(lib().oracle_check_div_identity(int(maxb)))


def resident_update(R, xf, cand, base_f, inv):
    """In place: R <- fma(x_f, cand, fma(-x_f, base_f, R * inv)), the device's incremental update of a resident
    sum (test helper, see fastrank_oracle.c)."""
    assert R.dtype == np.float64 and xf.dtype == np.float32 and R.flags.c_contiguous and xf.flags.c_contiguous
    lib().oracle_resident_update(_p(R), _p(xf), len(R), float(cand), float(base_f), float(inv))
