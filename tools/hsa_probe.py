"""Diagnosis of 'ncclCommInitAll: unhandled cuda error' (RCCL's loader wrapper finds HSA not initialised) seen after
tests/test_gpu_fullsize.py::test_config4_256_restarts_as_8_shards_equal_the_unsharded_run in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, fastrank_amd as fr
from fastrank_amd import native
def probe(tag):
    hsa = ctypes.CDLL("libhsa-runtime64.so")
    v = ctypes.c_uint16(0)
    rc = hsa.hsa_system_get_info(0, ctypes.byref(v))
    print(tag, "hsa_system_get_info rc", rc, "major", v.value, flush=True)
n, d, q, seed = bench.SHAPES["30k"]
X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
g = fr.CDataset.from_numpy(X, y, qid)
req = fr.TrainRequest.coordinate_ascent(); req.measure = "ndcg@10"
p = req.params; p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 42, True, 256, 25
whole = native.train_model_shard(g, req, 0, 256); probe("whole")
parts = []
for rank in range(8):
    b, e = native.shard_bounds(256, rank, 8)
    parts.extend(native.train_model_shard(g, req, b, e)["restarts"]); probe("shard %d" % rank)
gathered = native.gather_restarts(parts, 256); probe("gather")
a = native.select_model(gathered, False).to_dict(); probe("select")
m = g.train_model(req).to_dict(); probe("train_model")
print(native.rccl_selftest(0))
