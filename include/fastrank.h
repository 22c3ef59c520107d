/*
 * fastrank.h -- C ABI of libfastrank_amd.so, the MI355X-native drop-in for the fastrank cdylib.
 *
 * Part 1 re-declares, symbol for symbol, the `extern "C"` surface the reference exports from
 * src/lib.rs (file:line cited per entry, paths relative to the jjfiv/fastrank tree) and that
 * fastrank/clib.py binds through cffi.  Semantics (ownership, JSON envelopes, error strings)
 * follow src/ffi.rs; see INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Part 2 declares extensions (prefix `fr_`) that have no reference counterpart: device
 * selection, restart sharding for multi-GPU runs, dense (non-JSON) result buffers, direct
 * access to the batched line-search evaluator, and HIP-event kernel timing for bench.py.
 *
 * Strings: NUL-terminated UTF-8.  Every `const void*` / `const char*` JSON return value is
 * heap-allocated by the library and must be released with free_str().  A CResult is released
 * with free_c_result() (non-recursive: release error_message with free_str(), and the success
 * handle later with free_dataset()/free_model()/free_cqrel()).  Exactly one of the two CResult
 * fields is non-NULL.  Errors never use errno or exceptions: they are the JSON envelope
 * {"error":"error","context":"<Rust Debug form of the message>"} (src/ffi.rs:22-26,45-74).
 *
 * All ranking arithmetic runs on the GPU; there is no CPU fallback.  Calls that need the
 * device return an error envelope when no MI355X/HIP device is available.
 */
#ifndef FASTRANK_AMD_H
#define FASTRANK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:50-61 -- opaque handles */
typedef struct CDataset CDataset;
typedef struct CModel CModel;
typedef struct CQRel CQRel;

/* src/lib.rs:63-67 */
typedef struct CResult {
    const void *error_message;
    const void *success;
} CResult;

/* ---------------------------------------------------------------------------------------- */
/* Part 1: the reference surface                                                            */
/* ---------------------------------------------------------------------------------------- */

/* src/lib.rs:78-81 */
void free_str(void *originally_from_library);
/* src/lib.rs:85-91 (non-recursive) */
void free_c_result(CResult *originally_from_library);
/* src/lib.rs:95-98 */
void free_dataset(CDataset *originally_from_library);
/* src/lib.rs:102-105 */
void free_model(CModel *originally_from_library);
/* src/lib.rs:109-112 */
void free_cqrel(CQRel *originally_from_library);

/* src/lib.rs:117-121: TREC qrel file -> CQRel */
const CResult *load_cqrel(const void *data_path);
/* src/lib.rs:126-131: {"qid":{"docid":gain}} -> CQRel */
const CResult *cqrel_from_json(const void *json_str);
/* src/lib.rs:136-142: "to_json" | "queries" | <qid> */
const void *cqrel_query_json(const CQRel *cqrel, const void *query_str);

/* src/lib.rs:147-162: ranksvm/libsvm text file (+ optional feature-name JSON) -> CDataset */
const CResult *load_ranksvm_format(void *data_path, void *feature_names_path_or_null);
/* src/lib.rs:167-178: JSON list of qid strings -> sampled CDataset */
const CResult *dataset_query_sampling(CDataset *dataset, const void *queries_json_list);
/* src/lib.rs:183-197: JSON list of feature ids -> sampled CDataset */
const CResult *dataset_feature_sampling(CDataset *dataset, const void *feature_json_list);
/* src/lib.rs:202-211: is_sampled | num_features | feature_ids | num_instances | queries |
 * instances_by_query | feature_names */
const void *dataset_query_json(void *dataset, void *json_cmd_str);
/* src/lib.rs:216-218: coordinate_ascent_defaults | random_forest_defaults */
const void *query_json(const void *json_cmd_str);

/* src/lib.rs:224-240: borrowed row-major f32 X[n*d], f64 y[n], i64 qid[n].  The caller keeps
 * the arrays alive for the lifetime of the dataset and of every dataset sampled from it. */
const CResult *make_dense_dataset_f32_f64_i64(size_t n, size_t d, const float *x, const double *y,
                                              const int64_t *qids);

/* src/lib.rs:245-253: TrainRequest JSON -> CModel.  The coordinate-ascent line search runs as
 * batched HIP launches (src/coordinate_ascent.rs:87-254); RandomForest requests are trained on the device too,
 * level-synchronously over batches of trees (src/random_forest.rs:211-408; csrc/rf_train.hpp, kernels_rf.inc).
 * With several devices visible (FR_DEVICES, default: all) the restarts of a coordinate-ascent request are spread
 * over them inside this call, like the reference's rayon fan-out over restarts (src/coordinate_ascent.rs:215-225),
 * and so are the trees of a random-forest request (src/random_forest.rs:301-331). */
const CResult *train_model(void *train_request_json, void *dataset);
/* src/lib.rs:258-263 */
const CResult *model_from_json(const void *json_str);
/* src/lib.rs:268-277: "to_json" */
const void *model_query_json(const void *model, const void *json_cmd_str);

/* src/lib.rs:283-294: {qid: metric}; qrel may be NULL */
const void *evaluate_by_query(const CModel *model, const CDataset *dataset, const CQRel *qrel,
                              const void *evaluator_name);
/* src/lib.rs:299-303: {"<instance index>": score} */
const void *predict_scores(const CModel *model, const CDataset *dataset);
/* src/lib.rs:308-326: writes a TREC run file; JSON number of records written */
const void *predict_to_trecrun(const CModel *model, const CDataset *dataset, const void *output_path,
                               const void *system_name, size_t depth);

/* ---------------------------------------------------------------------------------------- */
/* Part 2: extensions (no reference counterpart)                                            */
/* ---------------------------------------------------------------------------------------- */

/* Number of visible HIP devices (0 when none / no driver). */
int fr_device_count(void);
/* Select the device used by datasets created afterwards on this thread. 0 on success.  The first call for a device also
 * pays the runtime's first-stream cost there (~80 ms once per process), so that the first dataset built on it does not. */
int fr_set_device(int ordinal);
/* Library/ABI version string (static storage; do NOT free). */
const char *fr_version(void);

/* Restart-sharded coordinate ascent for one-process-per-GPU runs: trains restarts
 * [restart_begin, restart_end) of the request's num_restarts (child seeds are still drawn in
 * order from the master RNG, src/coordinate_ascent.rs:211-213) and returns JSON
 * {"restarts":[{"restart_id":r,"score":s,"weights":[...]}...],"stats":{...}}.
 * The caller gathers shards (RCCL all_gather) and applies fr_select_model(). */
const void *fr_train_model_shard(const void *train_request_json, const CDataset *dataset,
                                 uint32_t restart_begin, uint32_t restart_end);
/* Steppable form of the same trainer (what bench.py times): begin -> step(k ticks)* -> state.
 * One tick = one fused launch evaluating every line-search candidate of every live restart.
 * fr_ca_begin returns NULL and sets *error_out (free_str) on failure. */
void *fr_ca_begin(const void *train_request_json, const CDataset *dataset, uint32_t restart_begin,
                  uint32_t restart_end, const void **error_out);
/* Query-sharded form (fewer restarts than GPUs).  `dataset` holds this rank's block of
 * the queries; all ranks run ALL restarts in lock step.  After every device evaluation the trainer
 * hands the per-candidate SUMS over its queries to `allreduce(ctx, values, n)`, which must replace
 * values[i] with the sum over all ranks added in rank order (identical bits on every rank; e.g.
 * all_gather + a sequential add) and return 0; the trainer divides by total_queries.  Step and read
 * it with fr_ca_step / fr_ca_state; every rank ends with the same restarts. */
typedef int (*fr_allreduce_sum_fn)(void *ctx, double *values, size_t n);
void *fr_ca_begin_query_shard(const void *train_request_json, const CDataset *dataset, uint64_t total_queries,
                              fr_allreduce_sum_fn allreduce, void *ctx, const void **error_out);
/* Runs up to max_ticks ticks; NULL on success. *finished = 1 once every restart converged.  Inside the call the
 * restarts are stepped as a few sets with one line search of each in flight on the device; everything submitted has
 * been collected and applied when the call returns, so the state is the lock-step state after *ticks_done ticks. */
const void *fr_ca_step(void *trainer, uint64_t max_ticks, uint64_t *ticks_done, int *finished);
/* JSON {"restarts":[...],"stats":{...},"finished":bool} (free_str). */
const void *fr_ca_state(void *trainer);
void fr_ca_free(void *trainer);

/* Selection rule of src/coordinate_ascent.rs:232-252 over gathered restarts.
 * restarts_json: JSON list of {"restart_id","score","weights"}; returns a CModel. */
const CResult *fr_select_model(const void *restarts_json, int output_ensemble);
/* JSON stats of the most recent train_model / fr_train_model_shard call in this process:
 * {"useful_evals","raw_evals","ticks","groups","seconds","path","restarts","verify_pairs","verify_redone",
 *  "exact_ticks","exact_groups","verify_redo_entries","line_searches","audit_values","audit_mismatches","devices",
 *  "refills","chain_runs","chain_visits","rank_slots_on","rank_slots_off"} (chain_*: documents that made the NDCG@k verify
 * kernel run its insertion chain, out of (document, group) visits -- the kernel's own counter; rank_slots_*: restarts whose
 * R-rank upkeep that counter switched on again / off) and, after a call that spread over several devices, "rccl": the report
 * of the exchange (fr_rccl_allgather below); and, after a call that ran
 * on several devices, "per_device": the same object for every entry of the device list (its "device", "restarts",
 * "ticks", "seconds", "refills": times converged restarts handed their places to ids from the shared restart queue)
 * (path: "fused_linesearch" | "fused_fullrank" | "generic_sort"; verify_*: (query, group) pairs evaluated by the
 * bound-and-verify kernels and how many of them were recomputed by the exact kernels -- NDCG@k recomputes only the
 * 16-candidate slices of a pair that hold an undecided candidate: verify_redo_entries counts those; exact_ticks of
 * line_searches batched line searches went to the exact kernels alone, exact_groups of groups single restarts' line
 * searches were routed there while the rest of their tick stayed on the verify kernel; audit_*: with FR_VERIFY_AUDIT=1 every published NDCG@k
 * value is recomputed by the exact kernel and compared bit for bit -- values compared / values that differed). */
const void *fr_last_train_stats(void);

/* Dense results without JSON.  out[i] = score of instance i (instances outside the dataset
 * view are left untouched).  Returns NULL on success or an error-envelope string. */
const void *fr_predict_scores_dense(const CModel *model, const CDataset *dataset, double *out, size_t out_len);
/* Per-query metric in the dataset's device query order.  out_values[nq]; out_qids (optional)
 * receives a JSON list of the qid strings in the same order via *out_qids_json (free_str). */
const void *fr_evaluate_dense(const CModel *model, const CDataset *dataset, const CQRel *qrel,
                              const void *evaluator_name, double *out_values, size_t out_len,
                              const void **out_qids_json);
/* Full per-query rank order under the reference's total order (src/evaluators.rs:34-49):
 * out_instance_ids[n] grouped by query (device query order), best first; out_offsets[nq+1]. */
const void *fr_rank_order(const CModel *model, const CDataset *dataset, uint32_t *out_instance_ids,
                          size_t n, uint64_t *out_offsets, size_t nq_plus_1);
/* JSON about the dataset's device form (built on first use): {"hbm_bytes_owned","shares_parent_matrix",
 * "is_parent_device_dataset","queries","instances"}.  A view made by dataset_query_sampling /
 * dataset_feature_sampling does not tile a second copy of X: a feature sample uses its parent's device dataset as it is,
 * a query sample owns only query / run tables over the parent's tiles (src/dataset.rs:101-178 keeps views as id lists
 * over the parent for the same reason). */
const void *fr_dataset_device_info(const CDataset *dataset);
/* Number of queries / instances in the dataset view (0 for a NULL handle).  fr_dataset_num_queries returns SIZE_MAX
 * when the view cannot be grouped (e.g. a NaN label): the compute calls report the reason in their envelope. */
size_t fr_dataset_num_queries(const CDataset *dataset);
size_t fr_dataset_num_instances(const CDataset *dataset);

/* The hot operator itself: batched evaluate_mean of line-search candidates
 * (src/coordinate_ascent.rs:157-160 x src/evaluators.rs:173-184).
 *   n_groups line groups; group g shares base weights base_weights[g*d .. g*d+d) and feature
 *   features[g]; it has n_cand[g] (<=64) candidate values candidates[g*64 + c] for that
 *   feature's weight.  out_means[g*64 + c] receives the mean metric.  If out_per_query is
 *   non-NULL it receives the [nq][n_groups*64] per-query matrix.
 * Uses the fused HIP line-search kernel for ndcg@k (k<=20) and the general sort kernel
 * otherwise.  Returns NULL on success or an error-envelope string. */
const void *fr_evaluate_candidates(const CDataset *dataset, const CQRel *qrel, const void *evaluator_name,
                                   size_t n_groups, const uint32_t *features, const double *base_weights,
                                   const uint32_t *n_cand, const double *candidates, double *out_means,
                                   double *out_per_query);

/* HIP-event timing of the library's kernels on the stream they are launched on. */
void fr_profile_enable(int on);
void fr_profile_reset(void);
/* JSON list [{"kernel","launches","total_ms"}] (free_str). */
const void *fr_profile_json(void);
/* hipDeviceSynchronize; 0 on success. */
int fr_synchronize(void);
/* The constants of the resident-sum error bound the trainer and the bound-and-verify kernels use (DESIGN.md
 * section 4.2), so that a CPU test can replay the device's update arithmetic against extended precision with the
 * product's own numbers.  which = 0: bound after an exact refresh (a = feature count, b = T);
 * 1: bound after one incremental update (a = previous bound, b = norm, c = T);
 * 2: the term a candidate key's error bound gains from the resident form (a = bound, b = norm, c = T).
 * No device needed.  NaN for an unknown `which`. */
double fr_debug_resident_bound(int which, double a, double b, double c);
/* Size class of the full-ranking bound-and-verify kernel (depth-less NDCG, MAP, NDCG@>20) for a query of `len` documents:
 * (keys per lane << 16) | lanes per candidate.  No device needed (the CPU tests check that every length has a class that
 * holds it and that classes grow with the length). */
uint32_t fr_debug_fullrank_class(uint32_t len);
/* The device plan of a request with `num_restarts` units of work (coordinate-ascent restarts, random-forest trees) over the
 * devices listed in `devices_csv` (the syntax of the FR_DEVICES environment variable: comma-separated ordinals, the same
 * ordinal twice = two contexts on that device) on a node with `device_count` devices, when the dataset's first device
 * form lives on `primary_device`: JSON {"devices": [...], "slots": [...], "blocks": [[begin, end], ...]} -- which
 * device-side copy every entry trains on and the contiguous block of ids it gets under a static partition (random
 * forests, and coordinate ascent with FR_RESTART_QUEUE=0; by default the entries of a coordinate-ascent request pull
 * restart ids from one shared queue instead -- the reference fans restarts out with rayon,
 * src/coordinate_ascent.rs:215-225).  No device needed; errors come back in the usual envelope. */
const void *fr_debug_device_plan(const void *devices_csv, int device_count, uint32_t num_restarts, int primary_device);
/* Replays the restart queue of train_model on the CPU (no device): `n_workers` trainers with `capacity` live restarts
 * each take the ids 0 .. num_restarts-1 from one queue; restart r is given a length of lengths[r % n_lengths] ticks and a
 * worker refills converged places at the end of a tick.  JSON {"order": [[ids in the order worker w started them] ...],
 * "ticks": [...]} -- the CPU tests check that every id is started exactly once and that a worker never holds more than
 * `capacity`. */
const void *fr_debug_restart_queue(uint32_t num_restarts, uint32_t n_workers, uint32_t capacity, const uint32_t *lengths,
                                   uint32_t n_lengths);
/* One timed device-to-device copy of `bytes` bytes from device `src_device` to device `dst_device`, made the way
 * train_model's fan-out copies a dataset to another GPU (peer access enabled where the link allows, hipMemcpyPeerAsync):
 * JSON {"src","dst","bytes","can_access","enabled","ms","gbps"}.  A first-contact check of the peer path on a multi-GPU
 * node (bench.py --gpus N reports it per device pair); src == dst measures a copy inside one device. */
const void *fr_debug_peer_copy(int src_device, int dst_device, size_t bytes);
/* The exchange records of a job's single all-gather (what src/coordinate_ascent.rs:232-252 selects from): a
 * rank's restarts as a fixed-size block of `cap` records of 3 + dim doubles -- valid (1.0; 0.0 = padding), restart id, score,
 * weights[dim].  fr_pack_restart_records writes the block for a JSON list of {"restart_id","score","weights"} (returns NULL
 * or an error envelope); fr_unpack_restart_records returns the valid records of n_records records as such a list, in
 * restart order.  Same layout as native.gather_restarts (torch.distributed) and train_model's own fan-out. */
const void *fr_pack_restart_records(const void *restarts_json, size_t cap, size_t dim, double *out);
const void *fr_unpack_restart_records(const double *records, size_t n_records, size_t dim);
/* ONE single-process RCCL all-gather over `devices` (ncclCommInitAll over the list, a grouped ncclAllGather; librccl.so is
 * dlopen-ed, no link-time dependency): rank i contributes blocks[i * block_len ..], `out` (optional, n * block_len doubles)
 * receives what rank 0 gathered.  JSON {"ran":true,"ranks","devices","us","first_us","init_us","matches_host_gather"}, or
 * {"ran":false,"reason"} when it does not apply (one rank; ranks that share a GPU).  With n distinct GPUs a failure is an
 * error envelope, not a fallback.  train_model's multi-device path runs it on the restarts' records itself
 * (fr_last_train_stats: "rccl"). */
const void *fr_rccl_allgather(const int *devices, size_t n, const double *blocks, size_t block_len, double *out);
/* The walk tiles of the resident NDCG@k verify kernel (csrc/device.hpp: build_walk_tiles) for a layout of runs and queries,
 * computed on the host alone: JSON {"walk_tile":128,"wt_start":[...],"run_wt0":[...],"seg":[...],"wofs":[...]}. */
const void *fr_debug_walk_tiles(const uint32_t *run_pos, const uint32_t *run_q0, const uint32_t *run_q1, size_t nruns,
                                const uint32_t *qstart, const uint32_t *qlen, size_t nq, size_t np);
/* The same call sequence with a one-rank communicator on `device` (what a one-GPU box can run of it): same JSON. */
const void *fr_debug_rccl_selftest(int device);
/* Frees the device-to-device copies train_model made of this dataset on other devices / in other contexts (they are kept
 * with the dataset so that the next request reuses them; a node shared with other jobs may want the HBM back).  The
 * dataset's first device form stays.  Returns the number of copies released.  Views sampled from the dataset
 * (dataset_query_sampling, train_test_split ...) that were trained on several devices hold copies of their own -- over
 * the same copied matrix -- and are released the same way, each through its own handle. */
size_t fr_dataset_release_replicas(const CDataset *dataset);

#ifdef __cplusplus
}
#endif
#endif /* FASTRANK_AMD_H */
