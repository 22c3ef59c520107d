// HIP kernels + launchers for the fastrank hot path on MI355X (gfx950, wave64).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contract=off is REQUIRED: the
// reference's dot product is an unfused f64 multiply-then-add, src/dense_dataset.rs:71-74).
//
// One translation unit, split into parts for reading:
//   device_plumbing.inc     HIP error macro, HIP-event kernel timing, device buffers
//   kernels_score.inc       tile layout (xb_index) + score_linear_kernel (exact ordered f64 dot
//                           product, lane = document), single-feature / axpy helpers
//   kernels_tree.inc        tree_ensemble_lds_kernel (forest streamed through LDS, 8 walks per lane)
//   kernels_treerank.inc    tree_ensemble_rank_kernel (threshold ranks instead of features: u16 codes, 32-bit heap nodes)
//                           and tree_ensemble_kernel (L2 fallback)
//   kernels_metric.inc      metric_sort_kernel (LDS bitonic sort + NDCG/AP/RR in the reference's
//                           summation order) and the fixed-shape mean reduction
//   kernels_linesearch.inc  linesearch_ndcg_kernel: the exact fused NDCG@k line search
//   kernels_verify.inc      linesearch_verify_kernel: bound-and-verify NDCG@k line search (the hot path; the exact
//                           kernel recomputes the pairs it cannot verify)
//   kernels_order.inc       xslot_kernel / rslot_kernel: the verify kernel's visiting-order tables (by x_f, by the resident sums)
//   kernels_fullrank.inc    linesearch_scores_kernel + rank_metric_kernel: AP / RR / depth-less NDCG
//   kernels_rr.inc          rr_verify_kernel / rr_exact_kernel: reciprocal rank by bound-and-verify
//   fullverify.hpp          interface to fullverify.hip (own objects, compiled per slice of the size classes):
//     kernels_sortnet.inc     (generated, tools/gen_sortnet.py) compare-exchange networks over register-resident keys
//     kernels_fullverify.inc  fullrank_verify_kernel: NDCG of any depth / AP by sorting approximate keys in registers
//                             and verifying the gaps (the exact kernels of kernels_fullrank.inc redo what fails)
//   kernels_rf.inc          random-forest TRAINING: level-synchronous split search over a batch of trees (rocPRIM radix sort +
//                           sequential-association importance kernels)
//   device_dataset.inc      DeviceDataset: HBM layout (runs, tiles, tables) and every launcher
#include "device.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "fullverify.hpp"

#include <cstring>
#include <dlfcn.h>
#include <unistd.h>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>

namespace frdev {

#include "device_plumbing.inc"
#include "kernels_score.inc"
#include "kernels_tree.inc"
#include "kernels_treerank.inc"
#include "kernels_metric.inc"
#include "kernels_linesearch.inc"
#include "kernels_chain.inc"
#include "kernels_order.inc"
#include "kernels_verify.inc"
#include "kernels_fullrank.inc"
#include "kernels_rr.inc"
#include "kernels_rf.inc"
#include "device_dataset.inc"
#include "rccl_exchange.inc"

}  // namespace frdev
