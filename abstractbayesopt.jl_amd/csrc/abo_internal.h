// Internal C++ interface between api.hip (single-device handles) and mgpu.hip (the multi-device driver).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>

#include "../../include/abo_hip.h"

namespace abo {

// abo_acq / abo_cand_acq with separate memory spaces for the M scores and for the k selected (score, index) pairs
int32_t acq_ex(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, int32_t kind, double p0, double best_y,
               int64_t idx_base, double* scores, int32_t scores_space, int32_t k, double* top_val, int64_t* top_idx,
               int32_t top_space);
int32_t cand_acq_ex(abo_gp* g, abo_cand* c, int32_t kind, double p0, double best_y, int64_t idx_base, double* scores,
                    int32_t scores_space, int32_t k, double* top_val, int64_t* top_idx, int32_t top_space);

int32_t acq_terms_ex(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, const abo_acq_term* terms, int32_t nterms,
                     int64_t idx_base, double* scores, int32_t scores_space, int32_t k, double* top_val, int64_t* top_idx,
                     int32_t top_space);
// a shard's part of the grid stage (acq_utils.jl:44-52): rows j0 … j0+count−1 of the n-point Latin hypercube generated into grid_d
// (device, count·d doubles), scored under the objective, the k best (score, global index) pairs left in tv_d / ti_d (device)
int32_t acq_lhs_shard(abo_gp* g, const abo_acq_term* terms, int32_t nterms, int64_t n, int32_t d, const double* lower,
                      const double* upper, uint64_t seed, int64_t j0, int64_t count, int32_t k, double* grid_d, double* tv_d,
                      int64_t* ti_d);
int32_t refine_terms(abo_gp* g, const abo_acq_term* terms, int32_t nterms, const double* lower, const double* upper, int32_t d,
                     const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out);

// optimize_acquisition's last step on the host (acq_utils.jl:66-72): the refined point with the largest finite value (first on
// ties), or the best grid point when no refined value reaches its score
void pick_best_point(const double* starts_x, const double* starts_val, const double* rx, const double* rf, int k, int d,
                     double* best_x, double* best_val);

// ---- greedy q-EI, block form (api.hip; include/abo_hip.h: abo_cand_qei_*): the per-shard steps with records in DEVICE memory, and
// the batch driver over the n shards of one set (n = 1: abo_cand_qei; n > 1: abo_mgpu_cand_qei, whose shards run on their worker
// threads and exchange records through the group's all-gather)
int32_t qei_eligible(abo_gp* g, abo_cand* c, int q);       // ABO_OK when the block form can run on this shard (else the reason)
int32_t qei_begin(abo_gp* g, abo_cand* c, int q, int T, bool snapshot = true);
int32_t qei_top(abo_gp* g, abo_cand* c, double xi, double best_y, int64_t idx_base, int k, double* rec_d);
int32_t qei_block(abo_gp* g, abo_cand* c, const double* pts, const int64_t* gidx, int T);
int32_t qei_has(const abo_cand* c, int64_t gidx);
int32_t qei_pick(abo_gp* g, abo_cand* c, int64_t gidx, double var_x, const double* cx, int n, int64_t excl, int64_t* info);
int32_t qei_end(abo_gp* g, abo_cand* c);
void qei_get_stats(const abo_cand* c, int picks, double total_ms, abo_qei_stats* out);
struct QeiShards {
    int n;
    abo_gp* const* gp;
    abo_cand* const* cd;
    const int64_t* lo;                                               // global index of each shard's first candidate
    std::function<int32_t(const std::function<int32_t(int)>&)> run;  // f(i) on every shard
    std::function<double*(int)> rec;                                 // shard i's record block (device)
    std::function<int32_t(size_t, double*)> gather;                  // (words, out): every shard's block → host, n × words doubles
};
int qei_block_default();                                            // the process default block size (0: the plain loop)
size_t qei_max_words(int d, int q, int T);                           // doubles a record block must hold
int32_t qei_drive(const QeiShards& S, int q, double xi, double best_y, int distinct, int T, double* x_out, int64_t* idx_out,
                  double* ei_out, int64_t* info);

hipStream_t gp_stream(abo_gp* g);
int gp_device(const abo_gp* g);
const abo_params& gp_params(const abo_gp* g);
int gp_dim(const abo_gp* g);
const double* cand_points(const abo_cand* c);      // device, [M][d]
const double* cand_mu(const abo_cand* c);          // device, [M]
int64_t cand_size(const abo_cand* c);

// Process teardown.  `exiting()` turns true when the process has started to exit (an atexit hook registered behind the HIP
// runtime's own, so it runs BEFORE the runtime tears down, and the library's static destructor): from then on no entry point
// touches the device — a finaliser that runs late (a Julia / Python handle destroyed from an exit handler, a static
// destructor of the host) must find abo_destroy / abo_cand_destroy / abo_mgpu_destroy / pool frees as quiet no-ops, never a
// call into a runtime that is already gone.  `arm_exit_guard()` registers the hook (idempotent; called once a device exists);
// `at_exit(f)` adds work to it (the multi-device driver parks and joins its idle worker threads there).
bool exiting();
void arm_exit_guard();
void at_exit(void (*f)());
// hipSuccess, or an error that only says "the runtime is shutting down" (hipErrorDeinitialized / hipErrorContextIsDestroyed …)
bool gone(hipError_t e);

// the calling thread's last-error slot (abo_last_error reads it)
int32_t set_error(int32_t code, const char* text);
const char* last_error_text();

}  // namespace abo
