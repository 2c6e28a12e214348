// Native data-parallel step: K consecutive view-sharded optimizer steps enqueued by ONE native call,
//   per step:  eg_train_step (forward + loss + backward -> the fused [N,12] gradient buffer)
//              ncclAllReduce(sum, fp32) of that buffer ON THE LAUNCH STREAM (RCCL over xGMI)
//              eg_adam_emit (the four Adam steps on the reduced gradient + projection / binning of the rank's next view)
// -- what dist.DataParallelStep.step does from Python with three native enqueues and one torch.distributed call per step
// (host-bound at 110-140 us per step per rank before any wire time).  The reference itself is
// single-GPU and steps after every view (train_gaussians.py:104-106,311): a P-way step is a batch of P views, the
// guarantee is "all-reduced gradient == sum of the per-view gradients at the same parameters".
//
// RCCL is NOT a link dependency: librccl.so is dlopen'ed from the path the caller gives (the one PyTorch ships and has
// loaded already: torch/lib/librccl.so -- the same library instance, so the process holds one RCCL), the communicator is
// created from a ncclUniqueId the ranks exchange over torch.distributed (edgegaussians_amd/dist.py).  The communicator is
// process state (one of the stated exceptions in include/edgegs.h).
#include <dlfcn.h>

#include <chrono>
#include <cstring>

#include "common.h"

namespace eg {

// the subset of rccl.h this file needs (ABI-stable NCCL 2.x entry points)
typedef struct { char internal[128]; } NcclUniqueId;
typedef void *NcclComm;
constexpr int kNcclFloat = 7, kNcclSum = 0;
typedef int (*GetUniqueIdFn)(NcclUniqueId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);
typedef int (*CommCountFn)(NcclComm, int *);

static void *g_rccl = nullptr;
static GetUniqueIdFn p_get_unique_id = nullptr;
static CommInitRankFn p_comm_init_rank = nullptr;
static CommDestroyFn p_comm_destroy = nullptr;
static AllReduceFn p_all_reduce = nullptr;
static GetErrorStringFn p_error_string = nullptr;
static CommCountFn p_comm_count = nullptr;
static NcclComm g_comm = nullptr;
static int g_world = 0, g_rank = 0;
// eg_dp_force_all_reduce: issue the [12 N] collective of eg_train_steps_dp even through a one-rank communicator (tests)
static int g_force_all_reduce = 0;
// [12 N] ncclAllReduce calls issued by eg_train_steps_dp since eg_dp_init, and the floats they carried
static long long g_grad_all_reduces = 0, g_grad_all_reduce_floats = 0;
// eg_dp_comm_timing: HIP events on the launch stream around each [12 N] collective of the next n steps
static hipEvent_t *g_comm_ev = nullptr;  // [2 * g_comm_ev_cap]
static int g_comm_ev_cap = 0, g_comm_ev_n = 0;
// host seconds spent enqueueing {eg_train_step, ncclAllReduce, Adam + next projection} since the last eg_dp_host_profile
static double g_host_s[3] = {0.0, 0.0, 0.0};
static long long g_host_steps = 0;
static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int load_rccl(const char *path) {
  if (g_rccl) return EG_OK;
  g_rccl = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl) {
    set_error("eg_dp: dlopen(%s) failed: %s", path ? path : "librccl.so", dlerror());
    return EG_ERR_ARG;
  }
  p_get_unique_id = (GetUniqueIdFn)dlsym(g_rccl, "ncclGetUniqueId");
  p_comm_init_rank = (CommInitRankFn)dlsym(g_rccl, "ncclCommInitRank");
  p_comm_destroy = (CommDestroyFn)dlsym(g_rccl, "ncclCommDestroy");
  p_all_reduce = (AllReduceFn)dlsym(g_rccl, "ncclAllReduce");
  p_error_string = (GetErrorStringFn)dlsym(g_rccl, "ncclGetErrorString");
  p_comm_count = (CommCountFn)dlsym(g_rccl, "ncclCommCount");
  if (!p_get_unique_id || !p_comm_init_rank || !p_comm_destroy || !p_all_reduce) {
    set_error("eg_dp: %s does not export the NCCL entry points", path ? path : "librccl.so");
    dlclose(g_rccl);
    g_rccl = nullptr;
    return EG_ERR_ARG;
  }
  return EG_OK;
}

static int nccl_check(int rc, const char *what) {
  if (rc == 0) return EG_OK;
  set_error("eg_dp: %s failed: %s", what, p_error_string ? p_error_string(rc) : "nccl error");
  return EG_ERR_LAUNCH;
}

}  // namespace eg

using namespace eg;

extern "C" int eg_dp_unique_id(const char *librccl_path, void *id_out_host) {
  EG_REQUIRE(id_out_host != nullptr, "null pointer");
  const int rc = load_rccl(librccl_path);
  if (rc) return rc;
  NcclUniqueId id;
  const int e = nccl_check(p_get_unique_id(&id), "ncclGetUniqueId");
  if (e) return e;
  memcpy(id_out_host, &id, sizeof(id));
  return EG_OK;
}

extern "C" int eg_dp_init(const char *librccl_path, const void *id_host, int32_t rank, int32_t world) {
  EG_REQUIRE(id_host != nullptr && world >= 1 && rank >= 0 && rank < world, "bad arguments");
  EG_REQUIRE(g_comm == nullptr, "a communicator exists already (eg_dp_shutdown first)");
  const int rc = load_rccl(librccl_path);
  if (rc) return rc;
  NcclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  const int e = nccl_check(p_comm_init_rank(&g_comm, world, id, rank), "ncclCommInitRank");
  if (e) { g_comm = nullptr; return e; }
  g_world = world;
  g_rank = rank;
  g_grad_all_reduces = g_grad_all_reduce_floats = 0;
  return EG_OK;
}

extern "C" int eg_dp_world(void) { return g_comm ? g_world : 0; }

// what RCCL itself says the communicator's size is (ncclCommCount), 0 without a communicator, < 0 on an error
extern "C" int eg_dp_comm_count(void) {
  if (!g_comm) return 0;
  EG_REQUIRE(p_comm_count != nullptr, "librccl does not export ncclCommCount");
  int n = 0;
  const int e = nccl_check(p_comm_count(g_comm, &n), "ncclCommCount");
  return e ? e : n;
}

extern "C" int eg_dp_force_all_reduce(int32_t on) {
  g_force_all_reduce = on > 0 ? 1 : (on < 0 ? -1 : 0);
  return EG_OK;
}

extern "C" int64_t eg_dp_grad_all_reduces(int64_t *floats_out_host) {
  if (floats_out_host) *floats_out_host = g_grad_all_reduce_floats;
  return g_grad_all_reduces;
}

static void comm_timing_free() {
  for (int i = 0; i < 2 * g_comm_ev_cap; ++i) (void)hipEventDestroy(g_comm_ev[i]);
  delete[] g_comm_ev;
  g_comm_ev = nullptr;
  g_comm_ev_cap = g_comm_ev_n = 0;
}

extern "C" int eg_dp_comm_timing_begin(int32_t n_steps) {
  EG_REQUIRE(n_steps > 0 && n_steps <= (1 << 16), "bad step count");
  if (g_comm_ev) comm_timing_free();
  g_comm_ev = new hipEvent_t[2 * (size_t)n_steps];
  for (int i = 0; i < 2 * n_steps; ++i)
    if (hipEventCreate(&g_comm_ev[i]) != hipSuccess) {
      g_comm_ev_cap = i / 2;
      comm_timing_free();
      set_error("eg_dp_comm_timing_begin: hipEventCreate failed");
      return EG_ERR_LAUNCH;
    }
  g_comm_ev_cap = n_steps;
  g_comm_ev_n = 0;
  return EG_OK;
}

extern "C" int eg_dp_comm_timing_end(float *mean_us_host, float *max_us_host, int32_t *n_out_host) {
  EG_REQUIRE(g_comm_ev != nullptr, "no timing window open");
  const int n = g_comm_ev_n;
  double sum = 0.0;
  float mx = 0.f;
  if (n > 0) {
    if (hipEventSynchronize(g_comm_ev[2 * (n - 1) + 1]) != hipSuccess) {
      set_error("eg_dp_comm_timing_end: hipEventSynchronize failed");
      comm_timing_free();
      return EG_ERR_LAUNCH;
    }
    for (int i = 0; i < n; ++i) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, g_comm_ev[2 * i], g_comm_ev[2 * i + 1]);
      sum += ms;
      mx = ms > mx ? ms : mx;
    }
  }
  if (mean_us_host) *mean_us_host = n ? (float)(1e3 * sum / n) : 0.f;
  if (max_us_host) *max_us_host = 1e3f * mx;
  if (n_out_host) *n_out_host = n;
  comm_timing_free();
  return EG_OK;
}

extern "C" int eg_dp_shutdown(void) {
  if (g_comm_ev) comm_timing_free();
  g_force_all_reduce = 0;
  if (g_comm) {
    (void)p_comm_destroy(g_comm);
    g_comm = nullptr;
  }
  g_world = g_rank = 0;
  return EG_OK;
}

// sum over the ranks of a device buffer of n floats, in place, on `stream` (the regulariser's running loss sum, the
// read-back's words: small collectives that ride the same communicator)
extern "C" int eg_dp_all_reduce(float *buf, int64_t n, eg_stream_t stream) {
  EG_REQUIRE(g_comm != nullptr, "no communicator (eg_dp_init)");
  EG_REQUIRE(buf != nullptr && n > 0, "bad arguments");
  return nccl_check(p_all_reduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, g_comm, as_stream(stream)), "ncclAllReduce");
}

// `a`: step 0 as for eg_train_step in its gradient form -- adam_host == NULL, v_means / v_quats / v_scales / v_opacities
// the blocks of ONE contiguous [12 N] buffer starting at v_means (means 3 | quats 4 | log-scales 3 | logit-opacity 1 |
// absgrad increment 1: a->absgrads points at block 11) -- its viewmat / K / gt / wmap are ignored.  Step k takes view
// views_host[k] of the [V, ...] arrays and weight map wmaps_host[k]; `hyper` is step 0's Adam state (counts advance by k
// as in eg_train_steps); absgrads: the accumulator the reduced increment is added to.  have_projection as eg_train_step
// (step 0 only; inside the run every step's Adam launch projects the next view).  next_view_after: the view this rank
// rasterises in the step FOLLOWING the run (projected by the last Adam launch), or < 0.
extern "C" int eg_train_steps_dp(const eg_step_args *a, const eg_adam_hyper *hyper, float *absgrads, int32_t K,
                                 const int32_t *views_host, const float *const *wmaps_host, const float *viewmats,
                                 const float *Ks, const float *gts, int32_t next_view_after, eg_stream_t stream) {
  EG_REQUIRE(a != nullptr && hyper != nullptr && K >= 0 && (K == 0 || (views_host && wmaps_host)) && viewmats && Ks && gts,
             "bad arguments");
  EG_REQUIRE(g_comm != nullptr, "no communicator (eg_dp_init)");
  EG_REQUIRE(a->adam_host == nullptr && a->v_means && a->v_quats == a->v_means + 3 * (size_t)a->N &&
                 a->v_scales == a->v_means + 7 * (size_t)a->N && a->v_opacities == a->v_means + 10 * (size_t)a->N &&
                 a->absgrads == a->v_means + 11 * (size_t)a->N,
             "the gradient blocks must form one contiguous [12 N] buffer (means | quats | scales | opacities | absgrad)");
  EG_REQUIRE(a->seg_cap > 0 && a->ticket, "the data-parallel run uses the segmented layout");
  EG_REQUIRE(a->ws_tag <= 0 || (int64_t)a->ws_tag + K - 1 <= EG_MAX_WS_TAG, "ws_tag + K - 1 exceeds EG_MAX_WS_TAG");
  const size_t hw = (size_t)a->width * a->height;
  const int T = cdiv(a->width, kTile) * cdiv(a->height, kTile);
  const uint32_t flags = EG_FLAG_LOG_SCALES | EG_FLAG_LOGIT_OPACITIES | EG_FLAG_ANTIALIASED | EG_FLAG_TIGHT_TILES |
                         (T > kPrefixHereMaxTiles ? EG_FLAG_FRONT_PREFIX : 0u);
  float *g = a->v_means;
  for (int k = 0; k < K; ++k) {
    EG_REQUIRE(views_host[k] >= 0 && wmaps_host[k], "bad view / null weight map");
    eg_step_args s = *a;
    s.viewmat = viewmats + 16 * (size_t)views_host[k];
    s.K = Ks + 9 * (size_t)views_host[k];
    s.gt = gts + hw * (size_t)views_host[k];
    s.wmap = wmaps_host[k];
    if (a->ws_tag > 0) s.ws_tag = a->ws_tag + k;
    s.have_projection = k > 0 ? 1 : a->have_projection;
    s.next_viewmat = s.next_K = nullptr;
    const double t0 = now_s();
    int rc = eg_train_step(&s, stream);
    if (rc) return rc;
    const double t1 = now_s();
    // (a sum over ONE rank is the identity: RCCL implements a one-rank allReduce through the copy engine -- fill + copy
    // blits with ~90 us of stream stalls, profiles/r04_timeline_gaps_config2_dp_native.txt -- which a one-GPU run of this
    // leg would measure instead of the leg; eg_dp_all_reduce still goes through RCCL whatever the size)
    if ((g_world > 1 && g_force_all_reduce >= 0) || g_force_all_reduce > 0) {
      const bool timed = g_comm_ev && g_comm_ev_n < g_comm_ev_cap;
      if (timed) (void)hipEventRecord(g_comm_ev[2 * g_comm_ev_n], as_stream(stream));
      rc = nccl_check(p_all_reduce(g, g, 12 * (size_t)a->N, kNcclFloat, kNcclSum, g_comm, as_stream(stream)), "ncclAllReduce");
      if (rc) return rc;
      if (timed) (void)hipEventRecord(g_comm_ev[2 * g_comm_ev_n++ + 1], as_stream(stream));
      ++g_grad_all_reduces;
      g_grad_all_reduce_floats += 12 * (long long)a->N;
    }
    const double t2 = now_s();
    eg_adam_hyper h = *hyper;
    h.step += k;
    for (int i = 0; i < 4; ++i)
      if (h.group_steps[i] > 0) h.group_steps[i] += k;
    const int nv = k + 1 < K ? views_host[k + 1] : next_view_after;
    if (nv >= 0)
      rc = eg_adam_emit(a->means, a->log_scales, a->quats, a->logit_opacities, g, g + 7 * (size_t)a->N, g + 3 * (size_t)a->N,
                        g + 10 * (size_t)a->N, a->adam_m, a->adam_v, a->N, h, g + 11 * (size_t)a->N, absgrads,
                        viewmats + 16 * (size_t)nv, Ks + 9 * (size_t)nv, a->width, a->height, flags, a->splat, a->tile_counts,
                        a->seg_cap, a->keys, a->item_offsets, (int32_t)a->max_items, a->total,
                        T <= kPrefixHereMaxTiles ? nullptr : a->ticket, stream);
    else
      rc = eg_adam_multi(a->means, a->log_scales, a->quats, a->logit_opacities, g, g + 7 * (size_t)a->N, g + 3 * (size_t)a->N,
                         g + 10 * (size_t)a->N, a->adam_m, a->adam_v, a->N, h, g + 11 * (size_t)a->N, absgrads, stream);
    if (rc) return rc;
    g_host_s[0] += t1 - t0; g_host_s[1] += t2 - t1; g_host_s[2] += now_s() - t2;
    ++g_host_steps;
  }
  return EG_OK;
}

// measurement aid: mean HOST microseconds per step spent enqueueing {the gradient step's kernels, the all-reduce, Adam +
// the next projection} by eg_train_steps_dp since the last call of this function; returns the number of steps
extern "C" int64_t eg_dp_host_profile(double *us_out_host /*[3]*/) {
  const long long n = g_host_steps;
  for (int i = 0; i < 3; ++i) {
    if (us_out_host) us_out_host[i] = n ? 1e6 * g_host_s[i] / (double)n : 0.0;
    g_host_s[i] = 0.0;
  }
  g_host_steps = 0;
  return n;
}
