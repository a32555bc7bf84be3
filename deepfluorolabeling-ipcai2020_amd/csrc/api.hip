// Library-level entry points: version, thread-local error text, and the program executor that replays a recorded
// list of kernel calls with one host->library transition (include/dfl_hip.h: dfl_exec).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace dfl {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dfl

namespace dfl {
// Side streams and events of the program executor (dfl_op.stream / DFL_OP_RECORD / DFL_OP_WAIT): created on first use
// on the device that is current then (one process drives one GPU), never destroyed.
static std::mutex g_sync_mu;
static hipStream_t g_side[DFL_MAX_SIDE_STREAMS] = {nullptr};
static std::vector<hipEvent_t> g_events;

static hipStream_t pick_stream(int id, hipStream_t main) {
  if (id <= 0) return main;
  if (id > DFL_MAX_SIDE_STREAMS) return nullptr;
  std::lock_guard<std::mutex> lk(g_sync_mu);
  if (g_side[id - 1] == nullptr && hipStreamCreateWithFlags(&g_side[id - 1], hipStreamNonBlocking) != hipSuccess) return nullptr;
  return g_side[id - 1];
}

static hipEvent_t pick_event(int id) {
  if (id < 0 || id >= 65536) return nullptr;
  std::lock_guard<std::mutex> lk(g_sync_mu);
  while ((int)g_events.size() <= id) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_events.push_back(e);
  }
  return g_events[id];
}
}  // namespace dfl

namespace dfl {
static int g_math = -1;
int math_mode() {
  if (g_math < 0) {
    const char* e = getenv("DFL_MATH");
    g_math = (e != nullptr && strcmp(e, "fp32") == 0) ? 0 : (e != nullptr && strcmp(e, "bf16x6") == 0) ? 2
             : (e != nullptr && strcmp(e, "bf16x3") == 0) ? 1 : (e != nullptr && strcmp(e, "bf16") == 0) ? 3
             : (e != nullptr && strcmp(e, "bf16s") == 0) ? 4 : DFL_MATH_DEFAULT;
  }
  return g_math;
}
}  // namespace dfl

extern "C" int dfl_get_math_mode(void) { return dfl::math_mode(); }

extern "C" int dfl_set_math_mode(int32_t mode) {
  DFL_REQUIRE(mode >= 0 && mode <= 4, "dfl_set_math_mode: mode must be 0 (fp32), 1 (bf16x3), 2 (bf16x6), 3 (bf16) or 4 (bf16 storage)");
  dfl::g_math = mode;
  return DFL_OK;
}

extern "C" int dfl_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* dfl_last_error(void) { return dfl::g_err; }

extern "C" int dfl_sizeof(int which) {
  static const int sizes[] = {
      (int)sizeof(dfl_conv_args),       (int)sizeof(dfl_wgrad_args),           (int)sizeof(dfl_pack_job),
      (int)sizeof(dfl_bn_finalize_args), (int)sizeof(dfl_colstats_args),       (int)sizeof(dfl_bn_bwd_finalize_args),
      (int)sizeof(dfl_bn_relu_bwd_args), (int)sizeof(dfl_affine_copy_args),    (int)sizeof(dfl_pool_args),
      (int)sizeof(dfl_head_fwd_args),   (int)sizeof(dfl_head_bwd_args),        (int)sizeof(dfl_loss_args),
      (int)sizeof(dfl_ensemble_args),   (int)sizeof(dfl_op),                   (int)sizeof(dfl_reduce_job),
      (int)sizeof(dfl_prep_args),       (int)sizeof(dfl_est_lands_args),       (int)sizeof(dfl_upsample_args)};
  if (which < 0 || which >= (int)(sizeof(sizes) / sizeof(sizes[0]))) return -1;
  return sizes[which];
}

static int exec_one(const dfl_op* ops, int i, dfl_stream_t stream, bool serial);

extern "C" int dfl_exec(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream) {
  if (ops == nullptr || n_ops < 0) {
    dfl::set_error("dfl_exec: bad arguments");
    return DFL_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_ops; ++i) {
    int rc = exec_one(ops, i, stream, false);
    if (rc != DFL_OK) return rc;
  }
  return DFL_OK;
}

extern "C" int dfl_exec_timed(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream, float* ms_out) {
  if (ops == nullptr || n_ops <= 0 || ms_out == nullptr) {
    dfl::set_error("dfl_exec_timed: bad arguments");
    return DFL_ERR_INVALID_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = new hipEvent_t[n_ops + 1];
  for (int i = 0; i <= n_ops; ++i) (void)hipEventCreate(&ev[i]);
  int rc = DFL_OK;
  (void)hipEventRecord(ev[0], s);
  int done = 0;
  for (int i = 0; i < n_ops; ++i) {
    rc = exec_one(ops, i, stream, true);   // program order on one stream is a valid schedule of any program
    if (rc != DFL_OK) break;
    (void)hipEventRecord(ev[i + 1], s);
    done = i + 1;
  }
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < done; ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
    ms_out[i] = ms;
  }
  for (int i = 0; i <= n_ops; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

struct dfl_graph_s {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int nodes;
};

extern "C" int dfl_graph_capture(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream, dfl_graph_t* graph_out) {
  if (ops == nullptr || n_ops <= 0 || graph_out == nullptr) {
    dfl::set_error("dfl_graph_capture: bad arguments");
    return DFL_ERR_INVALID_ARG;
  }
  *graph_out = nullptr;
  (void)stream;   // the capture runs on a private stream: the caller's may be the null stream, which cannot capture
  hipStream_t s = nullptr;
  hipError_t err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (err != hipSuccess) {
    dfl::set_error("dfl_graph_capture: hipStreamCreateWithFlags: %s", hipGetErrorString(err));
    return DFL_ERR_LAUNCH;
  }
  err = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (err != hipSuccess) {
    (void)hipStreamDestroy(s);
    dfl::set_error("dfl_graph_capture: hipStreamBeginCapture: %s", hipGetErrorString(err));
    return DFL_ERR_LAUNCH;
  }
  int rc = DFL_OK;
  for (int i = 0; i < n_ops && rc == DFL_OK; ++i) rc = exec_one(ops, i, static_cast<dfl_stream_t>(s), true);
  hipGraph_t g = nullptr;
  err = hipStreamEndCapture(s, &g);   // always end the capture, also after a failed op
  (void)hipStreamDestroy(s);
  if (rc != DFL_OK) {
    if (g != nullptr) (void)hipGraphDestroy(g);
    return rc;
  }
  if (err != hipSuccess || g == nullptr) {
    dfl::set_error("dfl_graph_capture: hipStreamEndCapture: %s", hipGetErrorString(err));
    return DFL_ERR_LAUNCH;
  }
  hipGraphExec_t e = nullptr;
  err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    dfl::set_error("dfl_graph_capture: hipGraphInstantiate: %s", hipGetErrorString(err));
    return DFL_ERR_LAUNCH;
  }
  size_t nn = 0;
  if (hipGraphGetNodes(g, nullptr, &nn) != hipSuccess) nn = 0;
  dfl_graph_s* h = new dfl_graph_s{g, e, (int)nn};
  *graph_out = h;
  return DFL_OK;
}

extern "C" int dfl_graph_launch(dfl_graph_t graph, dfl_stream_t stream) {
  if (graph == nullptr) {
    dfl::set_error("dfl_graph_launch: null graph");
    return DFL_ERR_INVALID_ARG;
  }
  hipError_t err = hipGraphLaunch(graph->exec, static_cast<hipStream_t>(stream));
  if (err != hipSuccess) {
    dfl::set_error("dfl_graph_launch: %s", hipGetErrorString(err));
    return DFL_ERR_LAUNCH;
  }
  return DFL_OK;
}

extern "C" int dfl_graph_nodes(dfl_graph_t graph) { return graph == nullptr ? DFL_ERR_INVALID_ARG : graph->nodes; }

extern "C" int dfl_graph_destroy(dfl_graph_t graph) {
  if (graph == nullptr) return DFL_OK;
  (void)hipGraphExecDestroy(graph->exec);
  (void)hipGraphDestroy(graph->graph);
  delete graph;
  return DFL_OK;
}

static int exec_one(const dfl_op* ops, int i, dfl_stream_t main_stream, bool serial) {
  {
    const void* p = ops[i].args;
    int rc = DFL_OK;
    dfl_stream_t stream = main_stream;
    if (!serial && ops[i].stream != 0) {
      stream = dfl::pick_stream(ops[i].stream, static_cast<hipStream_t>(main_stream));
      if (stream == nullptr) {
        dfl::set_error("dfl_exec: cannot get side stream %d (op %d)", ops[i].stream, i);
        return DFL_ERR_LAUNCH;
      }
    }
    switch (ops[i].kind) {
      case DFL_OP_RECORD:
      case DFL_OP_WAIT: {
        if (serial) break;
        const dfl_sync_args* a = static_cast<const dfl_sync_args*>(p);
        hipEvent_t e = dfl::pick_event(a->event);
        hipError_t err = (e == nullptr) ? hipErrorInvalidValue
                         : (ops[i].kind == DFL_OP_RECORD ? hipEventRecord(e, static_cast<hipStream_t>(stream))
                                                         : hipStreamWaitEvent(static_cast<hipStream_t>(stream), e, 0));
        if (err != hipSuccess) {
          dfl::set_error("dfl_exec: event %d: %s", a->event, hipGetErrorString(err));
          rc = DFL_ERR_LAUNCH;
        }
        break;
      }
      case DFL_OP_CONV: rc = dfl_conv2d(static_cast<const dfl_conv_args*>(p), stream); break;
      case DFL_OP_CONV_PAIR: {
        const dfl_conv_pair_args* a = static_cast<const dfl_conv_pair_args*>(p);
        rc = dfl_conv2d_pair(a->a, a->b, stream);
        break;
      }
      case DFL_OP_WGRAD: rc = dfl_conv2d_wgrad(static_cast<const dfl_wgrad_args*>(p), stream); break;
      case DFL_OP_SUM_PARTIALS: {
        const dfl_sum_partials_args* a = static_cast<const dfl_sum_partials_args*>(p);
        rc = dfl_sum_partials(a->src, a->dst, a->n, a->splits, a->T, stream);
        break;
      }
      case DFL_OP_PACK: {
        const dfl_pack_args* a = static_cast<const dfl_pack_args*>(p);
        rc = a->tiled ? dfl_pack_weights_tiled(a->jobs_dev, a->njobs, (int32_t)a->max_elems, stream)
                      : dfl_pack_weights(a->jobs_dev, a->njobs, a->max_elems, stream);
        break;
      }
      case DFL_OP_BN_FINALIZE: rc = dfl_bn_finalize(static_cast<const dfl_bn_finalize_args*>(p), stream); break;
      case DFL_OP_BN_EVAL: {
        const dfl_bn_eval_args* a = static_cast<const dfl_bn_eval_args*>(p);
        rc = dfl_bn_eval_prepare(a->gamma, a->beta, a->running_mean, a->running_var, a->scale, a->shift, a->C, a->eps, stream);
        break;
      }
      case DFL_OP_COLSTATS: rc = dfl_colstats(static_cast<const dfl_colstats_args*>(p), stream); break;
      case DFL_OP_BN_BWD_FINALIZE: rc = dfl_bn_bwd_finalize(static_cast<const dfl_bn_bwd_finalize_args*>(p), stream); break;
      case DFL_OP_BN_RELU_BWD: rc = dfl_bn_relu_bwd_apply(static_cast<const dfl_bn_relu_bwd_args*>(p), stream); break;
      case DFL_OP_REDUCE_PARTIALS: {
        const dfl_reduce_partials_args* a = static_cast<const dfl_reduce_partials_args*>(p);
        rc = dfl_reduce_partials(a->partials, a->out, a->nblocks, a->stride, a->C, stream);
        break;
      }
      case DFL_OP_AFFINE_COPY: rc = dfl_affine_copy(static_cast<const dfl_affine_copy_args*>(p), stream); break;
      case DFL_OP_POOL_FWD: rc = dfl_maxpool2x2_fwd(static_cast<const dfl_pool_args*>(p), stream); break;
      case DFL_OP_POOL_BWD: rc = dfl_maxpool2x2_bwd(static_cast<const dfl_pool_args*>(p), stream); break;
      case DFL_OP_UPSAMPLE_FWD: rc = dfl_upsample2x_fwd(static_cast<const dfl_upsample_args*>(p), stream); break;
      case DFL_OP_UPSAMPLE_BWD: rc = dfl_upsample2x_bwd(static_cast<const dfl_upsample_args*>(p), stream); break;
      case DFL_OP_HEAD_FWD: rc = dfl_head_fwd(static_cast<const dfl_head_fwd_args*>(p), stream); break;
      case DFL_OP_HEAD_BWD: rc = dfl_head_bwd(static_cast<const dfl_head_bwd_args*>(p), stream); break;
      case DFL_OP_REDUCE_BATCH: {
        const dfl_reduce_batch_args* a = static_cast<const dfl_reduce_batch_args*>(p);
        rc = dfl_reduce_batch(a->jobs_dev, a->njobs, a->total_blocks, stream);
        break;
      }
      case DFL_OP_BN_FINALIZE_LIVE: {
        const dfl_bn_live_args* a = static_cast<const dfl_bn_live_args*>(p);
        rc = dfl_bn_finalize_live(a->jobs_dev, a->njobs, a->max_C, stream);
        break;
      }
      case DFL_OP_BN_BWD_FINALIZE_LIVE: {
        const dfl_bn_bwd_live_args* a = static_cast<const dfl_bn_bwd_live_args*>(p);
        rc = dfl_bn_bwd_finalize_live(a->jobs_dev, a->njobs, a->max_C, stream);
        break;
      }
      case DFL_OP_MEMSET: {
        const dfl_memset_args* a = static_cast<const dfl_memset_args*>(p);
        if (a->bytes > 0 && hipMemsetAsync(a->ptr, 0, (size_t)a->bytes, static_cast<hipStream_t>(stream)) != hipSuccess) {
          dfl::set_error("dfl_exec: hipMemsetAsync failed");
          rc = DFL_ERR_LAUNCH;
        }
        break;
      }
      default:
        dfl::set_error("dfl_exec: unknown op kind %d at index %d", ops[i].kind, i);
        rc = DFL_ERR_INVALID_ARG;
    }
    if (rc != DFL_OK) {
      char tmp[400];
      strncpy(tmp, dfl::g_err, sizeof(tmp) - 1);
      tmp[sizeof(tmp) - 1] = 0;
      dfl::set_error("dfl_exec: op %d (kind %d) failed: %s", i, ops[i].kind, tmp);
      return rc;
    }
  }
  return DFL_OK;
}

extern "C" int dfl_conv_config(const dfl_conv_args* a);
extern "C" int dfl_wgrad_config(const dfl_wgrad_args* a);
