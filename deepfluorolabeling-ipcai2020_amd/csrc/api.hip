// Library-level entry points: version, thread-local error text, and the program executor that replays a recorded
// list of kernel calls with one host->library transition (include/dfl_hip.h: dfl_exec).
#include <string.h>

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

extern "C" int dfl_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* dfl_last_error(void) { return dfl::g_err; }

extern "C" int dfl_sizeof(int which) {
  static const int sizes[] = {
      (int)sizeof(dfl_conv_args),       (int)sizeof(dfl_wgrad_args),           (int)sizeof(dfl_pack_job),
      (int)sizeof(dfl_bn_finalize_args), (int)sizeof(dfl_colstats_args),       (int)sizeof(dfl_bn_bwd_finalize_args),
      (int)sizeof(dfl_bn_relu_bwd_args), (int)sizeof(dfl_affine_copy_args),    (int)sizeof(dfl_pool_args),
      (int)sizeof(dfl_head_fwd_args),   (int)sizeof(dfl_head_bwd_args),        (int)sizeof(dfl_loss_args),
      (int)sizeof(dfl_ensemble_args),   (int)sizeof(dfl_op),                   (int)sizeof(dfl_reduce_job)};
  if (which < 0 || which >= (int)(sizeof(sizes) / sizeof(sizes[0]))) return -1;
  return sizes[which];
}

static int exec_one(const dfl_op* ops, int i, dfl_stream_t stream);

extern "C" int dfl_exec(const dfl_op* ops, int32_t n_ops, dfl_stream_t stream) {
  if (ops == nullptr || n_ops < 0) {
    dfl::set_error("dfl_exec: bad arguments");
    return DFL_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_ops; ++i) {
    int rc = exec_one(ops, i, stream);
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
  for (int i = 0; i <= n_ops; ++i) hipEventCreate(&ev[i]);
  int rc = DFL_OK;
  hipEventRecord(ev[0], s);
  int done = 0;
  for (int i = 0; i < n_ops; ++i) {
    rc = exec_one(ops, i, stream);
    if (rc != DFL_OK) break;
    hipEventRecord(ev[i + 1], s);
    done = i + 1;
  }
  hipStreamSynchronize(s);
  for (int i = 0; i < done; ++i) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
    ms_out[i] = ms;
  }
  for (int i = 0; i <= n_ops; ++i) hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

static int exec_one(const dfl_op* ops, int i, dfl_stream_t stream) {
  {
    const void* p = ops[i].args;
    int rc = DFL_OK;
    switch (ops[i].kind) {
      case DFL_OP_CONV: rc = dfl_conv2d(static_cast<const dfl_conv_args*>(p), stream); break;
      case DFL_OP_WGRAD: rc = dfl_conv2d_wgrad(static_cast<const dfl_wgrad_args*>(p), stream); break;
      case DFL_OP_SUM_PARTIALS: {
        const dfl_sum_partials_args* a = static_cast<const dfl_sum_partials_args*>(p);
        rc = dfl_sum_partials(a->src, a->dst, a->n, a->splits, a->T, stream);
        break;
      }
      case DFL_OP_PACK: {
        const dfl_pack_args* a = static_cast<const dfl_pack_args*>(p);
        rc = dfl_pack_weights(a->jobs_dev, a->njobs, a->max_elems, stream);
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
      case DFL_OP_HEAD_FWD: rc = dfl_head_fwd(static_cast<const dfl_head_fwd_args*>(p), stream); break;
      case DFL_OP_HEAD_BWD: rc = dfl_head_bwd(static_cast<const dfl_head_bwd_args*>(p), stream); break;
      case DFL_OP_REDUCE_BATCH: {
        const dfl_reduce_batch_args* a = static_cast<const dfl_reduce_batch_args*>(p);
        rc = dfl_reduce_batch(a->jobs_dev, a->njobs, a->total_blocks, stream);
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
