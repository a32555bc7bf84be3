"""ctypes binding of libdfl_hip.so (C ABI declared in include/dfl_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DFL_LIB_OVERRIDE') or os.path.join(_HERE, 'lib', 'libdfl_hip.so')   # override: docs/experiments variant builds

i32, i64, f32 = C.c_int32, C.c_int64, C.c_float
fp = C.c_void_p   # every device pointer is passed as a plain address


class ConvArgs(C.Structure):
    _fields_ = [('x', fp), ('w', fp), ('bias', fp), ('in_scale', fp), ('in_shift', fp), ('add', fp),
                ('add_scale', fp), ('add_shift', fp), ('stat_other', fp), ('y', fp), ('stat_partials', fp),
                ('partial', fp),
                ('N', i32), ('Hin', i32), ('Win', i32), ('Cin', i32), ('ldx', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('Hout', i32), ('Wout', i32), ('Ntot', i32), ('ldy', i32),
                ('ldadd', i32), ('ldso', i32), ('relu', i32), ('accumulate', i32), ('scatter2x2', i32),
                ('splits', i32), ('w_split', i32), ('x_split', i32), ('x_bf16', i32), ('y_bf16', i32),
                ('x2', fp), ('ldx2', i32), ('x_mode', i32),
                ('stat_totals', fp), ('in_tot', fp), ('in_gamma', fp), ('in_beta', fp), ('add_tot', fp), ('add_gamma', fp),
                ('add_beta', fp), ('in_count', C.c_double), ('add_count', C.c_double), ('bn_eps', f32), ('reserved3', i32),
                ('in_mean', fp), ('in_invstd', fp), ('x_out', fp), ('ldxo', i32), ('latency_form', i32), ('out_scale', fp), ('out_shift', fp)]


class WgradArgs(C.Structure):
    _fields_ = [('g', fp), ('d', fp), ('in_scale', fp), ('in_shift', fp), ('dw', fp), ('partial', fp),
                ('N', i32), ('Hin', i32), ('Win', i32), ('Cg', i32), ('ldg', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('Hout', i32), ('Wout', i32), ('Cm', i32), ('ldd', i32), ('splits', i32), ('d_split', i32),
                ('g_bf16', i32), ('d_bf16', i32), ('d_mode', i32), ('d2', fp), ('coef', fp), ('bias_partial', fp),
                ('ldd2', i32), ('reserved2', i32), ('coef_tot', fp), ('bn_gamma', fp), ('bn_mean', fp), ('bn_invstd', fp),
                ('bn_count', C.c_double)]


class PackJob(C.Structure):
    _fields_ = [('src', fp), ('dst', fp), ('A', i32), ('B', i32), ('C', i32), ('kind', i32), ('flip', i32),
                ('split', i32), ('dst2', fp), ('kind2', i32), ('flip2', i32), ('first_tile', i32), ('split2', i32)]


PACK_PLAIN, SGD_PLAIN_TILE = 100, 4096       # include/dfl_hip.h: DFL_PACK_PLAIN, DFL_SGD_PLAIN_TILE


class SgdPackArgs(C.Structure):
    _fields_ = [('jobs_dev', fp), ('grad_delta', i64), ('buf_delta', i64), ('njobs', i32), ('total_tiles', i32), ('lr', f32),
                ('momentum', f32), ('weight_decay', f32), ('grad_scale', f32), ('nesterov', i32), ('reserved', i32)]


class BnLiveJob(C.Structure):
    _fields_ = [('totals', fp), ('gamma', fp), ('beta', fp), ('running_mean', fp), ('running_var', fp), ('num_batches_tracked', fp),
                ('scale', fp), ('shift', fp), ('save_mean', fp), ('save_invstd', fp), ('count', i64), ('C', i32), ('eps', f32),
                ('momentum', f32), ('reserved', i32)]


class BnBwdLiveJob(C.Structure):
    _fields_ = [('totals', fp), ('save_mean', fp), ('save_invstd', fp), ('dgamma', fp), ('dbeta', fp), ('sum_out', fp), ('C', i32),
                ('reserved', i32)]


class BnBwdLiveArgs(C.Structure):
    _fields_ = [('jobs_dev', fp), ('njobs', i32), ('max_C', i32)]


class BnLiveArgs(C.Structure):
    _fields_ = [('jobs_dev', fp), ('njobs', i32), ('max_C', i32)]


class BnFinalizeArgs(C.Structure):
    _fields_ = [('partials', fp), ('gamma', fp), ('beta', fp), ('running_mean', fp), ('running_var', fp),
                ('num_batches_tracked', fp), ('scale', fp), ('shift', fp), ('save_mean', fp), ('save_invstd', fp),
                ('count', i64), ('nblocks', i32), ('C', i32), ('eps', f32), ('momentum', f32)]


class ColstatsArgs(C.Structure):
    _fields_ = [('a', fp), ('b', fp), ('partials', fp), ('M', i64), ('C', i32), ('lda', i32), ('ldb', i32),
                ('nblocks', i32), ('bf16', i32), ('reserved', i32)]


class BnBwdFinalizeArgs(C.Structure):
    _fields_ = [('partials', fp), ('gamma', fp), ('save_mean', fp), ('save_invstd', fp), ('dgamma', fp),
                ('dbeta', fp), ('coef', fp), ('count', i64), ('nblocks', i32), ('C', i32)]


class BnReluBwdArgs(C.Structure):
    _fields_ = [('dy', fp), ('r', fp), ('coef', fp), ('dpre', fp), ('partials', fp), ('M', i64), ('C', i32),
                ('lddy', i32), ('ldr', i32), ('ldo', i32), ('nblocks', i32), ('split_out', i32), ('bf16', i32),
                ('reserved', i32)]


class AffineCopyArgs(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('scale', fp), ('shift', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('ldx', i32), ('xH', i32), ('xW', i32), ('xoy', i32), ('xox', i32),
                ('ldy', i32), ('yH', i32), ('yW', i32), ('yoy', i32), ('yox', i32),
                ('accumulate', i32), ('bf16', i32)]


class PoolArgs(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dx', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('ldx', i32), ('ldy', i32), ('lddx', i32), ('bf16', i32)]


class HeadFwdArgs(C.Structure):
    _fields_ = [('x', fp), ('w_seg', fp), ('w_l1', fp), ('w_l2', fp), ('seg', fp), ('heat', fp),
                ('N', i32), ('H', i32), ('W', i32), ('F', i32), ('ldx', i32),
                ('NC', i32), ('NM', i32), ('L', i32), ('softmax', i32), ('x_bf16', i32)]


class HeadBwdArgs(C.Structure):
    _fields_ = [('x', fp), ('seg', fp), ('dseg', fp), ('dheat', fp), ('w_seg', fp), ('w_l1', fp), ('w_l2', fp),
                ('dx', fp), ('scratch', fp),
                ('N', i32), ('H', i32), ('W', i32), ('F', i32), ('ldx', i32), ('lddx', i32),
                ('NC', i32), ('NM', i32), ('L', i32), ('softmax', i32), ('scratch_ld', i32), ('x_bf16', i32),
                ('dw_seg', fp), ('dw_l1', fp), ('dw_l2', fp), ('wg_partial', fp),
                ('stat_other', fp), ('stat_totals', fp), ('ldso', i32), ('reserved4', i32)]


class LossArgs(C.Structure):
    _fields_ = [('seg', fp), ('tseg', fp), ('heat', fp), ('theat', fp), ('loss', fp), ('dseg', fp), ('dheat', fp),
                ('ncc_vals', fp), ('sums', fp),
                ('seg_sN', i64), ('seg_sC', i64), ('seg_sH', i64), ('tseg_sN', i64), ('tseg_sC', i64), ('tseg_sH', i64),
                ('heat_sN', i64), ('heat_sC', i64), ('heat_sH', i64), ('theat_sN', i64), ('theat_sC', i64),
                ('theat_sH', i64),
                ('B', i32), ('C', i32), ('L', i32), ('h', i32), ('w', i32), ('skip_bg', i32),
                ('dice_wgt', f32), ('heat_wgt', f32), ('grad_scale', fp),
                ('dseg_sN', i64), ('dseg_sC', i64), ('dseg_sH', i64), ('dheat_sN', i64), ('dheat_sC', i64), ('dheat_sH', i64),
                ('stage', i32), ('reserved', i32)]


class EnsembleArgs(C.Structure):
    _fields_ = [('seg_ptrs', fp), ('heat_ptrs', fp), ('labels', fp), ('avg_seg', fp), ('heat_out', fp),
                ('minmax', fp),
                ('nnets', i32), ('C', i32), ('L', i32), ('Hp', i32), ('Wp', i32), ('h', i32), ('w', i32),
                ('oy', i32), ('ox', i32), ('raw_heat', i32)]


class SumPartialsArgs(C.Structure):
    _fields_ = [('src', fp), ('dst', fp), ('n', i64), ('splits', i32), ('T', i32)]


class PackArgs(C.Structure):
    _fields_ = [('jobs_dev', fp), ('max_elems', i64), ('njobs', i32), ('tiled', i32)]


class BnEvalArgs(C.Structure):
    _fields_ = [('gamma', fp), ('beta', fp), ('running_mean', fp), ('running_var', fp), ('scale', fp),
                ('shift', fp), ('C', i32), ('eps', f32)]


class ReducePartialsArgs(C.Structure):
    _fields_ = [('partials', fp), ('out', fp), ('nblocks', i32), ('stride', i32), ('C', i32), ('reserved', i32)]


class MemsetArgs(C.Structure):
    _fields_ = [('ptr', fp), ('bytes', i64)]


class ReduceJob(C.Structure):
    _fields_ = [('src', fp), ('dst', fp), ('n', i64), ('stride', i64), ('count', i32), ('first_block', i32),
                ('T', i32), ('reserved', i32)]


class ReduceBatchArgs(C.Structure):
    _fields_ = [('jobs_dev', fp), ('njobs', i32), ('total_blocks', i32)]


class PrepArgs(C.Structure):
    _fields_ = [('proj', fp), ('labels', fp), ('lands', fp), ('x', fp), ('masks', fp), ('heats', fp), ('scratch', fp),
                ('B', i32), ('H', i32), ('W', i32), ('pad', i32), ('C', i32), ('L', i32), ('sigma', f32),
                ('standardize', i32)]


class EstLandsArgs(C.Structure):
    _fields_ = [('heats', fp), ('segs', fp), ('label_for_land', fp), ('rowcol', fp), ('ncc', fp),
                ('B', i32), ('L', i32), ('H', i32), ('W', i32), ('sigma', f32), ('min_ncc', f32)]


class UpsampleArgs(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('ldx', i32), ('ldy', i32),
                ('bf16', i32), ('accumulate', i32)]


class SyncArgs(C.Structure):
    _fields_ = [('event', i32), ('reserved', i32)]


CONV_CFG_LATENCY = 39      # dfl_conv_config: 16 + this = the latency form (csrc/convp.h: CONVS_TILE)


class ConvPairArgs(C.Structure):
    _fields_ = [('a', fp), ('b', fp)]


class Op(C.Structure):
    _fields_ = [('kind', i32), ('stream', i32), ('args', fp)]


OP_CONV, OP_WGRAD, OP_SUM_PARTIALS, OP_PACK, OP_BN_FINALIZE, OP_BN_EVAL, OP_COLSTATS, OP_BN_BWD_FINALIZE, \
    OP_BN_RELU_BWD, OP_REDUCE_PARTIALS, OP_AFFINE_COPY, OP_POOL_FWD, OP_POOL_BWD, OP_HEAD_FWD, OP_HEAD_BWD, \
    OP_MEMSET, OP_REDUCE_BATCH, OP_RECORD, OP_WAIT, OP_UPSAMPLE_FWD, OP_UPSAMPLE_BWD, OP_BN_FINALIZE_LIVE, OP_BN_BWD_FINALIZE_LIVE, \
    OP_CONV_PAIR = range(1, 25)

_KIND_OF = {ConvArgs: OP_CONV, WgradArgs: OP_WGRAD, SumPartialsArgs: OP_SUM_PARTIALS, PackArgs: OP_PACK,
            BnFinalizeArgs: OP_BN_FINALIZE, BnEvalArgs: OP_BN_EVAL, ColstatsArgs: OP_COLSTATS,
            BnBwdFinalizeArgs: OP_BN_BWD_FINALIZE, BnReluBwdArgs: OP_BN_RELU_BWD,
            ReducePartialsArgs: OP_REDUCE_PARTIALS, AffineCopyArgs: OP_AFFINE_COPY, HeadFwdArgs: OP_HEAD_FWD,
            HeadBwdArgs: OP_HEAD_BWD, MemsetArgs: OP_MEMSET, ReduceBatchArgs: OP_REDUCE_BATCH, BnLiveArgs: OP_BN_FINALIZE_LIVE, BnBwdLiveArgs: OP_BN_BWD_FINALIZE_LIVE,
            ConvPairArgs: OP_CONV_PAIR}

_SIZEOF_ORDER = [ConvArgs, WgradArgs, PackJob, BnFinalizeArgs, ColstatsArgs, BnBwdFinalizeArgs, BnReluBwdArgs,
                 AffineCopyArgs, PoolArgs, HeadFwdArgs, HeadBwdArgs, LossArgs, EnsembleArgs, Op, ReduceJob, PrepArgs, EstLandsArgs,
                 UpsampleArgs]

EXPORTS = ['dfl_version', 'dfl_last_error', 'dfl_sizeof', 'dfl_conv2d', 'dfl_conv_grid_m', 'dfl_conv2d_wgrad',
           'dfl_wgrad_suggest_splits', 'dfl_sum_partials', 'dfl_pack_weights', 'dfl_bn_finalize',
           'dfl_bn_eval_prepare', 'dfl_rowblock_count', 'dfl_colstats', 'dfl_bn_bwd_finalize',
           'dfl_bn_relu_bwd_apply', 'dfl_reduce_partials', 'dfl_affine_copy', 'dfl_maxpool2x2_fwd',
           'dfl_maxpool2x2_bwd', 'dfl_head_fwd', 'dfl_head_bwd', 'dfl_head_scratch_ld', 'dfl_head_scratch_off',
           'dfl_dice_ncc_loss', 'dfl_loss_scratch_doubles', 'dfl_ensemble_reduce', 'dfl_sgd_step', 'dfl_exec',
           'dfl_exec_timed', 'dfl_conv_config', 'dfl_wgrad_config', 'dfl_conv_suggest_splits', 'dfl_reduce_batch',
           'dfl_reduce_job_blocks', 'dfl_prep_batch', 'dfl_prep_scratch_doubles', 'dfl_est_lands', 'dfl_hard_dice', 'dfl_get_math_mode',
           'dfl_set_math_mode', 'dfl_graph_capture', 'dfl_graph_launch', 'dfl_graph_nodes', 'dfl_graph_destroy',
           'dfl_set_conv_rows_min_tiles', 'dfl_conv_candidates', 'dfl_conv_force_geometry', 'dfl_conv_tune_add',
           'dfl_head_wgrad_blocks', 'dfl_head_scratch_ld_for', 'dfl_head_scratch_off_for', 'dfl_upsample2x_fwd',
           'dfl_upsample2x_bwd', 'dfl_bn_finalize_live', 'dfl_bn_bwd_finalize_live', 'dfl_pack_weights_tiled',
           'dfl_sgd_pack_tiled', 'dfl_conv2d_pair', 'dfl_conv_pair_ok']


class DflError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DflError('libdfl_hip.so not found at %s -- build it with __graft_entry__.build() '
                       '(csrc/build.sh); there is no fallback path' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise DflError('libdfl_hip.so does not export %s' % name)
    L.dfl_last_error.restype = C.c_char_p
    L.dfl_loss_scratch_doubles.restype = i64
    L.dfl_loss_scratch_doubles.argtypes = [i32, i32, i32]
    L.dfl_rowblock_count.argtypes = [i64, i32]
    L.dfl_sum_partials.argtypes = [fp, fp, i64, i32, i32, fp]
    L.dfl_pack_weights.argtypes = [fp, i32, i64, fp]
    L.dfl_pack_weights_tiled.argtypes = [fp, i32, i32, fp]
    L.dfl_sgd_pack_tiled.argtypes = [fp, fp]
    L.dfl_bn_eval_prepare.argtypes = [fp, fp, fp, fp, fp, fp, i32, f32, fp]
    L.dfl_reduce_partials.argtypes = [fp, fp, i32, i32, i32, fp]
    L.dfl_reduce_batch.argtypes = [fp, i32, i32, fp]
    L.dfl_bn_finalize_live.argtypes = [fp, i32, i32, fp]
    L.dfl_bn_bwd_finalize_live.argtypes = [fp, i32, i32, fp]
    L.dfl_reduce_job_blocks.argtypes = [i64, i32]
    L.dfl_prep_scratch_doubles.restype = i64
    L.dfl_prep_scratch_doubles.argtypes = [i32]
    L.dfl_prep_batch.argtypes = [fp, fp]
    L.dfl_est_lands.argtypes = [fp, fp]
    L.dfl_set_math_mode.argtypes = [i32]
    L.dfl_hard_dice.argtypes = [fp, fp, i64, i32, i32, fp, fp, fp]
    L.dfl_sgd_step.argtypes = [fp, fp, fp, i64, f32, f32, f32, f32, i32, i32, fp]
    L.dfl_exec.argtypes = [fp, i32, fp]
    L.dfl_exec_timed.argtypes = [fp, i32, fp, fp]
    L.dfl_set_conv_rows_min_tiles.argtypes = [i32]
    L.dfl_graph_capture.argtypes = [fp, i32, fp, fp]
    L.dfl_graph_launch.argtypes = [fp, fp]
    L.dfl_graph_nodes.argtypes = [fp]
    L.dfl_graph_destroy.argtypes = [fp]
    L.dfl_conv_config.argtypes = [fp]
    L.dfl_wgrad_config.argtypes = [fp]
    for fn in ('dfl_conv2d', 'dfl_conv2d_wgrad', 'dfl_bn_finalize', 'dfl_colstats', 'dfl_bn_bwd_finalize',
               'dfl_bn_relu_bwd_apply', 'dfl_affine_copy', 'dfl_maxpool2x2_fwd', 'dfl_maxpool2x2_bwd',
               'dfl_head_fwd', 'dfl_head_bwd', 'dfl_dice_ncc_loss', 'dfl_ensemble_reduce', 'dfl_upsample2x_fwd',
               'dfl_upsample2x_bwd'):
        getattr(L, fn).argtypes = [fp, fp]
    for fn in ('dfl_conv_grid_m', 'dfl_wgrad_suggest_splits', 'dfl_conv_suggest_splits'):
        getattr(L, fn).argtypes = [fp]
    L.dfl_conv_candidates.argtypes = [fp, fp, i32]
    L.dfl_head_wgrad_blocks.argtypes = [i64]
    L.dfl_head_scratch_ld_for.argtypes = [i32, i32, i32, i32]
    L.dfl_head_scratch_off_for.argtypes = [i32, i32, i32, i32, i32]
    L.dfl_conv_force_geometry.argtypes = [fp]
    L.dfl_conv_tune_add.argtypes = [fp, fp]
    L.dfl_conv2d_pair.argtypes = [fp, fp, fp]
    L.dfl_conv_pair_ok.argtypes = [fp, fp]
    for k, cls in enumerate(_SIZEOF_ORDER):
        if L.dfl_sizeof(k) != C.sizeof(cls):
            raise DflError('struct mirror %s has size %d, library says %d' % (cls.__name__, C.sizeof(cls), L.dfl_sizeof(k)))
    _lib = L
    if os.environ.get('DFL_TUNE', '1') != '0':
        load_tuning(L, TUNE_PATH)
    return L


TUNE_PATH = os.environ.get('DFL_TUNE_FILE') or os.path.join(_HERE, 'tune', 'gfx950_convp.txt')


def load_tuning(L, path):
    """Geometry table of the bf16 convolution, measured on the device by tools/tune_convp.py: one layer per line,
    10 key integers (N Hin Win Cin Ntot KH KW stride pad scatter) and 5 geometry integers (include/dfl_hip.h:
    dfl_conv_tune_add).  Layers that are not listed keep the cost model's choice.  Returns the number of entries."""
    if not os.path.exists(path):
        return 0
    n = 0
    with open(path) as f:
        for line in f:
            line = line.split('#')[0].split()
            if len(line) != 15:
                continue
            v = (i32 * 15)(*[int(t) for t in line])
            if L.dfl_conv_tune_add(C.addressof(v), C.addressof(v) + 40) < 0:
                raise DflError('bad tuning entry in %s: %s' % (path, ' '.join(line)))
            n += 1
    return n


def check(rc, what=''):
    if rc < 0:
        raise DflError('%s failed (%d): %s' % (what or 'libdfl_hip call', rc, lib().dfl_last_error().decode()))
    return rc


def ptr(t):
    """Device address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def byref(s):
    return C.addressof(s)


def call(fn_name, args_struct, stream):
    return check(getattr(lib(), fn_name)(C.addressof(args_struct), stream), fn_name)


class Graph:
    """An instantiated hipGraph of (part of) a Program; keeps the program (argument structs, tensors) alive."""

    def __init__(self, handle, program):
        self.handle, self.program = handle, program

    def launch(self, stream):
        check(lib().dfl_graph_launch(self.handle, stream), 'dfl_graph_launch')

    @property
    def nodes(self):
        return lib().dfl_graph_nodes(self.handle)

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.dfl_graph_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


class Program:
    """A recorded list of library calls replayed by dfl_exec (one ctypes transition per replay)."""

    def __init__(self):
        self.structs = []      # keeps the argument structs alive
        self.kinds = []
        self.streams = []      # 0 = the caller's stream, 1.. = library side streams (dfl_op.stream)
        self.volatile = []     # ops whose argument block is rewritten between replays (caller-owned addresses): never captured
        self._ops = None
        self.keep = []         # tensors referenced by raw pointers
        self._chunks = {}      # (start, count) -> [(first op, count, Graph or None)], see run_graphed

    def add(self, args_struct, kind=None, stream=0, volatile=False):
        self.structs.append(args_struct)
        self.kinds.append(kind if kind is not None else _KIND_OF[type(args_struct)])
        self.streams.append(stream)
        self.volatile.append(bool(volatile))
        self._ops = None
        self._chunks = {}
        return args_struct

    def insert(self, index, args_struct, kind=None, stream=0, volatile=False):
        """Like add(), in front of op `index`."""
        self.structs.insert(index, args_struct)
        self.kinds.insert(index, kind if kind is not None else _KIND_OF[type(args_struct)])
        self.streams.insert(index, stream)
        self.volatile.insert(index, bool(volatile))
        self._ops = None
        self._chunks = {}
        return args_struct

    def pop(self):
        """Take the last op back (its argument struct is returned)."""
        self.kinds.pop()
        self.streams.pop()
        self.volatile.pop()
        self._ops = None
        self._chunks = {}
        return self.structs.pop()

    def record(self, event, stream=0):
        """Record library event `event` on `stream` (DFL_OP_RECORD)."""
        return self.add(SyncArgs(event=event), OP_RECORD, stream)

    def wait(self, event, stream=0):
        """Make `stream` wait for library event `event` (DFL_OP_WAIT)."""
        return self.add(SyncArgs(event=event), OP_WAIT, stream)

    def add_pool(self, args_struct, backward):
        return self.add(args_struct, OP_POOL_BWD if backward else OP_POOL_FWD)

    def extend(self, other):
        for s, k, st, v in zip(other.structs, other.kinds, other.streams, other.volatile):
            self.add(s, k, st, v)
        self.keep.extend(other.keep)

    def __len__(self):
        return len(self.structs)

    def _build(self):
        arr = (Op * len(self.structs))()
        for i, (s, k) in enumerate(zip(self.structs, self.kinds)):
            arr[i].kind = k
            arr[i].stream = self.streams[i]
            arr[i].args = C.addressof(s)
        self._ops = arr

    def run_timed(self, stream):
        """Replay with a hipEvent pair around every op (on `stream`); returns the per-op milliseconds."""
        if self._ops is None:
            self._build()
        n = len(self.structs)
        ms = (C.c_float * n)()
        check(lib().dfl_exec_timed(C.addressof(self._ops), n, stream, C.addressof(ms)), 'dfl_exec_timed')
        return list(ms)

    def capture(self, stream, start=0, count=None):
        """hipGraph of ops [start, start + count) (dfl_graph_capture): nothing runs now; Graph.launch(stream) replays."""
        if self._ops is None:
            self._build()
        n = len(self.structs) - start if count is None else count
        h = C.c_void_p()
        check(lib().dfl_graph_capture(C.addressof(self._ops) + start * C.sizeof(Op), n, stream, C.addressof(h)),
              'dfl_graph_capture')
        return Graph(h.value, self)

    MIN_GRAPH_OPS = 2

    def graph_chunks(self, stream, start=0, count=None):
        """Cut ops [start, start + count) into maximal runs that one hipGraph can stand for -- main-stream kernels whose
        argument blocks do not change between replays -- and the ops in between (event record / wait, side-stream work,
        volatile ops), which stay plain launches in program order.  Graphs are captured here, once per range."""
        n = len(self.structs) - start if count is None else count
        key = (start, n)
        chunks = self._chunks.get(key)
        if chunks is None:
            chunks, i, end = [], start, start + n
            while i < end:
                plain = self.kinds[i] in (OP_RECORD, OP_WAIT) or self.streams[i] != 0 or self.volatile[i]
                j = i + 1
                while j < end and (self.kinds[j] in (OP_RECORD, OP_WAIT) or self.streams[j] != 0 or self.volatile[j]) == plain:
                    j += 1
                if not plain and j - i >= self.MIN_GRAPH_OPS:
                    chunks.append((i, j - i, self.capture(stream, i, j - i)))
                else:
                    chunks.append((i, j - i, None))
                i = j
            self._chunks[key] = chunks
        return chunks

    def run_graphed(self, stream, start=0, count=None):
        """Replay ops [start, start + count): hipGraph launches for the capturable runs, dfl_exec for the rest."""
        if not self.structs:
            return
        for first, cnt, graph in self.graph_chunks(stream, start, count):
            if graph is not None:
                graph.launch(stream)
            else:
                self.run(stream, first, cnt)

    def run(self, stream, start=0, count=None):
        if not self.structs:
            return
        if self._ops is None:
            self._build()
        n = len(self.structs) - start if count is None else count
        base = C.addressof(self._ops) + start * C.sizeof(Op)
        check(lib().dfl_exec(base, n, stream), 'dfl_exec')
