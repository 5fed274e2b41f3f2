"""
ctypes binding of ``libcollie_hip.so`` (C ABI declared in ``include/collie_hip.h``).

There is no CPU fallback: every compute entry point of this package goes through this library and
raises ``CollieHipError`` if it cannot be loaded or a call fails.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libcollie_hip.so')

I32, I64 = 0, 1
LOSS_BPR, LOSS_HINGE, LOSS_ADAPTIVE_BPR, LOSS_ADAPTIVE_HINGE, LOSS_WARP = range(5)
OPT_SGD, OPT_ADAGRAD, OPT_SPARSE_ADAM, OPT_ADAM = range(4)
SIDE_USER, SIDE_ITEM = 0, 1
MAX_META = 16
MAX_NEG = 63

LOSS_KINDS = {'bpr': LOSS_BPR, 'hinge': LOSS_HINGE, 'adaptive_bpr': LOSS_ADAPTIVE_BPR,
              'adaptive_hinge': LOSS_ADAPTIVE_HINGE, 'warp': LOSS_WARP}
OPT_ALGOS = {'sgd': OPT_SGD, 'adagrad': OPT_ADAGRAD, 'sparse_adam': OPT_SPARSE_ADAM,
             'adam': OPT_ADAM}


class CollieHipError(RuntimeError):
    pass


class Tables(ctypes.Structure):
    _fields_ = [('user_emb', c_void_p), ('item_emb', c_void_p), ('user_bias', c_void_p),
                ('item_bias', c_void_p), ('num_users', c_int64), ('num_items', c_int64),
                ('dim', c_int32)]


class Batch(ctypes.Structure):
    _fields_ = [('users', c_void_p), ('pos_items', c_void_p), ('neg_items', c_void_p),
                ('users_dtype', c_int32), ('pos_dtype', c_int32), ('neg_dtype', c_int32),
                ('num_neg', c_int32), ('batch', c_int64), ('neg_stride_b', c_int64),
                ('neg_stride_k', c_int64)]


class LossCfg(ctypes.Structure):
    _fields_ = [('kind', c_int32), ('n_meta', c_int32), ('num_items', c_int64),
                ('meta', c_void_p * MAX_META), ('meta_weight', c_float * MAX_META)]


class OptimCfg(ctypes.Structure):
    _fields_ = [('algo', c_int32), ('step', c_int32), ('lr', c_float), ('beta1', c_float),
                ('beta2', c_float), ('eps', c_float), ('weight_decay', c_float),
                ('lr_decay', c_float)]


DENSE_MAX_SEGS = 4


class DenseSeg(ctypes.Structure):
    _fields_ = [('src', c_void_p), ('idx', c_void_p), ('act_ref', c_void_p), ('ld', c_int64),
                ('act_ld', c_int64), ('rows', c_int64), ('idx_dtype', c_int32), ('width', c_int32)]


class DenseOperand(ctypes.Structure):
    _fields_ = [('n_seg', c_int32), ('seg', DenseSeg * DENSE_MAX_SEGS)]


# name -> (restype, argtypes): exactly the symbols include/collie_hip.h declares
SIGNATURES = {
    'collie_last_error': (c_char_p, []),
    'collie_version': (ctypes.c_int, []),
    'collie_mf_score': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int32, c_void_p, c_int32,
                                       c_int64, c_int32, c_float, c_float, c_void_p, c_void_p,
                                       c_void_p]),
    'collie_mf_score_backward': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int32, c_void_p,
                                                c_int32, c_int64, c_int32, c_float, c_float,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    'collie_mf_score_dropout': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int32, c_void_p, c_int32,
                                               c_int64, c_int32, c_float, c_float, c_float,
                                               ctypes.c_uint64, ctypes.c_uint64, c_void_p, c_void_p,
                                               c_void_p]),
    'collie_mf_score_dropout_backward': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int32, c_void_p,
                                                        c_int32, c_int64, c_int32, c_float, c_float,
                                                        c_float, ctypes.c_uint64, ctypes.c_uint64,
                                                        c_void_p, c_void_p, c_void_p, c_void_p,
                                                        c_void_p, c_void_p]),
    'collie_loss_forward_backward': (ctypes.c_int, [POINTER(LossCfg), c_void_p, c_void_p, c_int64,
                                                    c_int32, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_loss_partials': (c_int64, [c_int64]),
    'collie_mf_train_workspace_bytes': (c_size_t, [c_int64, c_int32, c_int32, c_int64, c_int64]),
    'collie_mf_train_forward': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), POINTER(LossCfg),
                                               c_void_p, c_size_t, c_int32, c_int32, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_mf_train_build_segments': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                      c_size_t, c_void_p]),
    'collie_mf_train_update': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p, c_size_t,
                                              c_int32, c_int32, c_int32, c_int32,
                                              POINTER(OptimCfg), POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                              c_void_p]),
    'collie_mf_train_export_grad': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                   c_size_t, c_int32, c_int32, c_int32, c_void_p,
                                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_mf_train_update_needs_segments': (ctypes.c_int, [POINTER(Tables), POINTER(Batch),
                                                             c_int32]),
    'collie_mf_train_update_all': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                  c_size_t, c_int32, c_int32, c_int32,
                                                  POINTER(OptimCfg), POINTER(c_void_p),
                                                  POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                                  c_void_p]),
    'collie_mf_train_step': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), POINTER(LossCfg),
                                            c_void_p, c_size_t, c_int32, c_int32, c_void_p, c_void_p,
                                            POINTER(OptimCfg), POINTER(c_void_p), POINTER(c_void_p),
                                            POINTER(c_void_p), c_void_p]),
    'collie_mf_train_schedule_steps': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                      c_size_t, c_int32, c_int32, POINTER(OptimCfg),
                                                      c_void_p]),
    'collie_mf_train_advance_step': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                    c_size_t, c_void_p, c_void_p, c_void_p]),
    'collie_mf_batch_copy': (ctypes.c_int, [POINTER(Batch), POINTER(Batch), c_void_p]),
    'collie_sparse_rows_workspace_bytes': (c_size_t, [c_int64, c_int32]),
    'collie_shard_wire_rows': (c_int64, [c_void_p, c_int32, c_int32, c_int32]),
    'collie_shard_plan_workspace_bytes': (c_size_t, [c_int64, c_int32, c_int32, c_int64, c_int64,
                                                    c_int32]),
    'collie_shard_plan': (ctypes.c_int, [POINTER(Batch), c_int64, c_int64, c_int32, c_int32, c_int32,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p]),
    'collie_shard_serve_rows': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                               c_void_p, c_int32, c_void_p, c_int64, c_void_p,
                                               c_int64, c_int32, c_void_p, c_void_p]),
    'collie_shard_unpack_bias': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32,
                                                c_void_p, c_void_p]),
    'collie_shard_export_workspace_bytes': (c_size_t, [c_int64]),
    'collie_shard_export_grads': (ctypes.c_int, [POINTER(Tables), POINTER(Batch), c_void_p,
                                                 c_size_t, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                                 c_void_p]),
    'collie_shard_owner_workspace_bytes': (c_size_t, [c_int64, c_int64, c_int32]),
    'collie_shard_owner_update': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                                 c_void_p, c_int64, c_void_p, c_int32, c_void_p,
                                                 c_void_p, POINTER(OptimCfg), POINTER(OptimCfg),
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_size_t, c_void_p, c_void_p]),
    'collie_mf_eval_topk_capacity': (c_int32, []),
    'collie_mf_eval_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'collie_mf_eval_topk': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int64, c_int64, c_int64,
                                           c_int64, c_int64, c_int32, c_int32, c_float, c_float,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'collie_mf_eval_topk_static': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int64, c_int64,
                                                  c_int64, c_int64, c_int64, c_int32, c_int32,
                                                  c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'collie_mf_eval_bf16x3_ok': (c_int32, [c_int32, c_int32]),
    'collie_mf_eval_split_bf16x3': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    'collie_mf_eval_bf16x3_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'collie_mf_eval_topk_bf16x3': (ctypes.c_int, [POINTER(Tables), c_void_p, c_void_p, c_int64,
                                                  c_int64, c_int64, c_int64, c_int64, c_int32,
                                                  c_int32, c_float, c_float, c_void_p, c_void_p,
                                                  c_void_p, c_size_t, c_void_p]),
    'collie_mf_eval_merge_topk': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                                 c_void_p, c_void_p, c_void_p]),
    'collie_mf_eval_pos_scores': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int64, c_void_p,
                                                 c_void_p, c_int64, c_int32, c_float, c_float,
                                                 c_int32, c_int64, c_int64, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_mf_eval_rank_counts': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int64, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                  c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'collie_mf_eval_topk_rank_counts': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int64, c_int64,
                                                       c_int64, c_int64, c_int64, c_int32, c_int32,
                                                       c_float, c_float, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_size_t,
                                                       c_void_p]),
    'collie_preds_topk': (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p,
                                         c_void_p, c_void_p]),
    'collie_preds_pos_scores': (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                               c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_preds_rank_counts': (ctypes.c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    'collie_embedding_similarity': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p,
                                                   c_void_p]),
    'collie_mf_explicit_forward': (ctypes.c_int, [POINTER(Tables), c_void_p, c_int32, c_void_p,
                                                  c_int32, c_void_p, c_int64, c_int32, c_int32,
                                                  c_float, c_float, c_void_p, c_size_t, c_int32,
                                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_sample_batch': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                           c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                           ctypes.c_uint64, ctypes.c_uint64, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    'collie_mf_eval_metrics': (ctypes.c_int, [c_void_p, c_int32, c_int64, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int32, c_float,
                                              c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_dense_set_kernels': (ctypes.c_int, [c_int32]),
    'collie_dense_forward': (ctypes.c_int, [POINTER(DenseOperand), c_int64, c_void_p, c_int64,
                                            c_int64, c_void_p, c_int32, c_int32, c_int32, c_float,
                                            c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int32,
                                            c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                            c_void_p]),
    'collie_dense_forward_add': (ctypes.c_int, [POINTER(DenseOperand), c_int64, c_void_p, c_int64,
                                                c_int64, c_void_p, c_int32, c_int32, c_int32, c_float,
                                                c_void_p, c_int32, c_void_p, c_int64, c_void_p,
                                                c_int32, c_void_p, c_int64, c_void_p, c_int64,
                                                c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    'collie_active_rows': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_pair_ids': (ctypes.c_int, [POINTER(Batch), c_void_p, c_void_p, c_void_p]),
    'collie_gather_rows': (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_int64),
                                          POINTER(c_int32), POINTER(c_void_p), c_void_p, c_int64,
                                          c_int64, c_void_p, c_void_p]),
    'collie_dense_optim_step_multi': (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p),
                                                     POINTER(c_int64), POINTER(OptimCfg),
                                                     POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    'collie_dense_outer_sum': (ctypes.c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                              c_void_p, c_int32, c_int32, c_float, c_void_p, c_int64,
                                              c_void_p]),
    'collie_dense_grid_tail': (ctypes.c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_float, c_void_p, c_int32,
                                              c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                              c_void_p, c_int64, c_void_p, c_void_p]),
    'collie_dense_weight_grad_workspace_bytes': (c_size_t, [c_int64, c_int32, c_int32]),
    'collie_dense_weight_grad': (ctypes.c_int, [POINTER(DenseOperand), POINTER(DenseOperand),
                                                c_int64, c_int32, c_int32, c_float, c_void_p,
                                                c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'collie_dense_optim_step': (ctypes.c_int, [c_void_p, c_void_p, c_int64, POINTER(OptimCfg),
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'collie_dense_rows_update': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32,
                                                c_void_p, c_int64, POINTER(OptimCfg), c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                                c_void_p]),
}

_lib = None


def lib_path():
    return _LIB_PATH


def load():
    """Load the shared library (once) and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise CollieHipError(
            f'{_LIB_PATH} not found. Build it with ``python -c "import __graft_entry__ as g; '
            'g.build()"`` (or ``make -C collie_amd/csrc``). collie_amd has no CPU fallback.'
        )
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().collie_last_error().decode('utf-8', 'replace')
        raise CollieHipError(f'{what} failed ({rc}): {msg}')


def require_hip(tensor, what):
    if not tensor.is_cuda:
        raise CollieHipError(
            f'{what}: tensor is on {tensor.device}; collie_amd computes on a HIP device only '
            '(no CPU fallback). Move the model and data with ``.to("cuda")``.'
        )


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


def stream_ptr():
    """torch's current HIP stream of the current device as a raw handle (the private accessor is
    ~30x cheaper than building a ``torch.cuda.Stream`` object per launch)."""
    try:
        return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:                       # accessor renamed: the public route still works
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def idx_dtype(t):
    if t.dtype == torch.int64:
        return I64
    if t.dtype == torch.int32:
        return I32
    raise CollieHipError(f'index tensors must be int32 or int64, not {t.dtype}')


def make_tables(user_emb, item_emb, user_bias, item_bias):
    for name, t in (('user_embeddings', user_emb), ('item_embeddings', item_emb),
                    ('user_biases', user_bias), ('item_biases', item_bias)):
        require_hip(t, name)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise CollieHipError(f'{name} must be a contiguous float32 tensor')
    if user_emb.shape[1] != item_emb.shape[1]:
        raise CollieHipError('user and item embedding dims differ')
    return Tables(user_emb.data_ptr(), item_emb.data_ptr(), user_bias.data_ptr(),
                  item_bias.data_ptr(), user_emb.shape[0], item_emb.shape[0], user_emb.shape[1])
