"""ctypes loader for the C-ABI library (include/multinn_hip.h).

The product path has NO CPU fallback: if ``libmultinn_hip.so`` is missing or a call fails,
a ``RuntimeError`` is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MULTINN_HIP_LIB", os.path.join(HERE, "libmultinn_hip.so"))   # override: A/B builds of the same ABI

ABI_VERSION = 121          # == MNN_ABI_VERSION of include/multinn_hip.h; load() refuses a library built for another one
F32, BF16, U8, F16 = 0, 1, 2, 3
GEMM_ACCUMULATE, GEMM_ATOMIC, GEMM_A_KBLOCK32 = 1, 2, 8

_p, _i, _l, _f, _u64, _u32, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t

# name -> (restype, argtypes); mirrors include/multinn_hip.h one to one
SIGNATURES = {
    "mnn_version": (_i, []),
    "mnn_last_error": (C.c_char_p, []),
    "mnn_gemm_tn": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _i]),
    "mnn_gemm_tn_rows": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _i, _p, _p]),
    "mnn_transpose": (_i, [_p, _p, _i, _i, _i, _i, _p, _i, _i]),
    "mnn_convert2d": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i]),
    "mnn_pianoroll_shift_timemajor": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _l]),
    "mnn_pianoroll_shift_timemajor_t": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _l, _i, _p, _p, _p]),
    "mnn_ragged_index": (_i, [_p, _p, _i, _i, _p, _f, _p, _p, _p]),
    "mnn_rows_gather16": (_i, [_p, _p, _i, _p, _p, _i, _i, _p, _i, _p, _i]),
    "mnn_rows_scatter_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "mnn_pianoroll_split_tracks": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mnn_lstm_pack_weights": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "mnn_lstm_unpack_grads": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "mnn_lstm_unpack_grads_consume": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "mnn_lstm_unpack_grads_cat": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "mnn_lstm_seq_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i]),
    "mnn_lstm_fused_outputs": (_i, [_i, _i]),
    "mnn_lstm_seq_bwd_workspace_bytes": (_sz, [_i, _i]),
    "mnn_lstm_seq_bwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "mnn_dropout_fwd": (_i, [_p, _i, _p, _p, _i, _i, _i, _f, _u64, _p, _u32, _i, _i]),
    "mnn_dropout_bwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _u64, _p, _u32, _i, _i, _i]),
    "mnn_nade_logprob_fwd": (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "mnn_nade_logprob_bwd": (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mnn_nade_sample": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _f, _u64, _u32, _u32, _p, _l, _i, _i, _p]),
    "mnn_rbm_workspace_bytes": (_sz, [_i, _i]),
    "mnn_rbm_gibbs": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _u64, _u32, _p, _u32, _p, _p, _p]),
    "mnn_rbm_gibbs_stepped": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _u64, _u32, _p, _u32, _p, _p, _p, _p]),
    "mnn_rbm_hidden": (_i, [_p, _i, _i, _i, _p, _i, _p, _p, _i, _i, _u64, _u32, _u32, _p, _p]),
    "mnn_rbm_visible": (_i, [_p, _i, _i, _i, _p, _i, _p, _p, _i, _i, _u64, _u32, _u32, _p, _p, _p]),
    "mnn_rbm_free_energy": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p]),
    "mnn_sumsq": (_i, [_p, _p, _l, _p]),
    "mnn_weighted_sum": (_i, [_p, _p, _p, _l, _p]),
    "mnn_clip_adam_step": (_i, [_p, _p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _f, _i, _p, _i, _p]),
    "mnn_step_increment": (_i, [_p, _p, _p, _f, _p, _p, _i]),
    "mnn_bias_grad": (_i, [_p, _p, _i, _i, _i, _p, _i]),
    "mnn_grad_rows_fanout": (_i, [_p, _p, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i]),
    "mnn_fill_f32": (_i, [_p, _p, _l, _f]),
    "mnn_lstm_step_det": (_i, [_p, _i, _i, _p]),
    "mnn_dense_det": (_i, [_p, _i, _i, _p]),
    "mnn_nade_sample_multi": (_i, [_p, _i, _p, _i, _i, _i, _f, _u32, _u32, _l, _i]),
    "mnn_generate_scan_workspace_bytes": (_sz, [_i, _i, _i, _p, _i]),
    "mnn_det_lstm_pack_bytes": (_sz, [_i, _i]),
    "mnn_det_lstm_pack": (_i, [_p, _p, _i, _i, _p]),
    "mnn_det_dense_pack_bytes": (_sz, [_i, _i]),
    "mnn_det_dense_pack": (_i, [_p, _p, _i, _i, _i, _p]),
    "mnn_generate_scan": (_i, [_p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _u64, _u32, _p, _p, _sz]),
    "mnn_comm_unique_id": (_i, [_p]),
    "mnn_comm_init": (_i, [C.POINTER(_p), _i, _i, _p]),
    "mnn_allreduce_flat": (_i, [_p, _p, _p, _l]),
    "mnn_comm_destroy": (_i, [_p]),
}

class DetLstmJob(C.Structure):
    """mnn_det_lstm_job (include/multinn_hip.h)."""
    _fields_ = [("x", _p), ("x_dtype", _i), ("n_x", _i), ("ld_x", _i), ("es_x", _i), ("x2", _p), ("n_x2", _i), ("ld_x2", _i), ("h_prev", _p), ("c_prev", _p),
                ("W", _p), ("bias", _p), ("c_out", _p), ("h_out", _p), ("units", _i), ("Wp", _p)]


class NadeSampleJob(C.Structure):
    """mnn_nade_sample_job (include/multinn_hip.h)."""
    _fields_ = [("bias", _p), ("ld_bias", _i), ("w_enc", _p), ("w_dec", _p), ("seed", _u64), ("samples", _p), ("nll", _p)]


class ScanLstmLayer(C.Structure):
    """mnn_scan_lstm_layer (include/multinn_hip.h)."""
    _fields_ = [("W", _p), ("bias", _p), ("units", _i)]


class DetDenseJob(C.Structure):
    """mnn_det_dense_job (include/multinn_hip.h)."""
    _fields_ = [("x", _p), ("ld_x", _i), ("K", _i), ("W", _p), ("ld_w", _i), ("N", _i), ("bias", _p), ("out", _p), ("ld_out", _i), ("Wp", _p)]


class LstmFwdLayer(C.Structure):
    _fields_ = [("units", _i), ("xproj", _p), ("wh_t", _p), ("h0", _p), ("c0", _p), ("gates", _p), ("c", _p), ("h", _p), ("hT", _p), ("ld_hT", _i), ("y", _p), ("mask", _p),
                ("wx_t", _p), ("ld_w", _i), ("bias_p", _p), ("yT", _p), ("ld_yT", _i), ("xproj_bf16", _i), ("f16", _i)]


class LstmBwdLayer(C.Structure):
    _fields_ = [("units", _i), ("dh_ext", _p), ("wh_p", _p), ("gates", _p), ("c", _p), ("c0", _p), ("dz", _p), ("dz_T", _p), ("workspace", _p),
                ("dzT_t", _p), ("ld_t", _i), ("db_p", _p), ("mask", _p), ("wx_p", _p), ("f16", _i)]


SIGNATURES["mnn_lstm2_seq_fwd"] = (_i, [_p, _i, _i, C.POINTER(LstmFwdLayer), C.POINTER(LstmFwdLayer), _f, _i, _i])
SIGNATURES["mnn_lstm2_seq_bwd"] = (_i, [_p, _i, _i, C.POINTER(LstmBwdLayer), C.POINTER(LstmBwdLayer), _f, _i, _i])
SIGNATURES["mnn_lstm_rows_gate_minor"] = (_i, [_p, _i, _i, _i, _p, _p, _p, _p])
SIGNATURES["mnn_lstm2_persist_ok"] = (_i, [_i, _i, _i])
SIGNATURES["mnn_lstm2_persist_workspace_bytes"] = (_sz, [_i, _i, _i, _i])
SIGNATURES["mnn_lstm2_persist_status"] = (_i, [_p, _i, _i, _i, C.POINTER(C.c_int)])
SIGNATURES["mnn_lstm2_persist_fwd"] = (_i, [_p, _i, _i, C.POINTER(LstmFwdLayer), C.POINTER(LstmFwdLayer), _f, _p])
SIGNATURES["mnn_lstm2_persist_bwd"] = (_i, [_p, _i, _i, C.POINTER(LstmBwdLayer), C.POINTER(LstmBwdLayer), _f, _p])
SIGNATURES["mnn_lstm_rowpar_ok"] = (_i, [_i, _i])
SIGNATURES["mnn_lstm_rowpar_workspace_bytes"] = (_sz, [_i, _i, _i])
SIGNATURES["mnn_lstm_rowpar_status"] = (_i, [_p, C.POINTER(C.c_int)])
SIGNATURES["mnn_lstm_rowpar_fwd"] = (_i, [_p, _i, _i, C.POINTER(LstmFwdLayer), _f, _p])
SIGNATURES["mnn_lstm_rowpar_bwd"] = (_i, [_p, _i, _i, C.POINTER(LstmBwdLayer), _f, _p])
SIGNATURES["mnn_lstm_resident_ok"] = (_i, [_i, _i])
SIGNATURES["mnn_lstm_resident_fwd"] = (_i, [_p, _i, _i, C.POINTER(LstmFwdLayer), _f])
SIGNATURES["mnn_lstm_resident_bwd"] = (_i, [_p, _i, _i, C.POINTER(LstmBwdLayer), _f])
SIGNATURES["mnn_lstm_cluster_ok"] = (_i, [_i, _i])
SIGNATURES["mnn_lstm_cluster_bwd_ok"] = (_i, [_i, _i])
SIGNATURES["mnn_lstm_cluster_bwd_multi_ok"] = (_i, [_i, _i, _i])
SIGNATURES["mnn_lstm_resident_fwd_multi"] = (_i, [_p, _i, _i, _i, C.POINTER(LstmFwdLayer), _f])
SIGNATURES["mnn_lstm_resident_bwd_multi"] = (_i, [_p, _i, _i, _i, C.POINTER(LstmBwdLayer), _f])
SIGNATURES["mnn_lstm_cluster_fwd_multi"] = (_i, [_p, _i, _i, _i, C.POINTER(LstmFwdLayer), _f, C.POINTER(C.c_void_p)])
SIGNATURES["mnn_lstm_cluster_bwd_multi"] = (_i, [_p, _i, _i, _i, C.POINTER(LstmBwdLayer), _f, C.POINTER(C.c_void_p)])
SIGNATURES["mnn_lstm_cluster_fwd"] = (_i, [_p, _i, _i, C.POINTER(LstmFwdLayer), _f, _p])
SIGNATURES["mnn_lstm_cluster_bwd"] = (_i, [_p, _i, _i, C.POINTER(LstmBwdLayer), _f, _p])
SIGNATURES["mnn_nade_mfma_ok"] = (_i, [_i])
SIGNATURES["mnn_nade_logprob_fwd_mfma"] = (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p])
SIGNATURES["mnn_density_gate"] = (_i, [_p, _p, _l, _l, _p, _p])
SIGNATURES["mnn_nade_logprob_fwd_gated"] = (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p])
SIGNATURES["mnn_nade_logprob_fwd_mfma_gated"] = (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p])
SIGNATURES["mnn_nade_logprob_fwd_mfma_f32"] = (_i, [_p, _i, _i, _i, _i, _p, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p])
SIGNATURES["mnn_nade_f32_pack"] = (_i, [_p, _p, _l, _i, _p])
SIGNATURES["mnn_musical_bar_stats"] = (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p])
SIGNATURES["mnn_musical_note_stats"] = (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p])
SIGNATURES["mnn_eval_counts"] = (_i, [_p, _p, _p, _l, _p])
SIGNATURES["mnn_log_loss_rows"] = (_i, [_p, _p, _p, _i, _i, _i, _p])
SIGNATURES["mnn_rbm_cd_bias_delta"] = (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _f, _p, _p])
SIGNATURES["mnn_rbm_visible_bias_init"] = (_i, [_p, _i, _p, _f, _p])
SIGNATURES["mnn_sigmoid_grad_f32"] = (_i, [_p, _l, _p, _p, _p])
SIGNATURES["mnn_rbm_cd_rows"] = (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p])
SIGNATURES["mnn_probe_sigmoid"] = (_i, [_p, _i, _i, _p])
SIGNATURES["mnn_axpby_f32"] = (_i, [_p, _l, _f, _p, _f, _p, _p])
SIGNATURES["mnn_dropout_mask"] = (_i, [_p, _p, _i, _i, _i, _f, _u64, _p, _u32, _i])

_lib = None


class MnnError(RuntimeError):
    pass


def load():
    """Load libmultinn_hip.so and bind every symbol of include/multinn_hip.h.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MnnError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(multinn_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.mnn_version() != ABI_VERSION:
        raise MnnError(f"{LIB_PATH} has ABI version {lib.mnn_version()}, this loader binds {ABI_VERSION} (include/multinn_hip.h): rebuild it")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise MnnError(f"{what} failed ({rc}): {load().mnn_last_error().decode()}")


TIMING = None      # bench.py sets this to a dict: entry point -> list of (start_event, end_event)


def call(name, *args):
    """Call an int-returning entry point and raise MnnError on a non-zero code."""
    lib = load()
    if TIMING is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()            # the library launches on torch's current stream, so these events bracket its kernels
        rc = getattr(lib, name)(*args)
        e1.record()
        TIMING.setdefault(name, []).append((e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MnnError(f"{name} failed ({rc}): {lib.mnn_last_error().decode()}")
