"""ctypes binding of libvmmt.so (the C-ABI declared in include/vmmt.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and every
kernel wrapper raises RuntimeError on a non-zero return code.  Build with `python -m variational_mmt_amd.build`
(or `__graft_entry__.build()`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmmt.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SOFTPLUS, ACT_SIGMOID = 0, 1, 2, 3, 4
GEMM_NT, GEMM_TN, GEMM_NN = 0, 1, 2
TILE_128_ONE_PER_CU = 129
(STAT_NLL, STAT_NWORDS, STAT_NCORRECT, STAT_KL_SUM, STAT_IMG_LOGPROB, STAT_IMG_COS, STAT_GRAD_SUMSQ) = range(7)
STAT_COUNT = 8
STAT_TICKET = 7          # vmmt.h: VMMT_STAT_TICKET (a work word of vmmt_gen_fwd_combine_stats)
SUMSQ_SLOTS, SUMSQ_MAXBLOCKS = 8, 768
SEQ_GUARD_WORD = 4 + 2 * 256                                        # vmmt.h: VMMT_SEQ_GUARD_WORD
SUMSQ_SCRATCH = 2 * SUMSQ_SLOTS + SUMSQ_SLOTS * SUMSQ_MAXBLOCKS      # vmmt.h: VMMT_SUMSQ_SCRATCH
LAZY_HIST = 128                                                     # vmmt.h: VMMT_LAZY_HIST
LAZY_HIST_WORDS = 4 + 4 * LAZY_HIST                                  # vmmt.h: VMMT_LAZY_HIST_WORDS

vp, i64, i32, f32, u64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64


class GemmArgs(C.Structure):
    _fields_ = [("dtype", i32), ("layout", i32), ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp),
                ("ldc", i64), ("M", i32), ("N", i32), ("K", i32), ("a_kmod", i32), ("b_kmod", i32),
                ("addend", vp), ("ld_add", i64), ("add_rows", i32), ("add_is_T", i32), ("act", i32),
                ("out_f32", i32), ("accumulate", i32), ("alpha", f32), ("scatter_ids", vp), ("pad_id", i32),
                ("tile", i32), ("split_k", i32), ("b_batch_rows", i32), ("b_batch_stride", i64),
                ("colsum_w", vp), ("colsum_w_stride", i64), ("colsum_out", vp), ("colsum_out2", vp),
                ("c_row_blk", i32), ("c_row_valid", i32), ("c_col_blk", i32), ("c_col_valid", i32), ("a_row_ids", vp)]


class LstmDirFwd(C.Structure):
    _fields_ = [("h_prev", vp), ("ld_hprev", i64), ("c_prev", vp), ("ld_cprev", i64), ("w_hh", vp), ("ld_w", i64),
                ("gx", vp), ("ld_gx", i64), ("gx2", vp), ("ld_gx2", i64), ("gates", vp), ("ld_gates", i64), ("c_out", vp), ("ld_c", i64),
                ("h_out", vp), ("ld_h", i64), ("h_n", vp), ("ld_hn", i64), ("c_n", vp), ("ld_cn", i64),
                ("t", i32), ("capture", i32)]


class LstmDirBwd(C.Structure):
    _fields_ = [("dgates_next", vp), ("ld_dgn", i64), ("w_hh_t", vp), ("ld_wt", i64), ("dh_above", vp),
                ("ld_dha", i64), ("gates", vp), ("ld_gates", i64), ("c_t", vp), ("ld_ct", i64), ("c_prev", vp),
                ("ld_cp", i64), ("dc_carry", vp), ("ld_dcc", i64), ("dgates_out", vp), ("ld_dgo", i64),
                ("dh_n", vp), ("ld_dhn", i64), ("dc_n", vp), ("ld_dcn", i64), ("dh0_out", vp), ("ld_dh0", i64),
                ("t", i32), ("inject", i32)]


class ZeroDesc(C.Structure):
    _fields_ = [("ptr", vp), ("bytes", i64), ("chunk_start", i64)]


class HistSeg(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("bytes", i64), ("stride_bytes", i64)]


class PackDesc(C.Structure):
    _fields_ = [("src", vp), ("src2", vp), ("dst", vp), ("ld_src", i64), ("ld_dst", i64), ("R", i32), ("C", i32),
                ("transpose", i32), ("dtype", i32), ("chunk_start", i32), ("chunks", i32)]


_SIGS = {
    "vmmt_version": (i32, []),
    "vmmt_stream_create_masked": (i32, [vp, i32, i32, vp]),
    "vmmt_stream_destroy": (i32, [vp]),
    "vmmt_probe_where": (i32, [vp, i32, i32, i32, vp]),
    "vmmt_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "vmmt_gemm_group": (i32, [C.POINTER(GemmArgs), i32, vp]),
    "vmmt_gemm_group_applies": (i32, [C.POINTER(GemmArgs), i32]),
    "vmmt_lstm_step_fwd": (i32, [i32, i32, C.POINTER(LstmDirFwd), vp, i32, i32, vp]),
    "vmmt_lstm_step_bwd": (i32, [i32, i32, C.POINTER(LstmDirBwd), vp, i32, i32, i32, vp]),
    "vmmt_lstm_chain_fwd": (i32, [i32, i32, i32, C.POINTER(LstmDirFwd), vp, i32, i32, vp]),
    "vmmt_lstm_seq_sync_words": (i32, []),
    "vmmt_lstm_seq_xchg_bytes": (i64, [i32, i32, i32]),
    "vmmt_lstm_seq_fwd": (i32, [i32, i32, i32, C.POINTER(LstmDirFwd), vp, vp, i32, i32, vp, vp, vp]),
    "vmmt_lstm_seq_xchg_bytes_bwd": (i64, [i32, i32, i32]),
    "vmmt_lstm_seq_bwd": (i32, [i32, i32, i32, C.POINTER(LstmDirBwd), vp, vp, i32, i32, i32, vp, vp, vp]),
    "vmmt_lstm_chain_bwd": (i32, [i32, i32, i32, C.POINTER(LstmDirBwd), vp, i32, i32, i32, vp]),
    "vmmt_attn_fwd": (i32, [i32, vp, i64, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp]),
    "vmmt_attn_bwd": (i32, [i32, vp, i64, vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "vmmt_attn_bwd_long": (i32, [i32, vp, i64, vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp]),
    "vmmt_masked_mean": (i32, [i32, vp, i64, vp, vp, i64, i32, i32, i32, vp]),
    "vmmt_gen_npart": (i32, [i32]),
    "vmmt_gen_argmax": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "vmmt_latent_cond_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, vp]),
    "vmmt_latent_cond_bwd": (i32, [i32, vp, vp, vp, vp, vp, f32, f32, i32, f32, f32, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "vmmt_reparam_dz": (i32, [vp, i64, i32, vp, i64, vp, vp, vp, vp, i32, i32, vp]),
    "vmmt_masked_mean_bm": (i32, [i32, vp, i64, vp, vp, i64, i32, i32, i32, vp]),
    "vmmt_masked_mean_bwd": (i32, [i32, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "vmmt_gen_loss_fwd": (i32, [i32, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "vmmt_gen_loss_bwd": (i32, [i32, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, f32, vp, i64, vp]),
    "vmmt_gen_loss_bwd_db": (i32, [i32, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, f32, vp, i64, vp, i32, vp]),
    "vmmt_gen_fused_applies": (i32, [i32, i64, i64, i32, i32, i32]),
    "vmmt_gen_fused_ws_floats": (i64, [i32, i32, i32]),
    "vmmt_gen_fwd_dO": (i32, [i32, vp, i64, i32, vp, vp, i64, vp, i32, i32, i32, vp, vp, vp, i64, vp, vp]),
    "vmmt_gen_fwd_combine": (i32, [i32, vp, i64, vp, i64, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp]),
    "vmmt_gen_fwd_combine_stats": (i32, [i32, vp, i64, vp, i64, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp]),
    "vmmt_gen_fwd_combine_dO": (i32, [i32, vp, i64, vp, i64, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp]),
    "vmmt_compact_nonpad": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "vmmt_gen_fused_geometry": (i32, [i32, i32, i32, vp, vp, vp]),
    "vmmt_gen_dW_finish": (i32, [i32, vp, i64, vp, vp, i64, vp, i32, i32, i32, f32, vp, i64, vp, i32, vp, vp]),
    "vmmt_gemm_colsum_applies": (i32, [C.POINTER(GemmArgs)]),
    "vmmt_gather_rows": (i32, [i32, vp, i64, vp, vp, i64, i32, i32, vp]),
    "vmmt_scatter_add_rows": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "vmmt_colsum": (i32, [i32, vp, i64, i32, i32, i32, i32, vp, vp, vp]),
    "vmmt_rowsum": (i32, [i32, vp, i64, i32, i32, vp, vp]),
    "vmmt_dropout_mask": (i32, [i32, vp, i64, f32, u64, vp]),
    "vmmt_randn": (i32, [vp, i64, u64, vp]),
    "vmmt_mul": (i32, [i32, vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "vmmt_act_bwd": (i32, [i32, i32, vp, i64, i32, vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "vmmt_qnet_fwd": (i32, [i32, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp,
                            i32, i32, i32, i32, i32, i32, i32, vp]),
    "vmmt_latent_fwd": (i32, [i32, vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, vp]),
    "vmmt_latent_bwd": (i32, [i32, vp, vp, vp, f32, f32, i32, f32, f32, vp, vp, vp, i64, vp, i64, i32, i32, vp]),
    "vmmt_gate_fwd": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "vmmt_gate_bwd": (i32, [vp, i64, vp, vp, vp, vp, i32, i32, vp]),
    "vmmt_image_loss": (i32, [i32, vp, i64, vp, i64, i32, i32, f32, vp, i64, vp, vp]),
    "vmmt_pack": (i32, [i32, vp, vp, i64, vp, i64, i32, i32, i32, vp]),
    "vmmt_pack_multi": (i32, [vp, i32, i32, vp]),
    "vmmt_prepare_batch": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, u64, vp, i32, vp, i32, i32, vp]),
    "vmmt_zero_multi": (i32, [vp, i32, i32, vp]),
    "vmmt_beam_advance_ws_bytes": (i64, [i32, i32, i32]),
    "vmmt_beam_advance": (i32, [vp, i64, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp]),
    "vmmt_rows_select": (i32, [vp, i64, vp, vp, i64, i32, i32, vp]),
    "vmmt_history_append": (i32, [vp, i32, vp, i32, i32, vp]),
    "vmmt_standardise_rows": (i32, [vp, i64, vp, vp, i64, i32, vp]),
    "vmmt_sumsq": (i32, [vp, i64, vp, i32, vp]),
    "vmmt_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp, f32, i32, vp, vp, vp]),
    "vmmt_adam_step_ranges": (i32, [vp, vp, vp, vp, i64, i64, i64, f32, f32, f32, f32, i32, f32, vp, f32, i32, vp, i64, i64, vp, vp]),
    "vmmt_dp_norm_pack": (i32, [vp, vp, vp, vp]),
    "vmmt_dp_norm_fold": (i32, [vp, i32, vp, vp, vp]),
    "vmmt_rows_mark": (i32, [vp, i64, vp, i32, vp, vp]),
    "vmmt_rows_catchup": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, i32, vp]),
    "vmmt_rows_catchup_shadow": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, vp, i64, vp]),
    "vmmt_adam_rows_step": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, f32, i32, i32, f32, vp, f32, vp, vp]),
    "vmmt_sumsq_rows": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, vp]),
}

EXPORTS = sorted(_SIGS)
_lib = None


def lib():
    """Load libvmmt.so (once).  Raises if it is missing or lacks a declared symbol -- never falls back."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64.so.7; it must be the HIP runtime of the process (device pointers and streams come
        # from torch), so it is loaded FIRST: the loader then binds libvmmt.so to it by soname.  Loading libvmmt.so before
        # torch would pull /opt/rocm's copy as a second runtime, and every kernel launch on torch's streams would fail.
        import torch  # noqa: F401
        if not os.path.exists(os.environ.get("VMMT_LIB_PATH", LIB_PATH)):
            raise RuntimeError("libvmmt.so not built (%s): run `python -m variational_mmt_amd.build`; "
                               "there is no CPU fallback" % LIB_PATH)
        h = C.CDLL(os.environ.get("VMMT_LIB_PATH", LIB_PATH))      # override: experiment builds (tools/)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


_ERR = {1: "invalid argument", 2: "kernel launch failed"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("libvmmt: %s returned %d (%s)" % (what, rc, _ERR.get(rc, "?")))
