"""ctypes prototypes for include/p5hip.h (one table, used by the product loader and by the test-only emulator
loader so both bind exactly the symbols the header declares)."""
import ctypes as C

c_i64p = C.POINTER(C.c_int64)
vp, i32, i64, f32, f64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint32


class P5Config(C.Structure):
    _fields_ = [
        ("vocab_size", i32), ("d_model", i32), ("d_kv", i32), ("d_ff", i32), ("n_enc_layers", i32),
        ("n_dec_layers", i32), ("n_heads", i32), ("rel_buckets", i32), ("rel_max_distance", i32),
        ("whole_word_size", i32), ("gated_gelu", i32), ("dtype", i32), ("eps", f32), ("dropout", f32),
        ("pad_id", i32), ("eos_id", i32),
    ]


class P5GemmProblem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("aux", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32),
                ("ldaux", i32), ("epi", i32), ("c_f32", i32), ("splitk", i32), ("alpha", f32), ("rowss", vp), ("rowss_eps", f32), ("ssq_out", vp), ("rowss_nt", i32), ("ssq_nt", i32),
                ("C2", vp), ("ldc2", i32), ("gate_F", i32),
                ("nb_dot", vp), ("nb_dot_nt", i32), ("nb_rin", vp), ("nb_rout", vp), ("nb_w", vp), ("nb_dw", vp)]


# name -> (restype, argtypes)
PROTOTYPES = {
    "p5_last_error": (C.c_char_p, []),
    "p5_set_option": (i32, [C.c_char_p, i32]),
    "p5_abi_version": (i32, []),
    "p5_is_emulator": (i32, []),
    "p5_engine_create": (i32, [C.POINTER(P5Config), C.POINTER(vp)]),
    "p5_engine_destroy": (i32, [vp]),
    "p5_param_count": (i64, [vp]),
    "p5_param_table": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
    "p5_engine_bind": (i32, [vp, vp, vp, vp, vp, vp, i32, vp]),
    "p5_refresh_shadow": (i32, [vp, vp]),
    "p5_transposed_bytes": (i64, [vp]),
    "p5_engine_bind_transposed": (i32, [vp, vp, vp]),
    "p5_refresh_transposed": (i32, [vp, vp]),
    "p5_engine_set_side_stream": (i32, [vp, vp]),
    "p5_train_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "p5_forward": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i64, vp]),
    "p5_forward_loss": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i64, vp]),
    "p5_backward_num_stages": (i32, [vp]),
    "p5_backward_stage": (i32, [vp, vp, i32, vp]),
    "p5_backward": (i32, [vp, vp, vp]),
    "p5_backward_stage_range": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    "p5_backward_final_range": (i32, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "p5_backward_stage_pairs": (i32, [vp, i32]),
    "p5_backward_staged": (i32, [vp, vp, vp, C.POINTER(i64), i32, C.POINTER(i32)]),
    "p5_backward_staged_wait": (i32, [vp, i32, vp]),
    "p5_allreduce_range": (i32, [vp, i32, vp, vp, vp]),
    "p5_allreduce_sum": (i32, [vp, i64, i32, vp, vp]),
    "p5_grad_sumsq": (i32, [vp, i64, vp, vp]),
    "p5_adamw_step": (i32, [vp, vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, f64, f64, f64, i32, vp]),
    "p5_engine_adamw_step": (i32, [vp, vp, vp, vp, f64, f64, f64, f64, f64, f64, f64, i32, C.POINTER(i32), vp]),
    "p5_decode_fold_count": (i64, [vp]),
    "p5_engine_bind_decode_fold": (i32, [vp, vp]),
    "p5_refresh_decode_fold": (i32, [vp, vp]),
    "p5_engine_grads_zeroed": (i32, [vp]),
    "p5_engine_clear_grads": (i32, [vp, vp]),
    "p5_engine_discard_grads": (i32, [vp]),
    "p5_generate_workspace_bytes": (i64, [vp, i32, i32, i32, i32, i32, i32]),
    "p5_generate": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]),
    "p5_generate_set_forced_prefix": (i32, [vp, vp, vp, i32]),
    "p5_generate_history_count": (i64, [i32, i32, i32]),
    "p5_generate_draft": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i64, vp]),
    "p5_verify_workspace_bytes": (i64, [vp, i32, i32, i32, i32, i32, i32, i32]),
    "p5_verify_begin": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, i64]),
    "p5_verify_encoder_output": (vp, [vp]),
    "p5_generate_set_encoder_output": (i32, [vp, vp]),
    "p5_verify_plan": (i32, [vp, vp, vp]),
    "p5_verify_plan_header": (vp, [vp]),
    "p5_verify_row_capacity": (i32, [i32, i32]),
    "p5_verify_encode": (i32, [vp, vp, vp, vp, vp]),
    "p5_verify_run": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
    "p5_generate_timing": (i32, [vp, i32, C.POINTER(f32), C.POINTER(f32)]),
    "p5_decode_begin": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, i64, vp]),
    "p5_decode_step": (i32, [vp, vp]),
    "p5_decode_done_flag": (vp, [vp]),
    "p5_decode_finish": (i32, [vp, vp, vp, vp, vp]),
    "p5_encode": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, i64, vp]),
    "p5_op_gemm": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, u32, f32, vp]),
    "p5_op_gemm_group": (i32, [i32, i32, i32, C.POINTER(P5GemmProblem), vp, u32, f32, vp]),
    "p5_op_rmsnorm_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, f32, vp]),
    "p5_op_rmsnorm_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "p5_op_attn_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, u32, f32, vp]),
    "p5_op_attn_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                            i32, i32, i32, i32, vp, u32, f32, vp]),
    "p5_op_attn_bwd_dot": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                i32, i32, i32, i32, vp, u32, f32, vp, vp]),
    "p5_op_ce_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "p5_op_dec_cross_attn": (i32, [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "p5_op_skinny_gemm": (i32, [i32, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, f32, vp]),
    "p5_op_tr_probe": (i32, [vp, vp, vp]),
    "p5_profile_begin": (i32, []),
    "p5_profile_end": (i32, [C.c_char_p, i32]),
}


def bind(cdll):
    """Attach restype/argtypes for every symbol of include/p5hip.h; raises AttributeError if one is missing."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


class P5Error(RuntimeError):
    pass


def check(lib, rc, what=""):
    if rc != 0:
        msg = lib.p5_last_error()
        raise P5Error(f"{what} failed: {msg.decode() if msg else rc}")
