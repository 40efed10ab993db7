"""ctypes signatures of libdrl_hip.so (csrc/hip/*.hip). Every entry point returns a hipError_t-style int."""
from ctypes import c_double, c_float, c_int, c_longlong, c_ulonglong, c_void_p, POINTER

P, I, L, F, U = c_void_p, c_int, c_longlong, c_float, c_ulonglong
LP = POINTER(c_longlong)

SIGS = {
    # sumtree.hip
    "drl_step_begin": [P, P, P],
    "drl_step_end": [P, P],
    "drl_sumtree_sample": [P, I, LP, L, P, U, I, P, P, P, P, P, P, P, F, P],
    "drl_seq_sample": [P, I, LP, L, P, U, I, P, P, P, P, P, P, P, I, P, P, I, P, P, P, P, P, P, P, P, P, P, I, P],
    "drl_sumtree_update": [P, I, LP, L, P, P, I, P, P, P, P, P],
    "drl_sumtree_ingest": [P, I, LP, L, L, I, P, P, P, P, L, P],
    "drl_gather_rows": [P, P, P, I, I, P],
    "drl_gather_f32": [P, P, P, I, I, P],
    "drl_gather_i32": [P, P, P, I, I, P],
    # td.hip
    "drl_apex_head_td_dh": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, P, P, P, P, P, F, F, F, F, P, P, P, P],
    "drl_apex_td_dh": [P, P, I, I, P, P, P, P, P, P, F, F, F, F, P, P, P, P, P, P, P],
    "drl_apex_td": [P, P, I, I, P, P, P, P, P, P, F, F, F, F, I, P, P, P, P, P],
    "drl_wmax_put": [P, P, I, I, I, P],
    "drl_wmax_take": [P, P, I, P],
    "drl_is_weights": [P, I, P, F, P, P, I, P],
    "drl_is_wmax_put": [P, I, P, F, P, I, I, I, P],
    "drl_r2d2_td": [P, P, I, I, I, P, P, P, P, P, P, P, I, I, I, F, F, F, P, P, P, P],
    # optim.hip
    "drl_grad_sumsq": [P, L, P, POINTER(c_int), P, P, P, P, P, P],
    "drl_rmsprop_step": [P, P, P, P, L, P, I, F, F, F, F, F, P, I, P, P, L, L, L, P],
    "drl_adam_step": [P, P, P, P, P, L, P, I, F, F, F, F, F, F, P, I, P, L, L, P],
    "drl_cast_bf16": [P, P, L, P],
    # head.hip
    "drl_head_fwd": [P, I, I, P, P, P, P, P, P, P, I, U, P, P],
    "drl_head_bwd": [P, P, I, I, P, P, P, P, P, P, P, I, P],
    "drl_head_fwd_dual": [P, I, P, P, P, P, P, P, I, P, P, P, P, P, I, P],
    # gemm.hip
    "drl_gemm": [I, I, P, L, P, L, I, I, I, P, L, P, P, L, F, I, I, P],
    "drl_conv_fwd": [I, P, P, I, P, P, P, I, P],
    "drl_conv_wgrad": [I, P, P, I, P, P, I, P],
    "drl_conv_fwd_dual": [I, P, I, P, P, P, P, I, P, P, P, P],
    "drl_gemm_nt_dual": [P, I, P, P, P, P, I, P, P, P, I, I, I, P],
    "drl_conv1_fwd_dual": [P, P, I, P, P, P, P, I, P, P, P, P],
    "drl_conv1_fwd_pair": [P, P, I, P, P, P, P, P, P, P],
    # conv_direct.hip
    "drl_conv1_fwd": [P, P, P, P, I, P, P],
    "drl_conv1_set_waves": [I],
    # conv_halo.hip
    "drl_conv_halo_fwd": [I, P, I, P, P, P, P, I, P, P, P, I, P, P, P, P, P],
    "drl_conv_halo_wprep": [P, P, P, P, P],
    "drl_conv_halo_dgrad": [I, P, I, P, P, P, P],
    "drl_halo_set_busy": [I, I],
    # pingpong.hip
    "drl_pingpong": [I, I, I, I, I, I, P, P],
    "drl_allgather_bench": [I, I, I, I, I, P, P],
    # ingest.hip
    "drl_ingest_scatter": [P, I, P, P, P, I, P, P, P, L, P],
    "drl_ingest_block": [P, P, L, P, P, L, I, P, P, P, I, P, P, P, L, P],
    # ipc_mirror.hip
    "drl_ipc_alloc": [U, P, P],
    "drl_ipc_free": [P],
    "drl_ipc_open": [P, P],
    "drl_ipc_close": [P],
    "drl_copy_d2d_async": [P, P, U, P],
    "drl_device_bus_id": [I, P],
    # conv_torso.hip
    "drl_torso_fwd": [P, P, I, P, P, P, I, I, P, I, P, P, P, I, I, P, P, P, P, P],
    "drl_torso_set_busy": [I],
    "drl_torso_set_debug": [I],
    # conv_wgrad_halo.hip
    "drl_conv_wgrad_halo": [I, P, P, P, P, P, I, I, P],
    "drl_conv1_wgrad_halo": [P, P, P, P, P, P, I, I, P],
    "drl_conv_dgrad": [I, P, I, P, P, P, P],
    "drl_conv_bwd": [I, P, P, I, P, P, P, P, P, I, P, P, P],
    "drl_fc_bwd": [P, P, P, P, P, P, I, I, I, P, P, I, P, P, P, P, I, P],
    "drl_lstm_step": [P, P, P, P, P, P, P, I, I, P],
    # lstm_seq.hip
    "drl_lstm_seq_fwd_x": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P],
    "drl_lstm_bptt": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P],
    "drl_lstm_set_bptt_groups": [I],
    # lstm.hip
    "drl_lstm_bwd_point": [P, P, P, P, P, P, P, I, I, P],
    "drl_lstm_bwd_step": [P, P, P, P, P, P, P, P, P, P, P, I, I, P],
    "drl_seq_offsets": [P, I, I, L, L, P, P],
    "drl_seq_win_offsets": [P, P, I, I, P, P],
    "drl_gather_state": [P, P, I, I, P, P, P],
    "drl_gather_state2": [P, P, P, I, I, P, P, P, P, P, P, P, P, P, P, I, P],
    "drl_add2": [P, P, P, I, P],
    # conv_bwd.hip
    "drl_colsum": [P, I, I, P, P],
    # xgmi_ar.hip
    "drl_xar_create": [I, I, L, I, c_double, POINTER(c_void_p)],
    "drl_xar_create2": [I, I, L, I, c_double, I, POINTER(c_void_p)],
    "drl_xar_set_blocks": [P, I],
    "drl_xar_handles": [P, P],
    "drl_xar_open": [P, I, P, I],
    "drl_xar_all_reduce": [P, L, L, I, P],
    "drl_xar_error": [P],
    "drl_lstm_set_spin_limit": [I],
    # gemm_big.hip
    "drl_gemm_big": [I, I, P, L, P, L, I, I, I, P, L, P, P, L, I, P, P, P, P, P, P],
    "drl_xar_set_timeout": [P, c_double],
    "drl_xar_destroy": [P],
}


def declare(lib) -> None:
    for name, args in SIGS.items():
        if not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    if hasattr(lib, "drl_xar_data_ptr"):
        lib.drl_xar_data_ptr.restype = c_void_p
        lib.drl_xar_data_ptr.argtypes = [c_void_p]
        lib.drl_xar_err_ptr.restype = c_void_p
        lib.drl_xar_err_ptr.argtypes = [c_void_p]
