"""ctypes binding of include/grample_hip.h (the C ABI is the product boundary;
this file only declares signatures).  Fails loudly when the HIP library is
missing: there is no Python or CPU fallback for the sampling path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgrample_hip.so")

vp, i32, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
pi32, pu32, pu64, pd = C.POINTER(i32), C.POINTER(u32), C.POINTER(u64), C.POINTER(dbl)


class ModelDesc(C.Structure):
    _fields_ = [("n_vars", i32), ("card", pi32), ("fixed", pi32), ("n_funcs", i32), ("scope_off", pi32),
                ("scope_var", pi32), ("tab_off", pi32), ("table", pd)]


class SamplerOpts(C.Structure):
    _fields_ = [("struct_size", u32), ("n_chains", u32), ("n_variants", u32), ("device", i32), ("seed", u64),
                ("chain_offset", u64), ("variant_offset", u32), ("flags", u32), ("block_threads", u32),
                ("fast_form", u32)]


class SamplerInfo(C.Structure):
    _fields_ = [("struct_size", u32), ("block_threads", u32), ("n_blocks", u32), ("lds_bytes", u32), ("path", u32),
                ("max_card", u32), ("tables_in_lds", u32), ("n_launches_last", u32), ("kernel_ms_last", dbl),
                ("step", u64)]


class Position(C.Structure):
    _fields_ = [("struct_size", u32), ("n_active", u32), ("step", u64), ("counted", u64), ("skipped_updates", u64)]


# name -> (restype, argtypes); mirrors include/grample_hip.h one to one
SIGNATURES = {
    "gr_version": (C.c_char_p, []),
    "gr_last_error": (C.c_char_p, []),
    "gr_device_count": (C.c_int, []),
    "gr_model_create": (vp, [C.POINTER(ModelDesc)]),
    "gr_model_from_uai": (vp, [C.c_char_p, C.c_size_t]),
    "gr_model_from_file": (vp, [C.c_char_p, C.c_int]),
    "gr_model_apply_evidence": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "gr_model_clone": (vp, [vp]),
    "gr_model_destroy": (None, [vp]),
    "gr_model_n_vars": (C.c_int, [vp]),
    "gr_model_n_funcs": (C.c_int, [vp]),
    "gr_model_total_card": (C.c_int, [vp]),
    "gr_model_cards": (C.c_int, [vp, pi32]),
    "gr_model_fixed": (C.c_int, [vp, pi32]),
    "gr_model_func_arity": (C.c_int, [vp, C.c_int]),
    "gr_model_func_tabsize": (C.c_int, [vp, C.c_int]),
    "gr_model_func_scope": (C.c_int, [vp, C.c_int, pi32]),
    "gr_model_func_table": (C.c_int, [vp, C.c_int, pd]),
    "gr_model_func_eval": (C.c_int, [vp, C.c_int, pi32, C.c_int, pd]),
    "gr_model_systematic_order": (C.c_int, [vp, pi32, pi32, pi32]),
    "gr_solution_from_text": (vp, [C.c_char_p, C.c_size_t]),
    "gr_solution_from_file": (vp, [C.c_char_p]),
    "gr_solution_destroy": (None, [vp]),
    "gr_solution_n_vars": (C.c_int, [vp]),
    "gr_solution_total_card": (C.c_int, [vp]),
    "gr_solution_cards": (C.c_int, [vp, pi32]),
    "gr_solution_marginals": (C.c_int, [vp, pd]),
    "gr_error_suite": (C.c_int, [i32, pi32, pi32, pi32, pd, pd, pd]),
    "gr_sampler_create": (vp, [vp, C.POINTER(SamplerOpts)]),
    "gr_sampler_destroy": (None, [vp]),
    "gr_collapse": (C.c_int, [vp, u32, i32, pi32, pd]),
    "gr_blanket_size": (C.c_int, [vp, u32, i32]),
    "gr_function_count": (C.c_int, [vp, u32, i32]),
    "gr_burnin": (C.c_int, [vp, u64]),
    "gr_advance": (C.c_int, [vp, u64]),
    "gr_total_samples": (C.c_int, [vp, pu64]),
    "gr_counts": (C.c_int, [vp, pu64, pu64]),
    "gr_counts_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "gr_counts_upload": (C.c_int, [vp, pu64]),
    "gr_get_position": (C.c_int, [vp, C.POINTER(Position), pu64, pu32]),
    "gr_set_position": (C.c_int, [vp, C.POINTER(Position), pu64, pu32]),
    "gr_state_download": (C.c_int, [vp, C.POINTER(C.c_uint8)]),
    "gr_state_upload": (C.c_int, [vp, C.POINTER(C.c_uint8)]),
    "gr_merge": (C.c_int, [C.POINTER(vp), C.c_int, pd, pi32]),
    "gr_merge_local": (C.c_int, [vp, pd, pi32, pd]),
    "gr_allreduce_counts": (C.c_int, [C.POINTER(vp), C.c_int, pu64]),
    "gr_rccl_calls": (u64, []),
    "gr_state": (C.c_int, [vp, u32, pi32]),
    "gr_set_state": (C.c_int, [vp, u32, pi32]),
    "gr_conditional": (C.c_int, [vp, u32, i32, pi32, pd]),
    "gr_sampler_get_info": (C.c_int, [vp, C.POINTER(SamplerInfo)]),
    "gr_set_active_groups": (C.c_int, [vp, u32]),
    "gr_active_groups": (C.c_int, [vp]),
    "gr_advance_window": (C.c_int, [vp, u64]),
    "gr_advance_chain": (C.c_int, [vp, u32, pu64]),
    "gr_window_halves": (C.c_int, [vp, pd, pd, pd]),
    "gr_set_window_halves": (C.c_int, [vp, pd, pd, dbl]),
    "gr_chain_convergence": (C.c_int, [i32, i32, pi32, pi32, pi32, pd, pd, pd, dbl, i32, pd]),
    "gr_adapt_select": (C.c_int, [vp, pi32, pd, u32, pi32, pu32]),
    "gr_adapt": (C.c_int, [vp, u32, pi32, pu32, pd]),
    "gr_activate_group": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_int32)]),
    "gr_philox4x32_10": (None, [pu32, pu32, pu32]),
}

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                "grample_amd has no CPU or pure-Python sampling path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB
