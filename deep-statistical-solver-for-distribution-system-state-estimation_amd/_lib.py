"""ctypes binding of libdss2_hip.so (the C ABI declared in include/dss2_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
The kernels only exist for gfx950; CPU tensors are rejected by the callers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSS2_LIB", os.path.join(_HERE, "libdss2_hip.so"))   # DSS2_LIB: diagnostic builds only

c_f32p = C.c_void_p   # all device pointers travel as integers (tensor.data_ptr())
c_i32p = C.c_void_p


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ld", C.c_int32), ("transpose", C.c_int32), ("koff", C.c_int32), ("kpad", C.c_int32),
                ("ncg", C.c_int32), ("joff", C.c_int32)]


class GemmPropArgs(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("kreal", C.c_int32), ("kpad", C.c_int32),
                ("Bp", C.c_void_p), ("bias", C.c_void_p), ("rowscale", C.c_void_p),
                ("relu_src", C.c_void_p), ("ld_relu", C.c_int64),
                ("dmask", C.c_void_p), ("ld_dmask", C.c_int64),
                ("add_src", C.c_void_p), ("ld_add", C.c_int64),
                ("Y", C.c_void_p), ("ldy", C.c_int64), ("hout", C.c_int32), ("ncg", C.c_int32),
                ("relu", C.c_int32), ("nmat", C.c_int32), ("nrb", C.c_int32), ("ntiles", C.c_int32),
                ("tile_start", C.c_void_p),
                ("rowptr", C.c_void_p), ("col", C.c_void_p), ("w", C.c_void_p),
                ("max_nnz", C.c_int32), ("ell_width", C.c_int32),
                ("prop_in", C.c_int32), ("narrow_h", C.c_int32),
                ("prebias", C.c_void_p), ("pre_rowscale", C.c_void_p), ("ell_tiles", C.c_void_p),
                ("drop_state", C.c_void_p), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float), ("drop_id", C.c_int32),
                ("b_format", C.c_int32), ("max_tile_rows", C.c_int32)]


class ChainHead(C.Structure):
    _fields_ = [("W", C.c_void_p * 4), ("bias", C.c_void_p), ("add_src", C.c_void_p), ("Y", C.c_void_p),
                ("G", C.c_void_p), ("gate", C.c_void_p), ("Xout", C.c_void_p),
                ("ld_add", C.c_int64), ("ldy", C.c_int64), ("ldg", C.c_int64), ("ld_gate", C.c_int64), ("ldxo", C.c_int64),
                ("nout", C.c_int32), ("mode", C.c_int32), ("drop_id", C.c_int32), ("pad", C.c_int32), ("wg_slab", C.c_void_p)]


class WgradArgs(C.Structure):
    _fields_ = [("G", C.c_void_p), ("ldg", C.c_int64), ("hout", C.c_int32),
                ("X", C.c_void_p), ("ldx", C.c_int64), ("hin", C.c_int32),
                ("rowscale", C.c_void_p),
                ("slab", C.c_void_p), ("n_split", C.c_int32), ("nmat", C.c_int32), ("nrb", C.c_int32),
                ("ntiles", C.c_int32),
                ("tile_start", C.c_void_p),
                ("rowptrT", C.c_void_p), ("colT", C.c_void_p), ("wT", C.c_void_p),
                ("max_nnz", C.c_int32), ("ell_width", C.c_int32), ("ell_tiles", C.c_void_p), ("rowscale2", C.c_void_p),
                ("narrow", C.c_int32), ("mfma_bf16", C.c_int32)]


class CsrBuildArgs(C.Structure):
    _fields_ = [("edge_index", C.c_void_p), ("n_edges", C.c_int64), ("n_nodes", C.c_int64), ("doubled", C.c_int32), ("no_flip", C.c_int32),
                ("rowptr", C.c_void_p), ("col", C.c_void_p), ("ent", C.c_void_p), ("perm", C.c_void_p), ("w", C.c_void_p),
                ("rowptrT", C.c_void_p), ("colT", C.c_void_p), ("entT", C.c_void_p), ("permT", C.c_void_p), ("wT", C.c_void_p),
                ("inc_rowptr", C.c_void_p), ("inc_ent", C.c_void_p), ("efrom", C.c_void_p), ("eto", C.c_void_p),
                ("deg", C.c_void_p), ("lastcut", C.c_void_p), ("meta", C.c_void_p), ("work", C.c_void_p)]


class EllBuildArgs(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("ent", C.c_void_p), ("w", C.c_void_p),
                ("rowptrT", C.c_void_p), ("colT", C.c_void_p), ("entT", C.c_void_p), ("wT", C.c_void_p),
                ("tile_start", C.c_void_p), ("ntiles", C.c_int32), ("tm", C.c_int32), ("ell_width", C.c_int32),
                ("ellT_width", C.c_int32),
                ("ell_tiles", C.c_void_p), ("ell_ent_tiles", C.c_void_p), ("ellT_tiles", C.c_void_p),
                ("ellT_ent_tiles", C.c_void_p), ("meta", C.c_void_p), ("uniform_rows", C.c_int32), ("n_nodes", C.c_int64)]


class CsrAxpyArgs(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("w", C.c_void_p), ("T", C.c_void_p), ("ldt", C.c_int64),
                ("add", C.c_void_p), ("ld_add", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64), ("bias", C.c_void_p),
                ("relu_src", C.c_void_p), ("ld_relu", C.c_int64), ("add_src", C.c_void_p), ("ld_src", C.c_int64),
                ("drop_state", C.c_void_p), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float), ("drop_id", C.c_int32),
                ("relu", C.c_int32), ("n_rows", C.c_int64), ("h", C.c_int32), ("pad_", C.c_int32)]


class ReduceDesc(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("out", C.c_void_p), ("stride", C.c_int64), ("len", C.c_int64),
                ("n_slabs", C.c_int32), ("pad_", C.c_int32)]


class ChainLayer(C.Structure):
    _fields_ = [("Bp", C.c_void_p), ("bias", C.c_void_p), ("relu_src", C.c_void_p), ("dmask", C.c_void_p),
                ("add_src", C.c_void_p), ("prebias", C.c_void_p), ("Y", C.c_void_p), ("relu", C.c_int32), ("drop_id", C.c_int32),
                ("gate_bits", C.c_void_p), ("y_bits", C.c_void_p)]


class AdamaxDesc(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_inf", C.c_void_p), ("n", C.c_int64)]


class CollateDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("chunk", C.c_int32), ("kind", C.c_int32),
                ("shared", C.c_int32), ("pad_", C.c_int32), ("nodes_per_sample", C.c_int64)]


class SgemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p * 4), ("B", C.c_void_p * 4), ("C", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("c_off", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32),
                ("ldc", C.c_int32), ("transA", C.c_int32), ("transB", C.c_int32), ("nbatch", C.c_int32),
                ("accumulate", C.c_int32)]


class WlsArgs(C.Structure):
    _fields_ = [("input", C.c_void_p), ("ld_in", C.c_int64),
                ("edge_input", C.c_void_p), ("ld_ein", C.c_int64),
                ("output", C.c_void_p), ("ld_out", C.c_int64),
                ("node_param", C.c_void_p), ("ld_np", C.c_int64),
                ("edge_param", C.c_void_p), ("ld_ep", C.c_int64),
                ("x_mean", C.c_void_p), ("x_std", C.c_void_p),
                ("edge_mean", C.c_void_p), ("edge_std", C.c_void_p),
                ("efrom", C.c_void_p), ("eto", C.c_void_p),
                ("inc_rowptr", C.c_void_p), ("inc_ent", C.c_void_p),
                ("n_nodes", C.c_int64), ("n_edges", C.c_int64),
                ("lam_v", C.c_float), ("lam_p", C.c_float), ("lam_pf", C.c_float), ("lam_reg", C.c_float),
                ("sums", C.c_void_p), ("partials", C.c_void_p), ("vminmax", C.c_void_p),
                ("apq", C.c_void_p), ("loss", C.c_void_p), ("grad_output", C.c_void_p), ("pflow", C.c_void_p),
                ("flags", C.c_int32), ("counter", C.c_void_p), ("gscale", C.c_void_p)]


WLS_VMM_CACHED, WLS_FUSED_FINISH, WLS_NO_LOSS_WRITE = 1, 2, 4


class StackDims(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_hh", C.c_int32), ("dout_inner", C.c_int32), ("dout_last", C.c_int32),
                ("skip_inner", C.c_int32), ("skip_last", C.c_int32)]


class StackArgs(C.Structure):
    _fields_ = [("dims", StackDims),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("ea", C.c_void_p), ("ldea", C.c_int64), ("wpack", C.c_void_p),
                ("tile_start", C.c_void_p), ("ntiles", C.c_int32), ("tm", C.c_int32),
                ("ell_w", C.c_void_p), ("ell_e", C.c_void_p), ("ell_width", C.c_int32),
                ("ellT_w", C.c_void_p), ("ellT_e", C.c_void_p), ("ellT_width", C.c_int32),
                ("deg_pows", C.c_void_p), ("eacache", C.c_void_p), ("xs", C.c_void_p), ("acts", C.c_void_p),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("gout", C.c_void_p), ("ldg", C.c_int64),
                ("dxbuf", C.c_void_p), ("dx_out", C.c_void_p),
                ("slab", C.c_void_p), ("slab_stride", C.c_int64), ("n_wg", C.c_int32),
                ("drop_state", C.c_void_p), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float), ("drop_stride", C.c_int32),
                ("n_nodes", C.c_int64)]


_SIGNATURES = {
    # name: (restype, argtypes)
    "dss2_last_error": (C.c_char_p, []),
    "dss2_version": (C.c_int, []),
    "dss2_debug_chain_clock_probe": (None, [C.c_void_p]),
    "dss2_plan_begin": (C.c_int, [C.POINTER(C.c_void_p)]),
    "dss2_plan_end": (C.c_int, [C.c_void_p]),
    "dss2_plan_size": (C.c_int, [C.c_void_p]),
    "dss2_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dss2_plan_destroy": (None, [C.c_void_p]),
    "dss2_topology_probe": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dss2_csr_build": (C.c_int, [C.POINTER(CsrBuildArgs), C.c_void_p]),
    "dss2_csr_build_work_ints": (C.c_int64, [C.c_int64, C.c_int64, C.c_int]),
    "dss2_tiles_uniform": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "dss2_tiles_walk": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dss2_ell_tiles_build": (C.c_int, [C.POINTER(EllBuildArgs), C.c_void_p]),
    "dss2_deg_pows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_segment_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_int64, C.c_int, C.c_void_p]),
    "dss2_gather_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "dss2_rng_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "dss2_dropout_mask": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dss2_gate_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int32, C.c_float, C.c_int,
                                 C.c_void_p]),
    "dss2_dropout_params": (None, [C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "dss2_csr_axpy": (C.c_int, [C.POINTER(CsrAxpyArgs), C.c_void_p]),
    "dss2_pack_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dss2_edge_hidden_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "dss2_edge_hidden_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "dss2_edge_combine_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "dss2_edge_combine_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]),
    "dss2_edge_tile_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "dss2_edge_tile_fwd_paired": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]),
    "dss2_edge_tile_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dss2_gemm_prop": (C.c_int, [C.POINTER(GemmPropArgs), C.c_void_p]),
    "dss2_chain_sp6_single_group_min_tiles": (C.c_int, []),
    "dss2_gemm_prop_chain": (C.c_int, [C.POINTER(GemmPropArgs), C.c_void_p, C.c_int, C.c_void_p]),
    "dss2_gemm_prop_chain_head": (C.c_int, [C.POINTER(GemmPropArgs), C.c_void_p, C.c_int, C.POINTER(ChainHead), C.c_void_p]),
    "dss2_gemm_prop_chain_head_supported": (C.c_int, [C.c_int] * 6),
    "dss2_gemm_prop_chain_head_wgrad_supported": (C.c_int, [C.c_int] * 6),
    "dss2_gemm_prop_chain_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_gemm_prop_chain16_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_gemm_prop_chain_f16_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_gemm_prop_chain_gate_words": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_gemm_prop16_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_wgrad": (C.c_int, [C.POINTER(WgradArgs), C.c_void_p]),
    "dss2_wgrad_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dss2_reduce_slabs_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "dss2_reduce_slabs": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "dss2_wls_loss_partials": (C.c_int, [C.POINTER(WlsArgs), C.c_void_p]),
    "dss2_wls_loss_grad": (C.c_int, [C.POINTER(WlsArgs), C.c_void_p]),
    "dss2_wls_loss_value": (C.c_int, [C.POINTER(WlsArgs), C.c_void_p]),
    "dss2_get_pflow": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_void_p]),
    "dss2_eval_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_eval_scratch_doubles": (C.c_int64, []),
    "dss2_small_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dss2_measure_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_void_p, C.c_int64, C.c_void_p]),
    "dss2_measure_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_void_p, C.c_int64,
                                     C.c_void_p]),
    "dss2_masked_zscore": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_masked_zscore_scratch_doubles": (C.c_int64, [C.c_int64]),
    "dss2_collate": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "dss2_csr_build_graphs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dss2_csr_build_graphs_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "dss2_collate_cursor": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dss2_accum_scalar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_collate_ragged_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dss2_collate_ragged": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "dss2_adamax_step": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "dss2_adamax_step_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "dss2_stack_wpack_words": (C.c_int64, [C.POINTER(StackDims)]),
    "dss2_stack_flat_floats": (C.c_int64, [C.POINTER(StackDims)]),
    "dss2_stack_supported": (C.c_int, [C.POINTER(StackDims), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_stack_pack": (C.c_int, [C.POINTER(StackDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_stack_forward": (C.c_int, [C.POINTER(StackArgs), C.c_void_p]),
    "dss2_stack_backward": (C.c_int, [C.POINTER(StackArgs), C.c_void_p]),
    "dss2_stack_reduce": (C.c_int, [C.POINTER(StackDims), C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_stack_fold_scratch_floats": (C.c_int64, [C.POINTER(StackDims)]),
    "dss2_adamax_step_flat": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dss2_gemm_prop_lds_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_wgrad_batched_groups": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_wgrad_y_slices": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_wgrad_lds_bytes_ex": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dss2_wgrad_lds_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libdss2_hip.so in-tree with hipcc --offload-arch=gfx950 (works without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    res = subprocess.run(["bash", script], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building libdss2_hip.so failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(res.stdout.strip())
    return LIB_PATH


def lib():
    """The loaded library with typed signatures.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the DSS2 HIP kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_RAW_STREAM = None


def stream_ptr(device) -> int:
    """hipStream_t of torch's current stream on `device` as an integer.  torch._C._cuda_getCurrentRawStream is the call
    torch's own compiled-kernel launchers use: ~0.3 us against ~5 us for torch.cuda.current_stream(device).cuda_stream,
    which a step pays 14 times."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        _RAW_STREAM = raw if raw is not None else (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    if isinstance(device, int):
        idx = device
    else:
        import torch
        idx = (device if isinstance(device, torch.device) else torch.device(device)).index
        if idx is None:
            idx = torch.cuda.current_device()
    return _RAW_STREAM(idx)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().dss2_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
