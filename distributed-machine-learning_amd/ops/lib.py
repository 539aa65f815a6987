"""ctypes binding of ``libmitune_hip.so`` (the hand-written gfx950 kernels + C++ step driver).

The descriptor classes mirror the structs of ``csrc/mt_common.h`` / ``csrc/runtime.hip`` field for
field; ``load()`` checks ``sizeof`` of every struct against the library so a drift fails at import
time rather than as a wild pointer on the GPU.  There is deliberately no fallback: on a machine with a
GPU a missing or stale library is an error, never a silent eager-PyTorch path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build

MT_MAX_LAYERS = 8

c_void_p, c_int32, c_uint32, c_float, c_size_t = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_size_t


class GemmGroup(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("aux", c_void_p), ("aux_out", c_void_p),
                ("seed", c_void_p),
                ("I", c_int32), ("J", c_int32), ("KK", c_int32),
                ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldaux", c_int32),
                ("active_steps", c_int32),
                ("thresh", c_uint32), ("scale", c_float), ("gflags", c_int32), ("site", c_int32)]


class Tile(C.Structure):
    _fields_ = [("group", c_int32), ("ti", C.c_int16), ("tj", C.c_int16)]


class GatherSlot(C.Structure):
    _fields_ = [("X0", c_void_p), ("y", c_void_p), ("key", c_uint32 * 4), ("B", c_int32), ("active_steps", c_int32)]


class HeadSlot(C.Structure):
    _fields_ = [("H", c_void_p), ("dH", c_void_p), ("w", c_void_p), ("b", c_void_p), ("gw", c_void_p),
                ("gb", c_void_p), ("y", c_void_p), ("loss_acc", c_void_p), ("metrics", c_void_p),
                ("B", c_int32), ("W", c_int32), ("ldh", c_int32), ("loss_type", c_int32),
                ("active_steps", c_int32), ("pad", c_int32)]


class OptSlot(C.Structure):
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("wd", c_float),
                ("momentum", c_float), ("clip", c_float),
                ("opt_type", c_int32), ("warmup", c_int32), ("total", c_int32), ("n_params", c_int32),
                ("active_steps", c_int32), ("step_base", c_int32), ("first_end", c_int32),
                ("first_splits", c_int32), ("pad", c_int32)]


class MlpPlan(C.Structure):
    _fields_ = [("n_hidden", c_int32), ("n_active", c_int32), ("max_batch", c_int32), ("max_width", c_int32),
                ("max_params", c_int32), ("F", c_int32), ("Fpad", c_int32), ("hb", c_int32),
                ("n_rows", c_uint32), ("use_clip", c_int32),
                ("X", c_void_p), ("Y", c_void_p), ("gather", c_void_p),
                ("fwd_groups", c_void_p * MT_MAX_LAYERS), ("fwd_tiles", c_void_p * MT_MAX_LAYERS),
                ("dgrad_groups", c_void_p * MT_MAX_LAYERS), ("dgrad_tiles", c_void_p * MT_MAX_LAYERS),
                ("wgrad_groups", c_void_p * MT_MAX_LAYERS), ("wgrad_tiles", c_void_p * MT_MAX_LAYERS),
                ("fwd_ntiles", c_int32 * MT_MAX_LAYERS), ("dgrad_ntiles", c_int32 * MT_MAX_LAYERS),
                ("wgrad_ntiles", c_int32 * MT_MAX_LAYERS),
                ("head", c_void_p), ("P", c_void_p), ("G", c_void_p), ("M", c_void_p), ("V", c_void_p),
                ("Pb", c_void_p), ("slot_stride", c_size_t), ("opt", c_void_p), ("partials", c_void_p),
                ("active_ids", c_void_p), ("head_scratch", c_void_p), ("max_blocks", c_int32), ("ksplits", c_int32),
                ("gsplit_stride", c_size_t), ("ksplits0", c_int32), ("f32", c_int32),
                ("fwd_big", c_int32 * MT_MAX_LAYERS), ("bwd_big", c_int32 * MT_MAX_LAYERS),
                ("rb_blocks", c_void_p), ("rb_nblocks", c_int32), ("rb_mt", c_int32),
                ("wg_all_groups", c_void_p), ("wg_all_tiles", c_void_p), ("wg_all_ntiles", c_int32),
                ("wg_all_edge", c_int32), ("rb_flags", c_void_p),
                ("rb_ph_blocks", c_void_p * 3), ("rb_ph_nblocks", c_int32 * 3), ("rb_ph_from", c_int32 * 3),
                ("rb_nflags", c_int32), ("rb_pad", c_int32)]


class DpPlan(C.Structure):
    """Mirror of ``MtDpPlan`` (csrc/dp.hip): one data-parallel MLP trial's step, as the C++ driver walks it."""
    _fields_ = [("n_hidden", c_int32), ("batch", c_int32), ("F", c_int32), ("hb", c_int32), ("n_rows", c_uint32),
                ("world", c_int32), ("f32", c_int32), ("use_clip", c_int32), ("clip_blocks", c_int32), ("pad0", c_int32),
                ("widths", c_int32 * MT_MAX_LAYERS), ("ksplits", c_int32 * MT_MAX_LAYERS),
                ("w_off", C.c_int64 * (MT_MAX_LAYERS + 1)), ("b_off", C.c_int64 * (MT_MAX_LAYERS + 1)),
                ("bucket_lo", C.c_int64 * (MT_MAX_LAYERS + 1)), ("bucket_hi", C.c_int64 * (MT_MAX_LAYERS + 1)),
                ("numel", C.c_int64), ("X", c_void_p), ("Y", c_void_p), ("X0", c_void_p), ("ybuf", c_void_p),
                ("H", c_void_p * MT_MAX_LAYERS), ("dZ", c_void_p * MT_MAX_LAYERS),
                ("P", c_void_p), ("W", c_void_p), ("G32", c_void_p), ("Gb", c_void_p), ("Gr", c_void_p), ("M", c_void_p), ("V", c_void_p),
                ("head", c_void_p), ("opt", c_void_p), ("ids", c_void_p), ("head_scratch", c_void_p),
                ("all_reduce", c_void_p), ("comm", c_void_p), ("nccl_bf16", c_int32), ("nccl_avg", c_int32),
                ("nccl_f32", c_int32), ("pad1", c_int32), ("clip_partials", c_void_p),
                ("comm_stream", c_void_p), ("ready", c_void_p * (MT_MAX_LAYERS + 1)), ("reduced", c_void_p)]


HM_MAXL = 5


class FlatOpt(C.Structure):
    """Device-resident optimiser record of ``csrc/flat_optim.hip`` (hyper-parameters + step counter)."""
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("wd", c_float),
                ("momentum", c_float), ("clip", c_float), ("opt_type", c_int32), ("warmup", c_int32),
                ("total", c_int32), ("step", c_int32), ("pad", c_int32)]


# numpy views of the same layouts, for building whole descriptor tables vectorised
GEMM_GROUP_DTYPE = np.dtype([("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("aux", "<u8"), ("aux_out", "<u8"),
                             ("seed", "<u8"),
                             ("I", "<i4"), ("J", "<i4"), ("KK", "<i4"), ("lda", "<i4"), ("ldb", "<i4"),
                             ("ldc", "<i4"), ("ldaux", "<i4"), ("active_steps", "<i4"),
                             ("thresh", "<u4"), ("scale", "<f4"), ("gflags", "<i4"), ("site", "<i4")], align=True)
GROUP_ACT, GROUP_ADD = 1, 2


def gemm_group(A, B, C, aux, aux_out, I, J, KK, lda, ldb, ldc, ldaux, active_steps, seed=0, thresh=0, scale=0.0,
               gflags=0, site=0):
    """One record of ``GEMM_GROUP_DTYPE`` (field order of ``MtGemmGroup``); the trailing per-group extras default off."""
    return (A, B, C, aux, aux_out, seed, I, J, KK, lda, ldb, ldc, ldaux, active_steps, thresh, scale, gflags, site)
TILE_DTYPE = np.dtype([("group", "<i4"), ("ti", "<i2"), ("tj", "<i2")], align=True)
RB_BLOCK_DTYPE = np.dtype([("slot", "<i4"), ("row0", "<i4")], align=True)
GATHER_DTYPE = np.dtype([("X0", "<u8"), ("y", "<u8"), ("key", "<u4", (4,)), ("B", "<i4"), ("active_steps", "<i4")],
                        align=True)
HEAD_DTYPE = np.dtype([("H", "<u8"), ("dH", "<u8"), ("w", "<u8"), ("b", "<u8"), ("gw", "<u8"), ("gb", "<u8"),
                       ("y", "<u8"), ("loss_acc", "<u8"), ("metrics", "<u8"), ("B", "<i4"), ("W", "<i4"),
                       ("ldh", "<i4"), ("loss_type", "<i4"), ("active_steps", "<i4"), ("pad", "<i4")], align=True)
OPT_DTYPE = np.dtype([("lr", "<f4"), ("beta1", "<f4"), ("beta2", "<f4"), ("eps", "<f4"), ("wd", "<f4"),
                      ("momentum", "<f4"), ("clip", "<f4"), ("opt_type", "<i4"), ("warmup", "<i4"),
                      ("total", "<i4"), ("n_params", "<i4"), ("active_steps", "<i4"), ("step_base", "<i4"),
                      ("first_end", "<i4"), ("first_splits", "<i4"), ("pad", "<i4")], align=True)

TABLE_META_DTYPE = np.dtype([("trial_index", "<f8"), ("epoch", "<f8"), ("steps", "<i4"), ("live", "<i4")], align=True)

_lib: Optional[C.CDLL] = None


class KernelLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first when the sources changed and a compiler is present) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if build_if_missing and not _build.is_current() and os.environ.get("LOCAL_RANK", "0") != "0":
        # one process per GPU: local rank 0 rebuilds a stale library, the others wait for it instead of racing hipcc
        import time
        deadline = time.time() + 600
        while not _build.is_current() and time.time() < deadline:
            time.sleep(0.5)
    if build_if_missing and not _build.is_current():
        try:
            _build.build(verbose=False)
        except Exception as e:       # no hipcc on this machine: fall through to whatever was shipped in-tree
            if not os.path.exists(path):
                raise KernelLibraryError(f"{path} is missing and could not be built: {e}") from e
    if not os.path.exists(path):
        raise KernelLibraryError(f"{path} is missing: run `python -m mitune.ops.build`")
    lib = C.CDLL(path)
    sizes = (C.c_int * 8)()
    n = lib.mt_abi_sizes(sizes)
    expect = [C.sizeof(GemmGroup), C.sizeof(Tile), C.sizeof(GatherSlot), C.sizeof(HeadSlot), C.sizeof(OptSlot),
              C.sizeof(MlpPlan)]
    got = list(sizes[:n])
    if got != expect:
        raise KernelLibraryError(f"descriptor layout mismatch between lib.py {expect} and {path} {got}")
    np_sizes = [GEMM_GROUP_DTYPE.itemsize, TILE_DTYPE.itemsize, GATHER_DTYPE.itemsize, HEAD_DTYPE.itemsize,
                OPT_DTYPE.itemsize]
    if np_sizes != expect[:5]:
        raise KernelLibraryError(f"numpy descriptor dtypes {np_sizes} do not match ctypes {expect[:5]}")

    lib.mt_gemm_launch.argtypes = [C.c_int, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   c_size_t, C.c_int, c_void_p]
    lib.mt_gemm_bwd_launch.argtypes = [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int,
                                       c_size_t, C.c_int, c_void_p]
    lib.mt_gemm_bwd_launch.restype = C.c_int
    lib.mt_gather_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int, c_void_p, c_void_p, c_uint32, C.c_int,
                                     C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_gather_f32_launch.argtypes = lib.mt_gather_launch.argtypes
    lib.mt_gemm_f32_launch.argtypes = [C.c_int, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_size_t, C.c_int, c_void_p]
    lib.mt_gemm_f32_bwd_launch.argtypes = [c_void_p, c_void_p, C.c_int, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int,
                                           c_size_t, C.c_int, c_void_p]
    lib.mt_head_loss_f32_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p, C.c_int,
                                            c_void_p]
    lib.mt_head_eval_f32_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p,
                                            c_void_p]
    for fn in (lib.mt_gather_f32_launch, lib.mt_gemm_f32_launch, lib.mt_gemm_f32_bwd_launch, lib.mt_head_loss_f32_launch,
               lib.mt_head_eval_f32_launch):
        fn.restype = C.c_int
    lib.mt_head_loss_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p, c_void_p]
    lib.mt_head_eval_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p,
                                        c_void_p]
    lib.mt_optim_launch.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                    C.c_int, C.c_int, c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_size_t, c_void_p]
    lib.mt_optim_launch_bf16g.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                          c_void_p, C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_optim_launch_bf16g.restype = C.c_int
    lib.mt_head_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.mt_head_scratch_floats.restype = C.c_long
    lib.mt_attn_fwd_launch.argtypes = [c_void_p] * 5 + [c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float, c_float,
                                                        C.c_int, c_void_p]
    lib.mt_attn_bwd_launch.argtypes = [c_void_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_int, c_float, c_float, C.c_int,
                                                        c_void_p]
    lib.mt_attn_fwd_strided_launch.argtypes = [c_void_p] * 6 + [C.c_int] * 4 + [c_float, c_float, C.c_int, C.c_int,
                                                                   C.c_long, C.c_long, C.c_long, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_attn_bwd_strided_launch.argtypes = [c_void_p] * 9 + [C.c_int] * 4 + [c_float, c_float, C.c_int, C.c_int,
                                                                   C.c_long, C.c_long, C.c_long, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_attn_fwd_strided_launch.restype = lib.mt_attn_bwd_strided_launch.restype = C.c_int
    lib.mt_attn_exact_level.argtypes = [C.c_int]
    lib.mt_attn_exact_level.restype = C.c_int
    lib.mt_ln_fwd_launch.argtypes = [c_void_p] * 8 + [C.c_int, C.c_int, c_float, c_float, c_void_p, c_float, c_void_p]
    lib.mt_ln_bwd_launch.argtypes = [c_void_p] * 8 + [C.c_int, C.c_int, c_float, c_void_p, c_float, c_void_p]
    lib.mt_ln_bwd_blocks.argtypes = [C.c_int]
    for fn in (lib.mt_attn_fwd_launch, lib.mt_attn_bwd_launch, lib.mt_ln_fwd_launch, lib.mt_ln_bwd_launch,
               lib.mt_ln_bwd_blocks):
        fn.restype = C.c_int
    if lib.mt_flat_opt_sizeof() != C.sizeof(FlatOpt):
        raise KernelLibraryError("FlatOpt layout mismatch between lib.py and the library")
    if lib.mt_grad_seg_sizeof() != 24:
        raise KernelLibraryError("MtGradSeg is not 24 bytes")
    lib.mt_flat_optim_seg_launch.argtypes = [c_void_p] * 4 + [C.c_int, c_void_p, c_void_p, C.c_int, c_void_p]
    lib.mt_flat_optim_seg_launch.restype = C.c_int
    lib.mt_flat_optim_blocks.argtypes = [C.c_long]
    lib.mt_flat_optim_launch.argtypes = [c_void_p] * 4 + [C.c_long, c_void_p, c_void_p, C.c_int, C.c_int, c_void_p]
    lib.mt_posenc_fwd_launch.argtypes = [c_void_p] * 3 + [C.c_long, C.c_int, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_posenc_bwd_launch.argtypes = [c_void_p] * 2 + [C.c_long, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_wgrad_single_launch.argtypes = [c_void_p] * 4 + [C.c_int] * 4 + [c_size_t, c_void_p]
    lib.mt_wgrad_single_launch.restype = C.c_int
    lib.mt_gemm_single_launch.argtypes = [C.c_int] + [c_void_p] * 4 + [C.c_int] * 8 + [c_void_p]
    lib.mt_gemm_f32_single_launch.argtypes = [C.c_int] + [c_void_p] * 5 + [C.c_int] * 7 + [c_void_p]
    lib.mt_gemm_single_launch.restype = lib.mt_gemm_f32_single_launch.restype = C.c_int
    lib.mt_wgrad_f32_single_launch.argtypes = [c_void_p] * 3 + [C.c_int] * 4 + [c_size_t, C.c_int, c_void_p]
    lib.mt_slab_sum_launch.argtypes = [c_void_p, c_void_p, C.c_long, C.c_int, c_size_t, c_void_p]
    lib.mt_gemm_p8_min_tiles.argtypes = [C.c_int]
    lib.mt_gemm_p8_min_tiles.restype = C.c_int
    lib.mt_gemm_p8_option.argtypes = [C.c_int, C.c_int]
    lib.mt_gemm_p8_option.restype = C.c_int
    lib.mt_wgrad_f32_single_launch.restype = lib.mt_slab_sum_launch.restype = C.c_int
    if lib.mt_table_meta_sizeof() != TABLE_META_DTYPE.itemsize:
        raise KernelLibraryError("MtTableMeta layout mismatch between lib.py and the library")
    lib.mt_stamp_launch.argtypes = [c_void_p, c_void_p]
    lib.mt_metric_table_launch.argtypes = [c_void_p, c_void_p, c_void_p, C.c_int, C.c_int, C.c_int, c_void_p, c_void_p, c_void_p]
    lib.mt_stamp_launch.restype = lib.mt_metric_table_launch.restype = C.c_int
    if lib.mt_dp_plan_sizeof() != C.sizeof(DpPlan):
        raise KernelLibraryError(f"MtDpPlan is {lib.mt_dp_plan_sizeof()} bytes in the library, {C.sizeof(DpPlan)} in lib.py")
    lib.mt_dp_train_steps.argtypes = [C.POINTER(DpPlan), c_void_p, C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_dp_eval.argtypes = [C.POINTER(DpPlan), c_void_p, C.c_int, C.c_int, c_void_p]
    lib.mt_dp_train_steps.restype = lib.mt_dp_eval.restype = C.c_int
    lib.mt_small_linear_bwd_launch.argtypes = [c_void_p] * 6 + [C.c_int] * 3 + [c_void_p]
    lib.mt_small_linear_bwd_launch.restype = C.c_int
    lib.mt_small_linear_bwd_tiled_launch.argtypes = [c_void_p] * 6 + [C.c_int] * 4 + [c_void_p]
    lib.mt_small_linear_bwd_tiled_launch.restype = C.c_int
    lib.mt_cursor_batch_launch.argtypes = [c_void_p, c_void_p, C.c_long, c_void_p, c_void_p, C.c_long, C.c_long, C.c_long,
                                           c_void_p, c_void_p, C.c_int, c_void_p]
    lib.mt_cursor_batch_launch.restype = C.c_int
    lib.mt_mse_fwd_launch.argtypes = [c_void_p, c_void_p, C.c_long, c_void_p, c_void_p, c_void_p, c_void_p, C.c_long, c_void_p]
    lib.mt_mse_fwd_launch.restype = C.c_int
    lib.mt_seed_refill_launch.argtypes = [c_void_p, c_void_p, C.c_int, C.c_int64, c_void_p]
    lib.mt_seed_refill_launch.restype = C.c_int
    lib.mt_keep_mask_launch.argtypes = [c_void_p, C.c_long, c_void_p, c_float, c_void_p]
    lib.mt_dropout_launch.argtypes = [c_void_p, c_void_p, C.c_long, c_void_p, c_float, C.c_int, c_void_p]
    lib.mt_keep_mask_launch.restype = lib.mt_dropout_launch.restype = C.c_int
    lib.mt_relu_dropout_fwd_launch.argtypes = [c_void_p, c_void_p, C.c_long, c_void_p, c_float, C.c_int, c_void_p]
    lib.mt_relu_dropout_bwd_launch.argtypes = [c_void_p, c_void_p, c_void_p, C.c_long, c_float, C.c_int, c_void_p]
    lib.mt_relu_dropout_fwd_launch.restype = lib.mt_relu_dropout_bwd_launch.restype = C.c_int
    lib.mt_reglu_fwd_launch.argtypes = [c_void_p, c_void_p, C.c_long, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_reglu_bwd_launch.argtypes = [c_void_p, c_void_p, c_void_p, C.c_long, C.c_int, c_void_p, c_float, c_void_p]
    lib.mt_ln_fwd_mixed_launch.argtypes = [c_void_p, C.c_int] + [c_void_p] * 7 + [C.c_int, C.c_int, c_float, c_float, c_void_p,
                                                                                c_float, c_void_p]
    lib.mt_ln_bwd_mixed_launch.argtypes = [c_void_p] * 6 + [C.c_int, c_void_p, c_void_p, C.c_int, C.c_int, c_float, c_void_p,
                                                             c_float, c_void_p]
    for fn in (lib.mt_reglu_fwd_launch, lib.mt_reglu_bwd_launch, lib.mt_ln_fwd_mixed_launch, lib.mt_ln_bwd_mixed_launch):
        fn.restype = C.c_int
    lib.mt_tokens_fwd_launch.argtypes = [c_void_p] * 6 + [C.c_long, C.c_int, C.c_int, c_void_p]
    lib.mt_tokens_bwd_launch.argtypes = [c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_colsum_launch.argtypes = [c_void_p, c_void_p, C.c_long, C.c_int, c_void_p]
    lib.mt_tokens_fwd_launch.restype = lib.mt_tokens_bwd_launch.restype = lib.mt_colsum_launch.restype = C.c_int
    lib.mt_cls_head_fwd_launch.argtypes = [c_void_p] * 7 + [C.c_int, C.c_int, c_float, c_void_p]
    lib.mt_cls_head_bwd_launch.argtypes = [c_void_p] * 8 + [C.c_int, C.c_int, c_void_p]
    lib.mt_cls_head_bwd_blocks.argtypes = [C.c_int]
    lib.mt_cls_head_fwd_launch.restype = lib.mt_cls_head_bwd_launch.restype = lib.mt_cls_head_bwd_blocks.restype = C.c_int
    lib.mt_dwconv_chunks.argtypes = [C.c_long]
    lib.mt_dwconv_launch.argtypes = [c_void_p] * 4 + [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_dwconv_wgrad_launch.argtypes = [c_void_p] * 3 + [C.c_long, C.c_int, C.c_int, C.c_int, c_void_p]
    for fn in (lib.mt_flat_optim_blocks, lib.mt_flat_optim_launch, lib.mt_posenc_fwd_launch, lib.mt_posenc_bwd_launch,
               lib.mt_dwconv_chunks, lib.mt_dwconv_launch, lib.mt_dwconv_wgrad_launch):
        fn.restype = C.c_int
    lib.mt_rowblock_max_width.restype = C.c_int
    lib.mt_mlp_train_steps.argtypes = [C.POINTER(MlpPlan), C.c_int, C.c_int, c_void_p]
    lib.mt_mlp_eval.argtypes = [C.POINTER(MlpPlan), C.c_int, C.c_int, c_void_p]
    lib.mt_mlp_graph_create.argtypes = [C.POINTER(MlpPlan), C.c_int, c_void_p, c_void_p, C.POINTER(C.c_int)]
    lib.mt_mlp_graph_create.restype = c_void_p
    lib.mt_mlp_graph_replay.argtypes = [c_void_p, C.c_int, c_void_p]
    lib.mt_mlp_graph_replay.restype = C.c_int
    lib.mt_mlp_graph_destroy.argtypes = [c_void_p]
    lib.mt_mlp_graph_destroy.restype = None
    lib.mt_mlp_train_steps_at.argtypes = [C.POINTER(MlpPlan), c_void_p, C.c_int, c_void_p]
    lib.mt_mlp_train_steps_at.restype = C.c_int
    lib.mt_mlp_gather_at.argtypes = [C.POINTER(MlpPlan), c_void_p, c_void_p]
    lib.mt_mlp_gather_at.restype = C.c_int
    for fn in (lib.mt_gemm_launch, lib.mt_gather_launch, lib.mt_head_loss_launch, lib.mt_head_eval_launch,
               lib.mt_optim_launch, lib.mt_mlp_train_steps, lib.mt_mlp_eval):
        fn.restype = C.c_int
    from ..options import OPTIONS
    lib.mt_attn_exact_level(int(OPTIONS.attn_exact))
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise KernelLibraryError(f"{what} failed with code {code}")


def current_stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def raw_stream(device) -> int:
    """Handle of the current HIP stream of ``device`` (a ``torch.device`` or a tensor's ``.device``) without building a
    ``torch.cuda.Stream`` object: this sits on the launch path of every kernel binding."""
    import torch
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(idx if idx is not None else torch.cuda.current_device())
