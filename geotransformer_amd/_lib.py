"""ctypes binding of libgeotr_hip.so (the C ABI declared in include/geotr.h).

The product path has NO fallback: if the HIP library is missing, or a call fails, a RuntimeError is
raised.  Nothing here imports or calls anything under oracle/.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgeotr_hip.so')

c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t
c_int = ctypes.c_int

ABI_VERSION = 8  # geotr_abi_version() of the library this package's ctypes mirrors (native.py, kernels.py) were written against

# name -> (restype, argtypes); must list every symbol include/geotr.h declares (tests check this)
SIGNATURES = {
    'geotr_last_error': (ctypes.c_char_p, []),
    'geotr_abi_version': (ctypes.c_int, []),
    'geotr_grid_subsample_workspace_bytes': (c_size, [c_i64, c_i64]),
    'geotr_grid_subsample': (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i64, c_f32, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_radius_grid_workspace_bytes': (c_size, [c_i64, c_i64]),
    'geotr_radius_grid_build': (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_i64, c_f32, c_ptr, c_size, c_ptr]),
    'geotr_radius_grid_order': (ctypes.c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_radius_count': (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_f32, c_ptr, c_ptr, c_ptr]),
    'geotr_radius_query': (ctypes.c_int,
                           [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_f32, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'geotr_gemm': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                           c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_int, c_ptr]),
    'geotr_row_positive': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_kpconv_gather': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32,
                                    c_ptr, c_ptr, c_ptr]),
    'geotr_kpconv_fused_supported': (c_int, [c_i64, c_i64, c_i64]),
    'geotr_kpconv_c1_fused': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_kpconv_fused': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_ptr, c_ptr,
                                   c_int, c_ptr, c_ptr, c_ptr]),
    'geotr_maxpool': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_maxpool_ordered': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'geotr_upsample_concat': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_group_norm_workspace_bytes': (c_size, [c_i64, c_i64]),
    'geotr_group_norm': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    'geotr_layer_norm': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_ptr]),
    'geotr_gse_knn': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_gse_embed_workspace_bytes': (c_size, [c_i64, c_int]),
    'geotr_gse_embed': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_int, c_ptr,
                                c_size, c_ptr, c_ptr]),
    'geotr_attn_softmax': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_f32, c_ptr]),
    'geotr_point_to_node': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_superpoint_match_workspace_bytes': (c_size, [c_i64, c_i64]),
    'geotr_superpoint_match': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_int, c_i64, c_ptr, c_size, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_patch_sinkhorn': (c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr,
                                     c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_patch_gather': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_ptr,
                                   c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_node_correspondences_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'geotr_node_correspondences': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64,
                                           c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_registration_metrics': (c_int, [c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_ptr,
                                           c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    'geotr_attn_softmax_grouped': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_f32, c_ptr]),
    'geotr_kdtree_workspace_bytes': (c_size, [c_i64, c_i64]),
    'geotr_kdtree_build': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_size, c_ptr]),
    'geotr_kdtree_search_scratch_bytes': (c_size, [c_i64, c_i64]),
    'geotr_kdtree_radius_search': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_f32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                                           c_ptr, c_size, c_ptr]),
    'geotr_gemm_grouped': (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_i64, c_f32, c_ptr]),
    'geotr_gemm_pack_bytes': (c_size, [c_i64, c_i64]),
    'geotr_gemm_pack': (c_int, [c_ptr, c_i64, c_int, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_gemm_packed': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_int, c_ptr]),
    'geotr_gemm_packed_bf16': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_int, c_ptr]),
    'geotr_gemm_pack_f32': (c_int, [c_ptr, c_i64, c_int, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_gemm_pack_format': (c_int, [c_ptr]),
    'geotr_gemm_pack_forget': (None, [c_ptr]),
    'geotr_gemm_packed_f32': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_int, c_ptr]),
    'geotr_gemm_packed_splitk_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'geotr_gemm_packed_splitk_workspace_bytes_mode': (c_size, [c_i64, c_i64, c_i64, c_int]),
    'geotr_gemm_packed_splits': (c_int, [c_i64, c_i64, c_i64, c_int]),
    'geotr_gemm_packed_tile_width': (c_int, [c_i64, c_i64, c_i64, c_int, c_int]),
    'geotr_gemm_packed_splitk': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_int, c_int,
                                         c_ptr, c_size, c_ptr]),
    'geotr_gemm_packed_stats_rows_per_record': (c_i64, [c_i64]),
    'geotr_gemm_packed_stats_floats': (c_size, [c_ptr, c_i64, c_i64]),
    'geotr_gemm_packed_stats': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_int, c_int, c_ptr, c_i64, c_ptr,
                                        c_ptr]),
    'geotr_gemm_packed_tail': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                                       c_i64, c_ptr]),
    'geotr_group_norm_finalize': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr]),
    'geotr_gemm_packed_gather': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_int, c_int, c_ptr, c_i64, c_i64, c_ptr, c_i64,
                                         c_ptr, c_i64, c_ptr, c_ptr]),
    'geotr_group_norm_stats': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr,
                                       c_f32, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    'geotr_group_norm_segmented': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'geotr_group_norm_flags_supported': (c_int, [c_i64]),
    'geotr_group_norm_shortcut': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_i64, c_ptr, c_ptr, c_f32, c_int, c_ptr, c_ptr,
                                          c_i64, c_ptr, c_ptr]),
    'geotr_group_norm_segmented_flags': (c_int, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_f32, c_ptr, c_int, c_ptr, c_ptr, c_i64, c_ptr,
                                                 c_ptr, c_ptr]),
    'geotr_l2_normalize': (c_int, [c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_weighted_procrustes': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    'geotr_lgr_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'geotr_model_workspace_bytes': (c_size, [c_ptr, c_ptr]),   # struct pointers; typed in geotransformer_amd/native.py
    'geotr_model_forward': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_pyramid_workspace_bytes': (c_size, [c_i64, c_i64, c_i64]),
    'geotr_pyramid_build': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_pyramid_build_async': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_profile_gse': (c_int, [c_ptr, c_ptr, c_ptr, c_i64]),
    'geotr_profile_gse_count': (c_i64, []),
    'geotr_profile_stride': (c_int, [c_i64]),
    'geotr_gse_table_bytes': (c_size, [c_i64, c_i64]),
    'geotr_gse_table_build': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_gse_knn_clouds': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'geotr_gse_embed_table': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                      c_f32, c_f32, c_ptr, c_ptr]),
    'geotr_gse_embed_table_ex': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                         c_f32, c_f32, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'geotr_attn_softmax_grouped_pos': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr]),
    'geotr_attn_softmax_ex': (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_f32, c_ptr, c_ptr, c_ptr, c_i64,
                                      c_ptr, c_i64, c_ptr]),
    'geotr_lgr_ex_workspace_bytes': (c_size, [c_i64, c_i64, c_i64, c_i64]),
    'geotr_lgr_ex': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_int, c_f32, c_i64,
                             c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'geotr_stack_clouds': (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    'geotr_apply_transform': (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'geotr_pairwise_distance': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    'geotr_index_select': (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    'geotr_lgr': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_f32, c_int, c_f32, c_i64,
                          c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
}

_lib = None


def load():
    """Load the HIP library or fail loudly (no CPU / eager fallback exists by design)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C geotransformer_amd/csrc`. geotransformer_amd has no fallback path.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.geotr_abi_version() != ABI_VERSION:  # descriptor structs are mirrored field by field (native.py): a stale build would misread them
            raise RuntimeError(f'{LIB_PATH} has ABI version {lib.geotr_abi_version()}, this package needs {ABI_VERSION}: rebuild it '
                               f'(`python -c "import __graft_entry__ as g; g.build()"`)')
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().geotr_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed (code {code}): {msg}')


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('geotransformer_amd needs a HIP device (MI355X / gfx950); no CPU fallback exists.')
