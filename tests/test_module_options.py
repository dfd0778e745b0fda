"""CPU: plumbing of the module options no shipped experiment config sets (the kernels behind them are tested on the GPU against the
reference modules' own outputs: tests/test_module_options_gpu.py)."""
import ctypes

import pytest


def test_constructors_accept_what_the_reference_accepts_and_refuse_with_its_wording():
    from geotransformer_amd.modules.geotransformer import GeometricStructureEmbedding, LocalGlobalRegistration
    assert GeometricStructureEmbedding(64, 0.2, 15, 3, reduction_a='mean').reduction_a == 'mean'
    with pytest.raises(ValueError, match='Unsupported reduction mode: sum.'):  # geotransformer.py:22-23
        GeometricStructureEmbedding(64, 0.2, 15, 3, reduction_a='sum')
    head = LocalGlobalRegistration(3, 0.1, use_global_score=True, correspondence_limit=500)
    assert head.use_global_score and head.correspondence_limit == 500
    with pytest.raises(NotImplementedError, match='local_global_registration.py:78'):
        LocalGlobalRegistration(3, 0.1, use_dustbin=True)


def test_a_model_with_module_level_options_runs_module_by_module():
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.model import create_model
    assert create_model(make_cfg('modelnet')).use_native
    assert not create_model(make_cfg('modelnet', {'fine_matching.correspondence_limit': 100})).use_native
    assert not create_model(make_cfg('modelnet', {'fine_matching.use_global_score': True})).use_native
    m = create_model(make_cfg('modelnet', {'geotransformer.reduction_a': 'mean'}))
    assert m.use_native and m.transformer.embedding.reduction_a == 'mean'  # reduction_a travels in the native descriptor


def test_reduction_a_sits_where_the_header_puts_it():
    """geotr_transformer.reduction_a took the place of a padding word (ABI 6): the ctypes mirror must agree on the offset."""
    import re
    import os
    from geotransformer_amd import native
    t = native.Transformer if hasattr(native, 'Transformer') else next(c for c in vars(native).values() if isinstance(c, type) and
                                                                      issubclass(c, ctypes.Structure) and any(f[0] == 'reduction_a' for f in c._fields_))
    assert t.reduction_a.offset == 12 and t.sigma_d.offset == 16
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'geotr.h')).read()
    body = re.search(r'typedef struct geotr_transformer \{(.*?)\} geotr_transformer;', header, flags=re.S).group(1)
    assert re.search(r'int32_t num_layers, num_heads, angle_k;\s*int32_t reduction_a;', body)


def test_attention_signatures_mirror_the_reference():
    import inspect
    from geotransformer_amd.modules.transformer.rpe_transformer import RPEMultiHeadAttention
    from geotransformer_amd.modules.transformer.vanilla_transformer import MultiHeadAttention
    assert list(inspect.signature(RPEMultiHeadAttention.forward).parameters)[1:] == [
        'input_q', 'input_k', 'input_v', 'embed_qk', 'key_weights', 'key_masks', 'attention_factors']  # rpe_transformer.py:35
    assert list(inspect.signature(MultiHeadAttention.forward).parameters)[1:] == [
        'input_q', 'input_k', 'input_v', 'key_weights', 'key_masks', 'attention_factors', 'attention_masks']  # vanilla_transformer.py:36-38


def test_new_entry_points_validate_their_arguments_before_touching_the_device():
    """ABI 6 entry points: argument checks run on the host before any launch (no GPU needed), with the error text in geotr_last_error."""
    from geotransformer_amd import _lib
    lib = _lib.load()

    def err():
        lib.geotr_last_error.restype = ctypes.c_char_p
        return lib.geotr_last_error().decode()

    # the verification set of correspondence_limit needs room in the workspace, bounded by the number of correspondences that can exist
    base = lib.geotr_lgr_workspace_bytes(256, 64, 3)
    assert lib.geotr_lgr_ex_workspace_bytes(256, 64, 3, 0) == base
    assert base < lib.geotr_lgr_ex_workspace_bytes(256, 64, 3, 1000) < lib.geotr_lgr_ex_workspace_bytes(256, 64, 3, 10 ** 9)
    assert lib.geotr_lgr_ex_workspace_bytes(256, 64, 3, 10 ** 9) == lib.geotr_lgr_ex_workspace_bytes(256, 64, 3, 256 * 64 * 3)
    null = ctypes.c_void_p(0)
    rc = lib.geotr_gse_embed_table_ex(null, null, null, 9, 256, null, 2, null, 2, null, null, null, null, null, 0.2, 15.0, 0, null, null, null, null, null)
    assert rc != 0 and 'angle_k' in err()
    rc = lib.geotr_gse_embed_table_ex(null, null, null, 3, 100, null, 2, null, 2, null, null, null, null, null, 0.2, 15.0, 0, null, null, null, null, null)
    assert rc != 0 and 'hidden_dim' in err()
    rc = lib.geotr_attn_softmax_grouped_pos(null, null, null, null, 4, 0.125, null)
    assert rc != 0 and 'attn_softmax_grouped_pos' in err()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.geotr_attn_softmax_ex(p, 2, null, null, null, 2, 4, 0, 2, 0.5, p, null, null, 0, null, 0, null)  # ld < m
    assert rc != 0 and 'bad sizes' in err()
    rc = lib.geotr_attn_softmax_ex(p, 4, null, null, null, 2, 4, 0, 2, 0.5, null, null, p, 2, null, 0, null)  # factors' leading dimension < m
    assert rc != 0 and 'leading dimensions' in err()
