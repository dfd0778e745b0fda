"""CPU: the C-ABI library loads and exports every symbol include/geotr.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'geotr.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(geotr_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_something():
    names = _declared()
    assert 'geotr_radius_query' in names and 'geotr_grid_subsample' in names


def test_library_exports_every_declared_symbol():
    from geotransformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f'{name} declared in include/geotr.h but not exported'


def test_python_binding_covers_header():
    from geotransformer_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.geotr_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call, with a message from geotr_last_error()."""
    from geotransformer_amd import _lib
    lib = _lib.load()
    code = lib.geotr_radius_grid_build(None, None, 1, 0, 0.1, None, 0, None)
    assert code == -1
    assert b'null pointer' in lib.geotr_last_error()
    assert lib.geotr_radius_grid_workspace_bytes(1000, 2) > 0
    assert lib.geotr_grid_subsample_workspace_bytes(1000, 2) > 0
    # the entry points added in round 2 validate before they launch, too
    one = ctypes.c_void_p(16)  # a non-null address that is never dereferenced: every call below is rejected on its arguments
    assert lib.geotr_radius_grid_order(None, 10, 1, None, None) == -1 and b'radius_grid_order' in lib.geotr_last_error()
    assert lib.geotr_radius_grid_order(one, 10, 1000, one, None) == -1
    assert lib.geotr_kpconv_c1_fused(one, one, one, one, one, 8, 8, 65, 64, 15, 0.1, one, None, None, one, None) == -1
    assert b'h <= 64' in lib.geotr_last_error()
    assert lib.geotr_kpconv_c1_fused(one, one, one, one, one, 8, 8, 38, 64, 14, 0.1, one, None, None, one, None) == -1
    assert b'15 kernel points' in lib.geotr_last_error()
    assert lib.geotr_kpconv_fused_supported(32, 64, 38) == 1 and lib.geotr_kpconv_fused_supported(64, 256, 38) == 1 and lib.geotr_kpconv_fused_supported(128, 128, 38) == 0 and lib.geotr_kpconv_fused_supported(96, 128, 38) == 0 and lib.geotr_kpconv_fused_supported(64, 512, 38) == 0
    assert lib.geotr_kpconv_fused_supported(64, 96, 38) == 0 and lib.geotr_kpconv_fused_supported(64, 64, 41) == 0
    assert lib.geotr_kpconv_fused(one, one, one, one, one, one, 8, 8, 38, 96, 128, 15, 0.1, one, None, 0, None, one, None) == -1
    assert b'unsupported shape' in lib.geotr_last_error()
    assert lib.geotr_maxpool_ordered(None, None, 4, 4, 3, 8, None, None, None) == -1 and b'maxpool' in lib.geotr_last_error()
    segs = (ctypes.c_int64 * 2)(5, 4)
    assert lib.geotr_group_norm_shortcut(one, None, 9, 8, 2, one, one, 1e-5, 2, one, one, 1e-5, 0, one, segs, 2, one, None) == -1
    assert lib.geotr_group_norm_shortcut(one, one, 9, 8, 3, one, one, 1e-5, 2, one, one, 1e-5, 0, one, segs, 2, one, None) == -1
    assert b'channels' in lib.geotr_last_error()  # 8 channels / 3 groups
    assert lib.geotr_group_norm_shortcut(one, one, 10, 8, 2, one, one, 1e-5, 2, one, one, 1e-5, 0, one, segs, 2, one, None) == -1
    assert b'segments cover' in lib.geotr_last_error()
    assert lib.geotr_group_norm_flags_supported(128) == 1 and lib.geotr_group_norm_flags_supported(6) == 0
    assert lib.geotr_profile_stride(0) == -1 and lib.geotr_profile_stride(8) == 0 and lib.geotr_profile_stride(1) == 0
    assert lib.geotr_group_norm_workspace_bytes(1000, 64) >= 4 * 2 * 64 * (1000 // 32 + 2 * 16)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under geotransformer_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'geotransformer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'libneighbors_oracle' not in src and 'libgeoref' not in src, f


def test_ext_rejects_bad_inputs_like_the_reference():
    import torch
    from geotransformer_amd import ext
    pts = torch.zeros(4, 3)
    lens = torch.tensor([4])
    with pytest.raises(RuntimeError, match='must be a float tensor'):
        ext.radius_neighbors(pts.double(), pts, lens, lens, 0.1)
    with pytest.raises(RuntimeError, match='must be an long tensor'):
        ext.radius_neighbors(pts, pts, lens.int(), lens, 0.1)
    with pytest.raises(RuntimeError, match='must be contiguous'):
        ext.grid_subsampling(torch.zeros(3, 4).t(), lens, 0.1)
    if not torch.cuda.is_available():  # the product path must fail loudly, never fall back to CPU
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            ext.radius_neighbors(pts, pts, lens, lens, 0.1)


def test_integration_doc_indexes_every_entry_point():
    """INTEGRATION.md section 4 is the audit trail of the boundary: every symbol the header declares must appear there, next to the
    reference code it replaces and the place it is bound."""
    import re
    header = open(os.path.join(ROOT, 'include', 'geotr.h')).read()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    symbols = sorted(set(re.findall(r'\b(geotr_[a-z0-9_]+)\s*\(', header)))
    assert len(symbols) >= 49
    missing = [s for s in symbols if '`' + s + '`' not in doc]
    assert not missing, missing


def test_set_precision_is_host_logic_and_invalidates_model_descriptors():
    """The precision switch is plain module state (no GPU needed): modes round-trip, unknown names are rejected, and the native
    model descriptor's cache key includes the mode, so a model that already ran re-derives its packed weights after a switch."""
    import torch
    from geotransformer_amd import kernels
    from geotransformer_amd.native import NativeModel
    assert kernels.DEFAULT_PRECISION == 'fp32'
    assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('fp32', 5)  # the default: exact fp32 products (the reference's arithmetic) + the embedding by table (fp32)
    model = NativeModel(torch.nn.Linear(4, 4))
    default_key = model._version_key()
    try:
        assert kernels.set_precision('bf16') == 'fp32'
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('bf16', 5)
        assert model._version_key() != default_key
        assert kernels.set_precision('bf16x3') == 'bf16' and kernels.gemm_mode() == 0 and model._version_key() != default_key
        kernels.set_precision('bf16', gse='mfma')
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('bf16', 3)
        kernels.set_precision('bf16x3', gse='mfma')
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == (True, 1)
        kernels.set_precision('bf16')
        assert kernels.set_precision('fp32') == 'bf16'
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('fp32', 5)  # exact fp32 products on the packed pipeline + the fp32 table embedding
        assert kernels.gemm_mode() == 2
        kernels.set_precision('fp32', gse='mfma')
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('fp32', 0)
        assert kernels.set_precision('fp32-unpacked') == 'fp32'
        assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == (False, 5) and kernels.gemm_mode() == 0
        with pytest.raises(ValueError):
            kernels.set_precision('fp8')
        with pytest.raises(ValueError):
            kernels.set_precision('fp32', gse='lut')
    finally:
        kernels.set_precision(kernels.DEFAULT_PRECISION)
    assert model._version_key() == default_key


def test_documents_reference_existing_files():
    """DESIGN / README / INTEGRATION cite evidence by path (profiles, scripts, tests, sources): none of those may dangle."""
    import re
    missing = []
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md'):
        text = open(os.path.join(ROOT, doc)).read()
        pattern = r'`((?:profiles|scripts|tests|oracle|include|geotransformer_amd)/[\w./\-]+?\.(?:md|json|txt|csv|py|sh|h|hip|npz|cpp))`'
        cited = [m.group(1) for m in re.finditer(pattern, text)]
        cited += ['profiles/' + m.group(1) for m in re.finditer(r'`(r0\d_[\w.\-]+\.(?:md|json|txt|csv))`', text)]
        assert cited or doc == 'README.md'
        missing += [(doc, path) for path in cited if not os.path.exists(os.path.join(ROOT, path))]
    assert not missing, missing


def test_native_descriptor_builds_without_a_gpu(monkeypatch):
    """NativeModel._build walks the module tree into the C descriptor (native.py): run it on CPU tensors with the three GPU-only
    helpers stubbed, for every experiment -- a Python-level slip in it (round 3 had one) must not need a GPU session to surface."""
    import torch
    from geotransformer_amd import kernels, native
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.model import create_model
    monkeypatch.setattr(kernels, 'gemm_pack', lambda w, **kw: torch.zeros(16, dtype=torch.uint8))
    monkeypatch.setattr(kernels, 'decoder_packs', lambda w, c: (torch.zeros(16, dtype=torch.uint8), torch.zeros(16, dtype=torch.uint8)))
    small = {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 32, 'geotransformer.hidden_dim': 32,
             'geotransformer.output_dim': 32}
    for exp, in_dim, stages in (('3dmatch', 256, 4), ('kitti', 512, 5), ('modelnet', 128, 3)):
        model = create_model(make_cfg(exp, dict(small, **{'geotransformer.input_dim': in_dim})))
        monkeypatch.setattr(model.transformer.embedding, 'tables', lambda: (torch.zeros(4, 4, 32), torch.zeros(4, 4, 32)), raising=False)
        desc, keep = native.NativeModel(model)._build()
        bb = desc.backbone
        assert bb.num_stages == stages and bb.num_blocks == 2 + 3 * (stages - 1) and bb.num_decoders == stages - 1 - model.backbone.fine_stage
        for d in range(bb.num_decoders):
            assert bb.decoder_packed_latent[d] and bb.decoder_packed_skip[d] and bb.decoder[d].packed
        assert desc.transformer.num_layers == 6 and desc.num_points_in_patch == model.num_points_in_patch
        # the split point of every decoder weight: latent channels + skip channels = the Linear's input width
        net = model.backbone
        for i in range(stages - 2, net.fine_stage - 1, -1):
            dec = getattr(net, f'decoder{i + 1}')
            skip = getattr(net, f'encoder{i + 1}_{3 if i > 0 else 2}').out_channels
            assert net.decoder_latent_channels(i) + skip == dec.mlp.weight.shape[1], (exp, i)


def test_header_macro_library_and_python_loader_agree_on_the_abi_version():
    from geotransformer_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'geotr.h')).read()
    macro = int(re.search(r'^#define GEOTR_ABI_VERSION (\d+)$', text, flags=re.M).group(1))
    lib = ctypes.CDLL(os.path.join(ROOT, 'geotransformer_amd', 'libgeotr_hip.so'))
    lib.geotr_abi_version.restype = ctypes.c_int
    assert macro == lib.geotr_abi_version() == _lib.ABI_VERSION


def test_a_cpp_host_builds_against_the_header_and_the_library(tmp_path):
    """scripts/abi_bench.cpp (INTEGRATION.md "A host without Python"): compiles with the header as a C++ translation unit, links against the
    library, passes its ABI check and reaches its usage line -- no device call is made."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'abi_bench')
    lib_dir = os.path.join(ROOT, 'geotransformer_amd')
    build = subprocess.run([hipcc, '-O1', '-std=c++17', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'scripts', 'abi_bench.cpp'), '-L', lib_dir,
                            '-lgeotr_hip', '-Wl,-rpath,' + lib_dir, '-o', exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, 'help'], capture_output=True, text=True)
    assert run.returncode == 64 and 'usage:' in run.stderr, (run.returncode, run.stderr[-500:])
