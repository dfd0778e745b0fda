"""Thin torch-facing wrappers over the C ABI (include/geotr.h).  Device tensors in, new device tensors out;
every call is asynchronous on the current torch stream.  No eager/PyTorch fallback exists: a missing library
or a failing call raises RuntimeError."""
import ctypes
import math

import os

import torch

from . import _lib

ACT = {None: 0, 'none': 0, 'relu': 1, 'leaky': 2}

# bench.py sets PROFILE['<kernel>'] = [] to collect (start_event, end_event, problem_size) per launch: HIP events recorded
# on the stream the kernel is launched on (torch's current stream is the stream passed through the C ABI).
PROFILE = {}


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_cuda, 'expected a float32 device tensor'
    return t if t.is_contiguous() else t.contiguous()


def gemm(a, b, b_is_kn=False, bias=None, row_div=None, residual=None, alpha=1.0, act=None, out=None):
    """out = act(alpha * a @ op(b) / row_div + bias + residual).

    a: (M,K) or (B,M,K); b: (N,K) [b_is_kn=False, nn.Linear weight] or (K,N) [b_is_kn=True], optionally batched.
    Rows may be strided views (last dim contiguous)."""
    lib = _lib.load()
    batched = a.dim() == 3
    if not batched:
        a3, b3 = a.unsqueeze(0), b.unsqueeze(0)
    else:
        a3, b3 = a, (b if b.dim() == 3 else b.unsqueeze(0).expand(a.shape[0], -1, -1))
    assert a3.stride(-1) == 1 and b3.stride(-1) == 1
    B, M, K = a3.shape
    N = b3.shape[2] if b_is_kn else b3.shape[1]
    assert (b3.shape[1] if b_is_kn else b3.shape[2]) == K, 'inner dimensions differ'
    if out is None:
        out = torch.empty((B, M, N) if batched else (M, N), dtype=torch.float32, device=a.device)
    o3 = out if batched else out.unsqueeze(0)
    assert o3.stride(-1) == 1
    ldr = 0
    if residual is not None:
        assert residual.stride(-1) == 1 and not batched
        ldr = residual.stride(0)
    _lib.check(lib.geotr_gemm(_lib.ptr(a3), a3.stride(1), _lib.ptr(b3), b3.stride(1), int(b_is_kn), _lib.ptr(o3),
                              o3.stride(1), M, N, K, B, a3.stride(0) if B > 1 else 0, b3.stride(0) if B > 1 else 0,
                              o3.stride(0) if B > 1 else 0, _lib.ptr(bias), _lib.ptr(row_div), _lib.ptr(residual), ldr,
                              float(alpha), ACT[act], _lib.stream_ptr()), 'geotr_gemm')
    return out


DEFAULT_PRECISION = 'fp32'  # the reference's own arithmetic (round 4): what the package, the tests and bench.py run in unless told otherwise
GEMM_PACKED = 'fp32'    # arithmetic of the packed GEMM pipeline (see set_precision): True: split-bf16 "bf16x3";  'fp32': exact fp32 products
                        # (v_mfma_f32_32x32x2_f32, the reference's arithmetic) on the SAME pipeline;  'bf16': plain bf16 operands (BASELINE
                        # configs[4] "bf16 features");  False: no packed weights at all -- every contraction on the unpacked fp32 kernel
PACKED_MIN_ROWS = 1024  # activations with at least this many rows use the packed GEMM when a packed weight is given


def gemm_mode():
    """The `bf16_operands` / `gemm_mode` argument of the C ABI for the current precision: 0 split-bf16, 1 plain bf16, 2 exact fp32."""
    return {True: 0, 'bf16': 1, 'fp32': 2}.get(GEMM_PACKED, 0)



def use_packed(a):
    """Same predicate as the native executor: tall, 16-byte aligned rows, K a multiple of 32."""
    return (GEMM_PACKED and a.shape[0] >= PACKED_MIN_ROWS and a.shape[1] % 32 == 0 and a.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0)


def _forget_when_freed(packed):
    """The library records the layout of every buffer it packs by ADDRESS; drop the record when the tensor dies, so that the caching
    allocator can hand the address to an unrelated buffer (geotr_gemm_pack_forget)."""
    import weakref
    weakref.finalize(packed, _lib.load().geotr_gemm_pack_forget, ctypes.c_void_p(packed.data_ptr())).atexit = False
    return packed


def gemm_pack(weight, b_is_kn=False, view=None):
    """Packed hi/lo bf16 planes of a static weight for gemm_packed.  `weight` is the parameter itself (2-D (N,K), or (K,N) with
    b_is_kn; `view` = 2-D shape to read it as, e.g. KPConv's (15*C_in, C_out)).  The result is cached ON the tensor object
    together with its version counter, so an in-place update re-packs and a freed tensor cannot leave a stale entry behind."""
    w2 = weight.detach() if view is None else weight.detach().view(*view)
    assert w2.dim() == 2 and w2.stride(-1) == 1 and w2.dtype == torch.float32
    f32 = GEMM_PACKED == 'fp32'  # one fp32 plane in the 32x32x2 fragment order instead of the hi / lo bf16 planes (same bytes)
    key = (weight._version, weight.data_ptr(), tuple(w2.shape), w2.stride(0), bool(b_is_kn), weight.device.index, f32)
    hit = getattr(weight, '_geotr_packed', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    lib = _lib.load()
    n, k = (w2.shape[1], w2.shape[0]) if b_is_kn else (w2.shape[0], w2.shape[1])
    packed = torch.empty(lib.geotr_gemm_pack_bytes(n, k), dtype=torch.uint8, device=weight.device)
    pack = lib.geotr_gemm_pack_f32 if f32 else lib.geotr_gemm_pack
    _lib.check(pack(_lib.ptr(w2), w2.stride(0), int(b_is_kn), n, k, _lib.ptr(packed), _lib.stream_ptr()), 'geotr_gemm_pack')
    _forget_when_freed(packed)
    try:
        weight._geotr_packed = (key, packed)
    except AttributeError:  # tensors that reject attributes are simply not cached
        pass
    return packed


def gemm_packed(a, packed, n, bias=None, row_div=None, residual=None, alpha=1.0, act=None, out=None, split_k=True):
    """out (M, n) = act(alpha * a @ W^T / row_div + bias + residual) with W given by gemm_pack (split-bf16 MFMA; plain bf16
    operands when GEMM_PACKED == 'bf16').  `split_k` (default): narrow, deep launches are split over K (gridDim.z slices +
    a deterministic reduce, geotr_gemm_packed_splitk); False forces the single-pass kernel."""
    lib = _lib.load()
    assert a.dim() == 2 and a.stride(-1) == 1
    M, K = a.shape
    if out is None:
        out = torch.empty((M, n), dtype=torch.float32, device=a.device)
    ldr = 0
    if residual is not None:
        assert residual.stride(-1) == 1
        ldr = residual.stride(0)
    mode = gemm_mode()
    nbytes = lib.geotr_gemm_packed_splitk_workspace_bytes_mode(M, n, K, mode) if split_k else 0
    if nbytes:
        ws = _lib.workspace(nbytes, a.device)
        _lib.check(lib.geotr_gemm_packed_splitk(_lib.ptr(a), a.stride(0), _lib.ptr(packed), _lib.ptr(out), out.stride(0), M, n, K,
                                                _lib.ptr(bias), _lib.ptr(row_div), _lib.ptr(residual), ldr, float(alpha), ACT[act],
                                                mode, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), 'geotr_gemm_packed_splitk')
        ws.record_stream(torch.cuda.current_stream())
        return out
    fn, name = [(lib.geotr_gemm_packed, 'geotr_gemm_packed'), (lib.geotr_gemm_packed_bf16, 'geotr_gemm_packed_bf16'),
                (lib.geotr_gemm_packed_f32, 'geotr_gemm_packed_f32')][mode]
    _lib.check(fn(_lib.ptr(a), a.stride(0), _lib.ptr(packed), _lib.ptr(out), out.stride(0), M, n, K, _lib.ptr(bias),
                  _lib.ptr(row_div), _lib.ptr(residual), ldr, float(alpha), ACT[act], _lib.stream_ptr()), name)
    return out


def linear(x, weight, bias=None, act=None, residual=None, packed=False):
    """F.linear(x, weight, bias) on the matrix cores; x may have leading batch dims.  packed=True (static backbone
    weights): activations with >= PACKED_MIN_ROWS rows go through the packed split-bf16 GEMM, like the native executor."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    res2 = residual.reshape(-1, weight.shape[0]) if residual is not None else None
    if packed and use_packed(x2):
        y = gemm_packed(x2, gemm_pack(weight), weight.shape[0], bias=bias, act=act, residual=res2)
    else:
        y = gemm(x2, weight, bias=bias, act=act, residual=res2)
    return y.view(*shape[:-1], weight.shape[0])


GN_EPILOGUE_STATS = os.environ.get('GEOTR_GN_EPILOGUE_STATS', '1') != '0'  # same A/B switch as the native executor (csrc/executor.hip)


def linear_gn(x, weight, bias=None, seg_rows=None):
    """A Linear whose output feeds a GroupNorm: (y, stats, rows_per_record).  On the packed path the producing GEMM's epilogue also writes
    the GroupNorm statistics records of y (geotr_gemm_packed_stats: no statistics pass over y later); otherwise stats is None and the
    norm computes its own.  Same predicate as the native executor (linear_gn in csrc/executor.hip): unsplit packed launches only."""
    import ctypes
    lib = _lib.load()
    assert x.dim() == 2
    x2 = x if x.stride(-1) == 1 else x.contiguous()
    M, K = x2.shape
    n = weight.shape[0]
    if not (GN_EPILOGUE_STATS and use_packed(x2) and lib.geotr_gemm_packed_splits(M, n, K, 0) == 1):  # the split-bf16 rule in every mode (as the executor)
        return linear(x2, weight, bias, packed=True), None, 0
    segs = [M] if seg_rows is None else [int(r) for r in seg_rows]
    seg_arr = (ctypes.c_int64 * len(segs))(*segs)
    y = torch.empty((M, n), dtype=torch.float32, device=x2.device)
    stats = torch.empty(lib.geotr_gemm_packed_stats_floats(seg_arr, len(segs), n), dtype=torch.float32, device=x2.device)
    _lib.check(lib.geotr_gemm_packed_stats(_lib.ptr(x2), x2.stride(0), _lib.ptr(gemm_pack(weight)), _lib.ptr(y), y.stride(0), M, n, K,
                                           _lib.ptr(bias), None, 0, gemm_mode(), seg_arr, len(segs), _lib.ptr(stats),
                                           _lib.stream_ptr()), 'geotr_gemm_packed_stats')
    return y, stats, int(lib.geotr_gemm_packed_stats_rows_per_record(n))


def group_norm_stats(x, groups, weight, bias, eps=1e-5, x_stats=None, x_rpr=0, residual=None, res_stats=None, res_rpr=0, res_norm=None,
                     act=None, seg_rows=None):
    """act(GN(x) + R) with the statistics of x (and of a residual carrying its own norm, res_norm = (groups, weight, bias, eps)) taken from
    the records their producing GEMMs wrote (linear_gn) where given (geotr_group_norm_stats)."""
    import ctypes
    lib = _lib.load()
    x = _f32c(x)
    N, C = x.shape
    segs = [N] if seg_rows is None else [int(r) for r in seg_rows]
    out = torch.empty_like(x)
    ws = _lib.workspace(lib.geotr_group_norm_workspace_bytes(N, C), x.device)
    if residual is not None:
        residual = _f32c(residual)
    rg, rw, rb, re = res_norm if res_norm is not None else (0, None, None, 0.0)
    _lib.check(lib.geotr_group_norm_stats(_lib.ptr(x), N, C, groups, _lib.ptr(weight), _lib.ptr(bias), float(eps), _lib.ptr(x_stats), int(x_rpr),
                                          _lib.ptr(residual), _lib.ptr(res_stats), int(res_rpr), int(rg), _lib.ptr(rw), _lib.ptr(rb), float(re),
                                          ACT[act], _lib.ptr(out), (ctypes.c_int64 * len(segs))(*segs), len(segs), _lib.ptr(ws), None,
                                          _lib.stream_ptr()), 'geotr_group_norm_stats')
    return out


def residual_tail(y, weight, bias, norm, shortcut, sc_weight=None, sc_bias=None, sc_norm=None, seg_rows=None):
    """leaky(GN(y W^T + b) + S) with S = shortcut, or GN'(shortcut W_s^T + b_s) when the shortcut has its own Linear + norm -- the tail of
    a ResidualBlock (kpconv/modules.py:204-224) WITHOUT an apply pass: every product is launched once for its statistics only and once
    more with its GroupNorm applied in the epilogue (geotr_gemm_packed_tail, geotr_group_norm_finalize), as the native executor does.
    norm / sc_norm = (groups, gamma, beta, eps).  Bit-identical to linear_gn + group_norm_stats.  Packed-path shapes only."""
    import ctypes
    lib = _lib.load()
    y, shortcut = _f32c(y), _f32c(shortcut)
    M, K = y.shape
    C = weight.shape[0]
    assert use_packed(y) and lib.geotr_gemm_packed_splits(M, C, K, 0) == 1
    segs = [M] if seg_rows is None else [int(r) for r in seg_rows]
    seg_arr = (ctypes.c_int64 * len(segs))(*segs)
    bf16 = gemm_mode()
    rec = torch.empty(lib.geotr_gemm_packed_stats_floats(seg_arr, len(segs), C), dtype=torch.float32, device=y.device)
    rpr = int(lib.geotr_gemm_packed_stats_rows_per_record(C))
    out = torch.empty((M, C), dtype=torch.float32, device=y.device)

    def stats_of(a, w, b, nm):
        _lib.check(lib.geotr_gemm_packed_tail(_lib.ptr(a), a.stride(0), _lib.ptr(gemm_pack(w)), None, C, M, C, a.shape[1], _lib.ptr(b), 0, bf16,
                                              seg_arr, len(segs), _lib.ptr(rec), None, None, 0, _lib.stream_ptr()), 'geotr_gemm_packed_tail')
        ab = torch.empty((len(segs), 2, C), dtype=torch.float32, device=y.device)
        _lib.check(lib.geotr_group_norm_finalize(_lib.ptr(rec), rpr, M, C, int(nm[0]), _lib.ptr(nm[1]), _lib.ptr(nm[2]), float(nm[3]), seg_arr,
                                                 len(segs), _lib.ptr(ab), _lib.stream_ptr()), 'geotr_group_norm_finalize')
        return ab

    def apply(a, w, b, ab, dst, residual, act):
        _lib.check(lib.geotr_gemm_packed_tail(_lib.ptr(a), a.stride(0), _lib.ptr(gemm_pack(w)), _lib.ptr(dst), C, M, C, a.shape[1], _lib.ptr(b),
                                              ACT[act], bf16, seg_arr, len(segs), None, _lib.ptr(ab), _lib.ptr(residual),
                                              residual.stride(0) if residual is not None else 0, _lib.stream_ptr()), 'geotr_gemm_packed_tail')

    ab_z = stats_of(y, weight, bias, norm)
    if sc_weight is None:
        apply(y, weight, bias, ab_z, out, shortcut, 'leaky')
    else:
        assert use_packed(shortcut) and lib.geotr_gemm_packed_splits(M, C, shortcut.shape[1], 0) == 1
        ab_t = stats_of(shortcut, sc_weight, sc_bias, sc_norm)
        part = torch.empty_like(out)
        apply(y, weight, bias, ab_z, part, None, None)
        apply(shortcut, sc_weight, sc_bias, ab_t, out, part, 'leaky')
    return out


DECODER_SPLIT = os.environ.get('GEOTR_DECODER_SPLIT', '1') != '0'  # same A/B switch as the native executor


def decoder_packs(weight, latent_ch):
    """The decoder weight W (out, latent_ch + skip_ch) packed in its two column slices [W_latent | W_skip] (gemm_pack of strided views)."""
    w = weight.detach()
    f32 = GEMM_PACKED == 'fp32'
    key = (weight._version, weight.data_ptr(), int(latent_ch), f32)
    hit = getattr(weight, '_geotr_split_packed', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    lib = _lib.load()
    out = []
    for view in (w[:, :latent_ch], w[:, latent_ch:]):
        n, k = view.shape
        packed = torch.empty(lib.geotr_gemm_pack_bytes(n, k), dtype=torch.uint8, device=w.device)
        pack = lib.geotr_gemm_pack_f32 if f32 else lib.geotr_gemm_pack
        _lib.check(pack(view.data_ptr(), view.stride(0), 0, n, k, _lib.ptr(packed), _lib.stream_ptr()), 'geotr_gemm_pack')
        out.append(_forget_when_freed(packed))
    try:
        weight._geotr_split_packed = (key, tuple(out))
    except AttributeError:
        pass
    return tuple(out)


def decoder_linear(latent, upsample_indices, skip, weight, bias=None, want_stats=False, seg_rows=None):
    """Linear(cat(nearest_upsample(latent), skip)) of a KPConv-FPN decoder (experiments/*/backbone.py:71-78) without the concatenation:
    up(latent W_latent^T) + skip W_skip^T + b -- the coarse-level product is gathered into the fine-level GEMM's epilogue
    (geotr_gemm_packed_gather).  Returns (y, stats, rows_per_record) like linear_gn (stats only with want_stats), or None when the
    shapes are not on the packed path (the caller then concatenates, as the native executor does under the same predicate)."""
    import ctypes
    lib = _lib.load()
    latent, skip = _f32c(latent), _f32c(skip)
    lat_ch, skip_ch = latent.shape[1], skip.shape[1]
    if not (DECODER_SPLIT and GEMM_PACKED and weight.shape[1] == lat_ch + skip_ch and use_packed(latent) and use_packed(skip)):
        return None
    p_lat, p_skip = decoder_packs(weight, lat_ch)
    n_out = weight.shape[0]
    coarse = gemm_packed(latent, p_lat, n_out)
    up = upsample_indices if upsample_indices.is_contiguous() else upsample_indices.contiguous()
    M = skip.shape[0]
    assert up.dtype == torch.int64 and up.shape[0] == M
    y = torch.empty((M, n_out), dtype=torch.float32, device=skip.device)
    segs = [M] if seg_rows is None else [int(r) for r in seg_rows]
    seg_arr = (ctypes.c_int64 * len(segs))(*segs)
    stats, rpr = None, 0
    if want_stats and GN_EPILOGUE_STATS:
        stats = torch.empty(lib.geotr_gemm_packed_stats_floats(seg_arr, len(segs), n_out), dtype=torch.float32, device=skip.device)
        rpr = int(lib.geotr_gemm_packed_stats_rows_per_record(n_out))
    _lib.check(lib.geotr_gemm_packed_gather(_lib.ptr(skip), skip.stride(0), _lib.ptr(p_skip), _lib.ptr(y), y.stride(0), M, n_out, skip_ch,
                                            _lib.ptr(bias), 0, gemm_mode(), _lib.ptr(coarse), coarse.stride(0), coarse.shape[0],
                                            _lib.ptr(up), up.stride(0), seg_arr, len(segs), _lib.ptr(stats), _lib.stream_ptr()),
               'geotr_gemm_packed_gather')
    return y, stats, rpr


def row_positive(feats):
    lib = _lib.load()
    feats = _f32c(feats)
    flag = torch.empty(feats.shape[0], dtype=torch.uint8, device=feats.device)
    _lib.check(lib.geotr_row_positive(_lib.ptr(feats), feats.shape[0], feats.shape[1], _lib.ptr(flag), _lib.stream_ptr()),
               'geotr_row_positive')
    return flag


KPCONV_FUSED = True  # layers whose shape geotr_kpconv_fused supports run as one kernel (no (M, 15 C) operand in HBM)


def kpconv_fused_supported(c_in, c_out, h):
    return bool(KPCONV_FUSED and GEMM_PACKED and _lib.load().geotr_kpconv_fused_supported(int(c_in), int(c_out), int(h)))


def _order(order, m):
    """Visiting order of m query rows for the gather kernels: an int32 permutation of 0 .. m-1 (None = row order)."""
    if order is None:
        return None
    assert order.dtype == torch.int32 and order.is_cuda and order.is_contiguous() and order.numel() == m
    return order


def kpconv_fused(s_feats, q_points, s_points, neighbor_indices, kernel_points, sigma, packed, c_out, bias=None, order=None):
    """Whole KPConv layer in one kernel (csrc/kpconv_fused.hip): -> (M, c_out).  `packed` = gemm_pack of the (15 C_in, c_out) weights.
    `order`: visiting order of the query rows (ext.RadiusGrid.order(); results do not depend on it)."""
    lib = _lib.load()
    s_feats, q_points, s_points = _f32c(s_feats), _f32c(q_points), _f32c(s_points)
    nb = neighbor_indices if neighbor_indices.is_contiguous() else neighbor_indices.contiguous()
    assert nb.dtype == torch.int64
    M, H = nb.shape
    Ns, C = s_feats.shape
    flag = row_positive(s_feats)
    out = torch.empty((M, c_out), dtype=torch.float32, device=s_feats.device)
    _lib.check(lib.geotr_kpconv_fused(_lib.ptr(s_feats), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(nb), _lib.ptr(_f32c(kernel_points)),
                                      _lib.ptr(flag), M, Ns, H, C, int(c_out), kernel_points.shape[0], float(sigma), _lib.ptr(packed),
                                      _lib.ptr(bias), gemm_mode(), _lib.ptr(_order(order, M)), _lib.ptr(out), _lib.stream_ptr()),
               'geotr_kpconv_fused')
    return out


def kpconv_c1_fused(s_feats, q_points, s_points, neighbor_indices, kernel_points, sigma, weights, bias=None, order=None):
    """First layer (C_in = 1) in one exact-fp32 kernel: weights (15, 1, c_out) -> (M, c_out)."""
    lib = _lib.load()
    s_feats, q_points, s_points, weights = _f32c(s_feats), _f32c(q_points), _f32c(s_points), _f32c(weights.detach())
    nb = neighbor_indices if neighbor_indices.is_contiguous() else neighbor_indices.contiguous()
    assert nb.dtype == torch.int64 and s_feats.shape[1] == 1
    M, H = nb.shape
    c_out = weights.shape[-1]
    out = torch.empty((M, c_out), dtype=torch.float32, device=s_feats.device)
    _lib.check(lib.geotr_kpconv_c1_fused(_lib.ptr(s_feats), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(nb), _lib.ptr(_f32c(kernel_points)),
                                         M, s_feats.shape[0], H, c_out, kernel_points.shape[0], float(sigma), _lib.ptr(weights), _lib.ptr(bias),
                                         _lib.ptr(_order(order, M)), _lib.ptr(out), _lib.stream_ptr()), 'geotr_kpconv_c1_fused')
    return out


def kpconv_gather(s_feats, q_points, s_points, neighbor_indices, kernel_points, sigma):
    """-> weighted (M, 15*C) fp32, nnum (M,) int32."""
    lib = _lib.load()
    s_feats, q_points, s_points = _f32c(s_feats), _f32c(q_points), _f32c(s_points)
    nb = neighbor_indices if neighbor_indices.is_contiguous() else neighbor_indices.contiguous()
    assert nb.dtype == torch.int64
    M, H = nb.shape
    Ns, C = s_feats.shape
    K = kernel_points.shape[0]
    flag = row_positive(s_feats) if C > 1 else None
    weighted = torch.empty((M, K * C), dtype=torch.float32, device=s_feats.device)
    nnum = torch.empty(M, dtype=torch.int32, device=s_feats.device)
    _lib.check(lib.geotr_kpconv_gather(_lib.ptr(s_feats), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(nb),
                                       _lib.ptr(_f32c(kernel_points)), _lib.ptr(flag), M, Ns, H, C, K, float(sigma),
                                       _lib.ptr(weighted), _lib.ptr(nnum), _lib.stream_ptr()), 'geotr_kpconv_gather')
    return weighted, nnum


def maxpool(x, neighbor_indices, order=None):
    lib = _lib.load()
    x = _f32c(x)
    nb = neighbor_indices if neighbor_indices.is_contiguous() else neighbor_indices.contiguous()
    M, H = nb.shape
    out = torch.empty((M, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.geotr_maxpool_ordered(_lib.ptr(x), _lib.ptr(nb), M, x.shape[0], H, x.shape[1], _lib.ptr(_order(order, M)), _lib.ptr(out),
                                         _lib.stream_ptr()), 'geotr_maxpool_ordered')
    return out


def upsample_concat(coarse, upsample_indices, skip=None):
    """[nearest_upsample(coarse, upsample_indices), skip] along channels; only column 0 of the indices is read."""
    lib = _lib.load()
    coarse = _f32c(coarse)
    assert upsample_indices.dtype == torch.int64
    if upsample_indices.dim() == 1:
        upsample_indices = upsample_indices.unsqueeze(1)
    M = upsample_indices.shape[0]
    c1 = coarse.shape[1]
    c2 = 0 if skip is None else skip.shape[1]
    if skip is not None:
        skip = _f32c(skip)
    out = torch.empty((M, c1 + c2), dtype=torch.float32, device=coarse.device)
    _lib.check(lib.geotr_upsample_concat(_lib.ptr(coarse), coarse.shape[0], c1, _lib.ptr(upsample_indices),
                                         upsample_indices.stride(0), _lib.ptr(skip), c2, M, _lib.ptr(out),
                                         _lib.stream_ptr()), 'geotr_upsample_concat')
    return out


def group_norm(x, groups, weight, bias, eps=1e-5, residual=None, act=None):
    lib = _lib.load()
    x = _f32c(x)
    N, C = x.shape
    out = torch.empty_like(x)
    stats = _lib.workspace(lib.geotr_group_norm_workspace_bytes(N, C), x.device)
    if residual is not None:
        residual = _f32c(residual)
    _lib.check(lib.geotr_group_norm(_lib.ptr(x), N, C, groups, _lib.ptr(weight), _lib.ptr(bias), float(eps),
                                    _lib.ptr(residual), ACT[act], _lib.ptr(out), _lib.ptr(stats), _lib.stream_ptr()),
               'geotr_group_norm')
    return out


def group_norm_shortcut(x, shortcut, groups, weight, bias, sc_groups, sc_weight, sc_bias, eps=1e-5, sc_eps=1e-5, act=None, seg_rows=None):
    """act(GN(x) + GN(shortcut)) with the shortcut's normalised tensor never materialised (geotr_group_norm_shortcut); bit-identical to
    group_norm(x, ..., residual=group_norm(shortcut, ...)).  `seg_rows`: rows per stacked pair (statistics stay inside a segment)."""
    import ctypes
    lib = _lib.load()
    x, shortcut = _f32c(x), _f32c(shortcut)
    N, C = x.shape
    assert shortcut.shape == x.shape
    segs = [N] if seg_rows is None else [int(r) for r in seg_rows]
    out = torch.empty_like(x)
    stats = _lib.workspace(lib.geotr_group_norm_workspace_bytes(N, C), x.device)
    _lib.check(lib.geotr_group_norm_shortcut(_lib.ptr(x), _lib.ptr(shortcut), N, C, groups, _lib.ptr(weight), _lib.ptr(bias), float(eps),
                                             sc_groups, _lib.ptr(sc_weight), _lib.ptr(sc_bias), float(sc_eps), ACT[act], _lib.ptr(out),
                                             (ctypes.c_int64 * len(segs))(*segs), len(segs), _lib.ptr(stats), _lib.stream_ptr()),
               'geotr_group_norm_shortcut')
    return out


def layer_norm(x, weight, bias, eps=1e-5, residual=None):
    lib = _lib.load()
    shape = x.shape
    x2 = _f32c(x.reshape(-1, shape[-1]))
    r2 = _f32c(residual.reshape(-1, shape[-1])) if residual is not None else None
    out = torch.empty_like(x2)
    _lib.check(lib.geotr_layer_norm(_lib.ptr(x2), _lib.ptr(r2), x2.shape[0], x2.shape[1], _lib.ptr(weight), _lib.ptr(bias),
                                    float(eps), _lib.ptr(out), _lib.stream_ptr()), 'geotr_layer_norm')
    return out.view(shape)


def gse_knn(points, k):
    """(n, k) int32 indices of the k nearest other points (reference: dist_map.topk(k+1)[..., 1:])."""
    lib = _lib.load()
    points = _f32c(points)
    knn = torch.empty((points.shape[0], k), dtype=torch.int32, device=points.device)
    _lib.check(lib.geotr_gse_knn(_lib.ptr(points), points.shape[0], k, _lib.ptr(knn), _lib.stream_ptr()), 'geotr_gse_knn')
    return knn


GSE_PRECISION = 5  # 5: by table (default: proj(sinusoid(x)) tabulated as cubic Taylor coefficients, fp32, ~1e-6 of the reference);
                   # 0: fp32 MFMA (exact fp32 products); 1: split-bf16 "bf16x3" MFMA (~2^-17 relative error per product);
                   # 3: plain bf16 operands (~2^-8 per product)

_PRECISIONS = {'fp32': ('fp32', 5), 'bf16x3': (True, 5), 'bf16': ('bf16', 5), 'fp32-unpacked': (False, 5)}
_GSE_MFMA = {'fp32': 0, 'bf16x3': 1, 'bf16': 3, 'fp32-unpacked': 0}


class PrecisionName(str):
    """A precision mode's name that remembers the GSE form it was set with: `set_precision(prev)` restores both."""
    gse = 'table'

    def __new__(cls, name, gse='table'):
        obj = super().__new__(cls, name)
        obj.gse = gse
        return obj


def set_precision(name, gse=None):
    """Arithmetic of the matrix-pipe kernel family (packed GEMMs of the backbone / transformer) and form of the GSE embedding:
    'fp32' (DEFAULT): IEEE fp32 MFMA products with fp32 accumulation in every GEMM / KPConv / attention contraction -- the reference's
        own arithmetic -- on the packed pipeline (weights packed by geotr_gemm_pack_f32, v_mfma_f32_32x32x2_f32); the mode every
        reference-parity claim and the benchmark headline are made in;
    'bf16x3': split-bf16 products (three bf16 MFMA terms per product, ~2^-17 relative: narrower than fp32; the default of rounds 1-3);
    'fp32-unpacked': the same arithmetic on the rounds-1..3 kernel (no packed weights, no fused KPConv / epilogue statistics; A/B only);
    'bf16': plain bf16 operands with fp32 accumulation (BASELINE configs[4] "bf16 features").
    `gse`: 'table' (default) = the geometric structure embedding proj(sinusoid(x)) is read from cubic-Taylor tables in fp32 -- an
        APPROXIMATION of the reference's contraction (remainder <= 3.2e-7 max|W|, ~1e-6 of the embedding), independent of the GEMM mode;
        'mfma' = the fused sinusoid -> MFMA kernel in the named arithmetic (under 'fp32': exact fp32 MFMA products, the reference's
        arithmetic end to end).  None: the form carried by `name` when it is a value this function returned, else 'table'.
    Returns the previous mode as a PrecisionName (a str that carries its GSE form), so `set_precision(prev)` restores both.
    Process-wide; running models pick it up at their next forward."""
    global GEMM_PACKED, GSE_PRECISION
    if name not in _PRECISIONS:
        raise ValueError(f'unknown precision {name!r}: expected one of {sorted(_PRECISIONS)}')
    if gse is None:
        gse = getattr(name, 'gse', 'table')
    if gse not in ('table', 'mfma'):
        raise ValueError("gse must be 'table' or 'mfma'")
    prev_name = next((k for k, v in _PRECISIONS.items() if v[0] == GEMM_PACKED), 'custom')
    prev = PrecisionName(prev_name, 'table' if GSE_PRECISION == 5 else 'mfma')
    GEMM_PACKED, GSE_PRECISION = _PRECISIONS[str(name)]
    if gse == 'mfma':
        GSE_PRECISION = _GSE_MFMA[str(name)]
    return prev


class GseClouds(ctypes.Structure):  # geotr_gse_clouds
    _fields_ = [('count', ctypes.c_int32), ('n', ctypes.c_int32 * 32), ('row0', ctypes.c_int32 * 32), ('emb_off', ctypes.c_int64 * 32)]


class AttnGroups(ctypes.Structure):  # geotr_attn_groups
    _fields_ = [('count', ctypes.c_int32), ('pad_', ctypes.c_int32), ('n', ctypes.c_int64 * 32), ('m', ctypes.c_int64 * 32),
                ('ld', ctypes.c_int64 * 32), ('scores_off', ctypes.c_int64 * 32), ('q_row0', ctypes.c_int64 * 32),
                ('emb', ctypes.c_void_p * 32)]


class GsePos(ctypes.Structure):  # geotr_gse_pos
    _fields_ = [('q_row0', ctypes.c_int32 * 32), ('ld', ctypes.c_int32 * 32), ('pos_off', ctypes.c_int64 * 32)]


GSE_TABLE_SPAN = 64.0  # the distance table covers d / sigma_d <= 64 (12.8 m at the 3DMatch sigma_d); beyond: direct evaluation in-kernel
GSE_TABLE_DENSITY = 16  # grid points per unit index (kGseTabInv in csrc/transformer.hip)


def gse_table(div_term, weight, span):
    """Cubic-Taylor table (points, 4, D) of proj(sinusoid(x)) for x in [0, span] (geotr_gse_table_build)."""
    lib = _lib.load()
    weight, div_term = _f32c(weight.detach()), _f32c(div_term)
    d = weight.shape[0]
    points = int(math.ceil(span * GSE_TABLE_DENSITY)) + 2
    nbytes = lib.geotr_gse_table_bytes(d, points)
    table = torch.empty((points, 4, d), dtype=torch.float32, device=weight.device)
    ws = _lib.workspace(nbytes, weight.device)
    _lib.check(lib.geotr_gse_table_build(_lib.ptr(div_term), _lib.ptr(weight), d, points, _lib.ptr(table), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr()), 'geotr_gse_table_build')
    ws.record_stream(torch.cuda.current_stream())
    return table


def gse_tables(div_term, w_d, w_a, sigma_a):
    """(distance table, angle table): angles are at most 180 degrees, i.e. indices <= 180 / sigma_a."""
    return gse_table(div_term, w_d, GSE_TABLE_SPAN), gse_table(div_term, w_a, 180.0 / float(sigma_a))


def gse_embed(points, knn, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, precision=None, tables=None, reduction_a='max', qt=None):
    """(n, n, D) geometric structure embedding of one cloud (n, 3).  `tables` (precision 5): a cached `gse_tables` result.
    `reduction_a`: 'max' (every reference config) or 'mean' over the angular slots (geotransformer.py:65-68; table form only).
    `qt` (n, 4, 256), table form only: additionally returns pos (4, n, ld) with pos[h, i, j] = emb[i, j, :] . qt[i, h, :] -- the positional
    attention term of the first self-attention layer out of the same pass (geotr_gse_embed_table_ex)."""
    lib = _lib.load()
    points = _f32c(points)
    n, d = points.shape[0], w_d.shape[0]
    out = torch.empty((n, n, d), dtype=torch.float32, device=points.device)
    prof = PROFILE.get('gse_embed')
    if prof is not None:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
    precision = GSE_PRECISION if precision is None else int(precision)
    if reduction_a not in ('max', 'mean'):
        raise ValueError(f'Unsupported reduction mode: {reduction_a}.')
    pos = None
    if precision == 5:
        tab_d, tab_a = tables if tables is not None else gse_tables(div_term, w_d, w_a, sigma_a)
        cl = GseClouds()
        cl.count, cl.n[0], cl.row0[0], cl.emb_off[0] = 1, n, 0, 0
        pl = GsePos()
        if qt is not None:
            qt = _f32c(qt)
            assert qt.shape == (n, 4, d) and d == 256, 'the positional by-product is built for 4 heads of a 256-wide model'
            ld = (n + 3) // 4 * 4
            pos = torch.empty((4, n, ld), dtype=torch.float32, device=points.device)
            pl.q_row0[0], pl.ld[0], pl.pos_off[0] = 0, ld, 0
        if n > 0:
            _lib.check(lib.geotr_gse_embed_table_ex(_lib.ptr(points), _lib.ptr(knn), ctypes.byref(cl), knn.shape[1], d, _lib.ptr(tab_d),
                                                    tab_d.shape[0], _lib.ptr(tab_a), tab_a.shape[0], _lib.ptr(_f32c(w_d)), _lib.ptr(_f32c(b_d)),
                                                    _lib.ptr(_f32c(w_a)), _lib.ptr(_f32c(b_a)), _lib.ptr(_f32c(div_term)), float(sigma_d),
                                                    float(sigma_a), int(reduction_a == 'mean'), _lib.ptr(qt),
                                                    ctypes.byref(pl) if qt is not None else None, _lib.ptr(pos), _lib.ptr(out),
                                                    _lib.stream_ptr()), 'geotr_gse_embed_table_ex')
    else:
        if reduction_a != 'max' or qt is not None:
            raise NotImplementedError("reduction_a='mean' and the positional by-product exist in the table form of the embedding only "
                                      "(set_precision(..., gse='table'))")
        ws = _lib.workspace(lib.geotr_gse_embed_workspace_bytes(d, precision), points.device)
        _lib.check(lib.geotr_gse_embed(_lib.ptr(points), _lib.ptr(knn), n, knn.shape[1], d, _lib.ptr(_f32c(div_term)),
                                       _lib.ptr(_f32c(w_d)), _lib.ptr(b_d), _lib.ptr(_f32c(w_a)), _lib.ptr(b_a), float(sigma_d),
                                       float(sigma_a), precision, _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.stream_ptr()),
                   'geotr_gse_embed')
    if prof is not None:
        end.record()
        prof.append((start, end, n))
    return out if qt is None else (out, pos[:, :, :n])


def attn_softmax(scores, scale, emb=None, qt=None, qb=None, key_weights=None, key_masks=None, attention_factors=None, attention_masks=None,
                 pos=None):
    """In-place softmax over the last dim of scores (H, n, m); with `emb` adds the relative-position term first.
    key_weights (m), key_masks (m) bool, attention_factors (n, m), attention_masks (n, m) bool: the optional modifiers of the reference's
    attention layers (rpe_transformer.py:59-64, vanilla_transformer.py:57-64), applied to the scaled scores in the reference's order.
    `pos` (H, n, ld) + qb: the positional term precomputed by gse_embed(..., qt=...) instead of `emb` / `qt`."""
    lib = _lib.load()
    H, n, m = scores.shape
    ld = scores.stride(1)
    assert scores.stride(2) == 1 and scores.stride(0) == n * ld, 'scores must be (H, n, m) rows with a common leading dimension'
    c = 0
    if pos is not None:
        assert emb is None and pos.shape == scores.shape and pos.stride() == scores.stride() and key_weights is None and key_masks is None \
            and attention_factors is None and attention_masks is None
        g = AttnGroups()
        g.count, g.n[0], g.m[0], g.ld[0], g.scores_off[0], g.q_row0[0] = 1, n, m, ld, 0, 0
        _lib.check(lib.geotr_attn_softmax_grouped_pos(_lib.ptr(scores), ctypes.byref(g), _lib.ptr(pos), _lib.ptr(_f32c(qb)), H, float(scale),
                                                      _lib.stream_ptr()), 'geotr_attn_softmax_grouped_pos')
        return scores
    if emb is not None:
        emb, qt, qb = _f32c(emb), _f32c(qt), _f32c(qb)
        c = emb.shape[-1]
    if key_weights is None and key_masks is None and attention_factors is None and attention_masks is None:
        _lib.check(lib.geotr_attn_softmax(_lib.ptr(scores), ld, _lib.ptr(emb), _lib.ptr(qt), _lib.ptr(qb), n, m, c, H, float(scale),
                                          _lib.stream_ptr()), 'geotr_attn_softmax')
        return scores
    kw = None if key_weights is None else _f32c(key_weights).reshape(m)
    km = None if key_masks is None else key_masks.reshape(m).to(torch.uint8).contiguous()
    af = None if attention_factors is None else _f32c(attention_factors).reshape(n, m)
    am = None if attention_masks is None else attention_masks.reshape(n, m).to(torch.uint8).contiguous()
    _lib.check(lib.geotr_attn_softmax_ex(_lib.ptr(scores), ld, _lib.ptr(emb), _lib.ptr(qt), _lib.ptr(qb), n, m, c, H, float(scale),
                                         _lib.ptr(kw), _lib.ptr(km), _lib.ptr(af), m, _lib.ptr(am), m, _lib.stream_ptr()),
               'geotr_attn_softmax_ex')
    return scores


def point_to_node(points, nodes, point_limit):
    """-> point_to_node (N,) int64, node_masks (M,) bool, knn_indices (M,K) int64, knn_masks (M,K) bool."""
    lib = _lib.load()
    points, nodes = _f32c(points), _f32c(nodes)
    N, M, K = points.shape[0], nodes.shape[0], int(point_limit)
    dev = points.device
    p2n = torch.empty(N, dtype=torch.int64, device=dev)
    node_masks = torch.empty(M, dtype=torch.bool, device=dev)
    knn_idx = torch.empty((M, K), dtype=torch.int64, device=dev)
    knn_masks = torch.empty((M, K), dtype=torch.bool, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.geotr_point_to_node(_lib.ptr(points), N, _lib.ptr(nodes), M, K, _lib.ptr(p2n), _lib.ptr(node_masks),
                                       _lib.ptr(knn_idx), _lib.ptr(knn_masks), _lib.ptr(overflow), _lib.stream_ptr()),
               'geotr_point_to_node')
    return p2n, node_masks, knn_idx, knn_masks, overflow


def superpoint_match(ref_feats, src_feats, ref_masks, src_masks, num_correspondences, dual_normalization=True):
    """Top-k superpoint correspondences.  Returns (ref_idx (k,), src_idx (k,), scores (k,), count (1,) int32 device);
    only the first `count` entries are valid (count < k only if fewer than k valid superpoint pairs exist)."""
    lib = _lib.load()
    ref_feats, src_feats = _f32c(ref_feats), _f32c(src_feats)
    n, m, k = ref_feats.shape[0], src_feats.shape[0], int(num_correspondences)
    dev = ref_feats.device
    scores = gemm(ref_feats, src_feats)  # (n, m) = <f_r, f_s>
    ws = _lib.workspace(lib.geotr_superpoint_match_workspace_bytes(n, m), dev)
    ref_idx = torch.empty(k, dtype=torch.int64, device=dev)
    src_idx = torch.empty(k, dtype=torch.int64, device=dev)
    vals = torch.empty(k, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(lib.geotr_superpoint_match(_lib.ptr(scores), n, m, _lib.ptr(ref_masks), _lib.ptr(src_masks),
                                          int(bool(dual_normalization)), k, _lib.ptr(ws), ws.numel(),
                                          _lib.ptr(ref_idx), _lib.ptr(src_idx), _lib.ptr(vals), _lib.ptr(count),
                                          _lib.stream_ptr()), 'geotr_superpoint_match')
    return ref_idx, src_idx, vals, count


def patch_sinkhorn(alpha, num_iterations, ref_knn_masks, src_knn_masks, scores=None, ref_feats=None, src_feats=None,
                   ref_knn_indices=None, src_knn_indices=None):
    """(P, K+1, K+1) log optimal-transport scores; either from `scores` (P,K,K) or fused from features + patch indices."""
    lib = _lib.load()
    P, K = ref_knn_masks.shape
    assert src_knn_masks.shape == (P, K), 'square patches only'
    dev = ref_knn_masks.device
    out = torch.empty((P, K + 1, K + 1), dtype=torch.float32, device=dev)
    rm, sm = ref_knn_masks.contiguous(), src_knn_masks.contiguous()
    if scores is not None:
        scores = _f32c(scores)
        args = (None, 0, None, 0, 4, None, None)
    else:
        ref_feats, src_feats = _f32c(ref_feats), _f32c(src_feats)
        ri, si = ref_knn_indices.contiguous(), src_knn_indices.contiguous()
        args = (ref_feats, ref_feats.shape[0], src_feats, src_feats.shape[0], ref_feats.shape[1], ri, si)
    _lib.check(lib.geotr_patch_sinkhorn(_lib.ptr(args[0]), args[1], _lib.ptr(args[2]), args[3], args[4], _lib.ptr(args[5]),
                                        _lib.ptr(args[6]), _lib.ptr(rm), _lib.ptr(sm), P, K, _lib.ptr(alpha),
                                        int(num_iterations), _lib.ptr(scores), None, _lib.ptr(out), _lib.stream_ptr()),
               'geotr_patch_sinkhorn')
    return out


def weighted_procrustes(src_points, ref_points, weights=None):
    """(B, 4, 4) rigid transforms aligning src (B,N,3) to ref (B,N,3)."""
    lib = _lib.load()
    src_points, ref_points = _f32c(src_points), _f32c(ref_points)
    B, N, _ = src_points.shape
    if weights is not None:
        weights = _f32c(weights)
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=src_points.device)
    _lib.check(lib.geotr_weighted_procrustes(_lib.ptr(src_points), _lib.ptr(ref_points), _lib.ptr(weights), B, N,
                                             _lib.ptr(out), _lib.stream_ptr()), 'geotr_weighted_procrustes')
    return out


def lgr(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, topk, confidence_threshold, mutual,
        acceptance_radius, correspondence_threshold, num_refinement_steps, global_scores=None, correspondence_limit=None):
    """Local-to-global registration.  score_mat (P, R, R) with R >= K: the leading (K, K) block is used (drops dustbins).
    global_scores (P,): use_global_score; correspondence_limit: size of the verification set (local_global_registration.py:145-152, 225-226).
    Returns (ref_corr (cap,3), src_corr (cap,3), scores (cap,), num_corr (1,) int32 device, transform (4,4))."""
    lib = _lib.load()
    ref_knn_points, src_knn_points = _f32c(ref_knn_points), _f32c(src_knn_points)
    P, K, _ = ref_knn_points.shape
    assert score_mat.stride(2) == 1
    dev = ref_knn_points.device
    cap = P * K * int(topk)
    ref_corr = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    src_corr = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    scores = torch.empty(cap, dtype=torch.float32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    limit = 0 if correspondence_limit is None else int(correspondence_limit)
    if correspondence_limit is not None and limit < 1:
        raise ValueError('correspondence_limit must be a positive integer')
    gs = None if global_scores is None else _f32c(global_scores).reshape(P)
    ws = _lib.workspace(lib.geotr_lgr_ex_workspace_bytes(P, K, int(topk), limit), dev)
    _lib.check(lib.geotr_lgr_ex(_lib.ptr(ref_knn_points), _lib.ptr(src_knn_points), _lib.ptr(ref_knn_masks.contiguous()),
                                _lib.ptr(src_knn_masks.contiguous()), _lib.ptr(score_mat), score_mat.stride(0), score_mat.stride(1),
                                P, K, int(topk), float(confidence_threshold), int(bool(mutual)), float(acceptance_radius),
                                int(correspondence_threshold), int(num_refinement_steps), None, _lib.ptr(gs), limit, _lib.ptr(ref_corr),
                                _lib.ptr(src_corr), _lib.ptr(scores), _lib.ptr(num), _lib.ptr(T), _lib.ptr(ws), ws.numel(),
                                _lib.stream_ptr()), 'geotr_lgr_ex')
    return ref_corr, src_corr, scores, num, T


def l2_normalize(x):
    """Row-wise x / max(|x|_2, 1e-12) (F.normalize(p=2, dim=1))."""
    lib = _lib.load()
    x = _f32c(x)
    out = torch.empty_like(x)
    _lib.check(lib.geotr_l2_normalize(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out), _lib.stream_ptr()), 'geotr_l2_normalize')
    return out


def node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius, ref_masks=None,
                         src_masks=None, ref_knn_masks=None, src_knn_masks=None):
    """Ground-truth superpoint correspondences (geotr_node_correspondences).  Returns the full-capacity buffers
    (indices (M*N, 2) int64, overlaps (M*N,) fp32) and the device count (1,) int32; rows past the count are unspecified."""
    lib = _lib.load()
    ref_nodes, src_nodes = _f32c(ref_nodes), _f32c(src_nodes)
    ref_knn_points, src_knn_points = _f32c(ref_knn_points), _f32c(src_knn_points)
    dev = ref_nodes.device
    transform = _f32c(transform.to(dev))
    m, n, k = ref_nodes.shape[0], src_nodes.shape[0], ref_knn_points.shape[1]
    assert ref_knn_points.shape == (m, k, 3) and src_knn_points.shape == (n, k, 3) and transform.shape == (4, 4)

    def mask(t, shape):
        if t is None:
            return torch.ones(shape, dtype=torch.bool, device=dev)
        assert t.dtype == torch.bool and tuple(t.shape) == tuple(shape)
        return t.contiguous()

    ref_knn_masks, src_knn_masks = mask(ref_knn_masks, (m, k)), mask(src_knn_masks, (n, k))
    ref_masks = None if ref_masks is None else mask(ref_masks, (m,))
    src_masks = None if src_masks is None else mask(src_masks, (n,))
    idx = torch.empty((m * n, 2), dtype=torch.int64, device=dev)
    ov = torch.empty(m * n, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.geotr_node_correspondences_workspace_bytes(m, n, k), dev)
    _lib.check(lib.geotr_node_correspondences(
        _lib.ptr(ref_nodes), _lib.ptr(src_nodes), _lib.ptr(ref_knn_points), _lib.ptr(src_knn_points), _lib.ptr(transform),
        float(pos_radius), _lib.ptr(ref_masks), _lib.ptr(src_masks), _lib.ptr(ref_knn_masks), _lib.ptr(src_knn_masks), m, n, k,
        _lib.ptr(idx), _lib.ptr(ov), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), 'geotr_node_correspondences')
    return idx, ov, count
