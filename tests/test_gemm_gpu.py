"""GPU: the GEMM family and the stacked-pair primitives through the C ABI vs plain torch fp32/fp64 references."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('M,N,K,b_is_kn', [(1500, 256, 384, False), (4096, 64, 960, True), (2900, 128, 1920, True), (40000, 32, 480, True),
                                           (1024, 96, 64, False), (5000, 512, 128, False), (3000, 256, 32, False)])
def test_packed_split_bf16_gemm_vs_fp64(M, N, K, b_is_kn, matrix_precision):
    """geotr_gemm_pack + geotr_gemm_packed (split-bf16 MFMA, LDS-DMA pipeline) with the full epilogue vs an fp64 product:
    error budget ~2^-17 per product (stated tolerance 2e-5 of the output scale), far inside the 1e-4 feature-MSE bound.
    Exact-fp32 mode (geotr_gemm_pack_f32 + v_mfma_f32_32x32x2_f32 on the same pipeline): fp32 rounding only, 2e-6 of the scale."""
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) if b_is_kn else torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    div = torch.randint(0, 5, (M,), generator=g, dtype=torch.int32).cuda()
    packed = kernels.gemm_pack(w, b_is_kn=b_is_kn)
    got = kernels.gemm_packed(a, packed, N, bias=bias, row_div=div, residual=res, act='leaky')
    wt = w.double() if b_is_kn else w.double().t()
    want = (a.double() @ wt) / div.clamp(min=1).double()[:, None] + bias.double() + res.double()
    want = torch.where(want > 0, want, 0.1 * want)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= (2e-6 if matrix_precision == 'fp32' else 2e-5) * scale
    # exact-fp32 kernels on the same problem, for comparison of the two paths
    exact = kernels.gemm(a, w, b_is_kn=b_is_kn, bias=bias, row_div=div, residual=res, act='leaky')
    assert float((exact.double() - want).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize('M,N,K,b_is_kn', [(2900, 128, 1920, True), (2570, 256, 1920, True), (560, 256, 3840, True), (1100, 512, 3840, True),
                                           (1030, 64, 960, True), (1024, 256, 512, False), (3000, 32, 2048, False)])
def test_split_k_packed_gemm(M, N, K, b_is_kn, matrix_precision):
    """The narrow, deep launches of the coarse stages (fewer than 256 output tiles, K up to 15 x 256): gridDim.z K slices + the
    z-ordered reduce with the full epilogue.  vs fp64 (2e-5 of the output scale, as the unsplit kernel); vs the unsplit kernel
    (the same products, summed in another association: 4e-6 of the scale); run-to-run bit-identical (no atomics)."""
    from geotransformer_amd import _lib, kernels
    lib = _lib.load()
    mode = kernels.gemm_mode()
    assert lib.geotr_gemm_packed_splits(M, N, K, mode) > 1 and lib.geotr_gemm_packed_splitk_workspace_bytes(M, N, K) > 0, 'this shape is expected to split'
    assert lib.geotr_gemm_packed_splits(40000, 256, 384, mode) == 1  # wide launches never split
    assert lib.geotr_gemm_packed_splits(300, 128, 256, 0) == 1       # shallow ones neither (split-bf16 rule)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) if b_is_kn else torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    div = torch.randint(0, 5, (M,), generator=g, dtype=torch.int32).cuda()
    packed = kernels.gemm_pack(w, b_is_kn=b_is_kn)
    got = kernels.gemm_packed(a, packed, N, bias=bias, row_div=div, residual=res, act='leaky')
    again = kernels.gemm_packed(a, packed, N, bias=bias, row_div=div, residual=res, act='leaky')
    single = kernels.gemm_packed(a, packed, N, bias=bias, row_div=div, residual=res, act='leaky', split_k=False)
    wt = w.double() if b_is_kn else w.double().t()
    want = (a.double() @ wt) / div.clamp(min=1).double()[:, None] + bias.double() + res.double()
    want = torch.where(want > 0, want, 0.1 * want)
    scale = float(want.abs().max())
    assert torch.equal(got, again)
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    assert float((got - single).abs().max()) <= 4e-6 * scale
    plain = kernels.gemm_packed(a, packed, N)  # no epilogue terms at all
    assert float((plain.double() - a.double() @ wt).abs().max()) <= 2e-5 * float((a.double() @ wt).abs().max())


def test_grouped_gemm_matches_per_group_calls():
    """geotr_gemm_grouped (ragged groups x heads in one launch) is bit-identical to one geotr_gemm call per group."""
    from geotransformer_amd import _lib, kernels
    from geotransformer_amd.native import _bind  # noqa: F401  (loads the library)
    lib = _lib.load()

    class Groups(ctypes.Structure):
        _fields_ = [('count', ctypes.c_int32), ('pad_', ctypes.c_int32)] + [(n, ctypes.c_int64 * 32) for n in (
            'm', 'n', 'k', 'lda', 'ldb', 'ldc', 'a_off', 'b_off', 'c_off', 'a_head_stride', 'b_head_stride', 'c_head_stride')]

    H, ch = 4, 64
    C = H * ch
    sizes = [(251, 293), (300, 17), (64, 401), (5, 5)]
    g = torch.Generator().manual_seed(3)
    rows_q, rows_k = sum(n for n, _ in sizes), sum(m for _, m in sizes)
    q = torch.randn(rows_q, C, generator=g).cuda()
    k = torch.randn(rows_k, C, generator=g).cuda()
    gr = Groups()
    gr.count = len(sizes)
    total, qo, ko = 0, 0, 0
    want = []
    for i, (n, m) in enumerate(sizes):
        mp = (m + 3) // 4 * 4
        gr.m[i], gr.n[i], gr.k[i], gr.lda[i], gr.ldb[i], gr.ldc[i] = n, m, ch, C, C, mp
        gr.a_off[i], gr.b_off[i], gr.c_off[i] = qo * C, ko * C, total
        gr.a_head_stride[i], gr.b_head_stride[i], gr.c_head_stride[i] = ch, ch, n * mp
        out_i = torch.zeros((H, n, mp), device='cuda')
        kernels.gemm(q[qo:qo + n].view(n, H, ch).permute(1, 0, 2), k[ko:ko + m].view(m, H, ch).permute(1, 0, 2), out=out_i[:, :, :m])
        want.append(out_i)
        total += H * n * mp
        qo, ko = qo + n, ko + m
    scores = torch.zeros(total, device='cuda')
    _lib.check(lib.geotr_gemm_grouped(_lib.ptr(q), _lib.ptr(k), 0, _lib.ptr(scores), ctypes.byref(gr), H, 1.0, _lib.stream_ptr()), 'grouped')
    off, q0, k0 = 0, 0, 0
    for (n, m), w in zip(sizes, want):
        mp = (m + 3) // 4 * 4
        got = scores[off:off + H * n * mp].view(H, n, mp)
        assert torch.equal(got[:, :, :m], w[:, :, :m])
        ref = torch.einsum('nhc,mhc->hnm', q[q0:q0 + n].view(n, H, ch), k[k0:k0 + m].view(m, H, ch))
        assert torch.allclose(got[:, :, :m], ref, atol=1e-4, rtol=1e-5)
        off += H * n * mp
        q0, k0 = q0 + n, k0 + m


def test_segmented_group_norm_equals_per_segment_calls():
    """geotr_group_norm_segmented: a segment's result does not depend on what it is stacked with (bit-identical to a lone call),
    and matches torch.nn.functional.group_norm over that segment's rows."""
    from geotransformer_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    C, groups = 64, 32
    seg = [1000, 37, 4096, 513]
    n = sum(seg)
    x = (torch.randn(n, C, generator=g) * 3 + 1).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    res = torch.randn(n, C, generator=g).cuda()

    def run(xs, rs, rows):
        out = torch.empty_like(xs)
        ws = _lib.workspace(lib.geotr_group_norm_workspace_bytes(xs.shape[0], C), xs.device)
        arr = (ctypes.c_int64 * len(rows))(*rows)
        _lib.check(lib.geotr_group_norm_segmented(_lib.ptr(xs), xs.shape[0], C, groups, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(rs), 2,
                                                  _lib.ptr(out), arr, len(rows), _lib.ptr(ws), _lib.stream_ptr()), 'gn_seg')
        return out

    stacked = run(x, res, seg)
    r0 = 0
    for rows in seg:
        alone = run(x[r0:r0 + rows].contiguous(), res[r0:r0 + rows].contiguous(), [rows])
        assert torch.equal(stacked[r0:r0 + rows], alone)
        ref = torch.nn.functional.group_norm(x[r0:r0 + rows].t().unsqueeze(0), groups, gamma, beta, 1e-5).squeeze(0).t() + res[r0:r0 + rows]
        ref = torch.where(ref > 0, ref, 0.1 * ref)
        assert torch.allclose(alone, ref, atol=2e-4, rtol=2e-4)
        r0 += rows


@pytest.mark.parametrize('C', [32, 64, 128, 256])
def test_group_norm_row_flags_and_row_positive(C):
    """geotr_group_norm_segmented_flags: the GroupNorm output's (row sum > 0) flags, produced inside the apply kernel, equal the
    separate geotr_row_positive pass over that output (same predicate KPConv's neighbour count uses, kpconv.py:113-115) and the
    torch row sums wherever the sum is not within rounding of zero."""
    from geotransformer_amd import _lib, kernels
    lib = _lib.load()
    assert lib.geotr_group_norm_flags_supported(C) == 1 and lib.geotr_group_norm_flags_supported(512) == 0
    g = torch.Generator().manual_seed(C)
    seg = [3000, 777, 2048]
    n = sum(seg)
    x = (torch.randn(n, C, generator=g) * 2).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    out, flags = torch.empty_like(x), torch.full((n,), 7, dtype=torch.uint8, device='cuda')
    ws = _lib.workspace(lib.geotr_group_norm_workspace_bytes(n, C), x.device)
    arr = (ctypes.c_int64 * len(seg))(*seg)
    _lib.check(lib.geotr_group_norm_segmented_flags(_lib.ptr(x), n, C, 32 if C >= 32 else 4, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, None, 2,
                                                    _lib.ptr(out), arr, len(seg), _lib.ptr(ws), _lib.ptr(flags), _lib.stream_ptr()), 'gn_flags')
    plain = torch.empty_like(x)
    _lib.check(lib.geotr_group_norm_segmented(_lib.ptr(x), n, C, 32 if C >= 32 else 4, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, None, 2,
                                              _lib.ptr(plain), arr, len(seg), _lib.ptr(ws), _lib.stream_ptr()), 'gn')
    assert torch.equal(out, plain)  # the flag output does not change the normalisation
    sums = out.double().sum(1)
    clear = sums.abs() > 1e-4 * out.abs().double().sum(1)
    assert int(flags.max()) <= 1
    assert torch.equal(flags.bool()[clear], (sums > 0)[clear])
    sep = kernels.row_positive(out)
    assert torch.equal(sep.bool()[clear], (sums > 0)[clear])
    assert float((sep.bool() != flags.bool()).float().mean()) <= 1e-3  # another summation order: only rows with |sum| ~ 0 may differ


@pytest.mark.parametrize('C,segs', [(128, [9000, 777, 2048]), (256, [300, 301]), (1024, [517]), (6, [40, 33])])
def test_group_norm_shortcut_is_bitwise_two_group_norms(C, segs):
    """geotr_group_norm_shortcut -- act(GN(x) + GN'(shortcut)) with the shortcut's affine applied inside the apply pass of x, its
    normalised tensor never written -- equals the two-pass form bit for bit (segments of both statistics-block sizes, a width that is
    not a multiple of 4), and the torch formula within fp32 tolerance."""
    from geotransformer_amd import _lib, kernels
    lib = _lib.load()
    g = torch.Generator().manual_seed(C)
    n = sum(segs)
    groups = 32 if C >= 32 else 3
    x, t = (torch.randn(n, C, generator=g) * 2 + 0.3).cuda(), (torch.randn(n, C, generator=g) * 0.7 - 0.1).cuda()
    w1, b1, w2, b2 = [torch.randn(C, generator=g).cuda() for _ in range(4)]
    arr = (ctypes.c_int64 * len(segs))(*segs)
    ws = _lib.workspace(lib.geotr_group_norm_workspace_bytes(n, C), x.device)
    sc, two = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.geotr_group_norm_segmented(_lib.ptr(t), n, C, groups, _lib.ptr(w2), _lib.ptr(b2), 1e-5, None, 0, _lib.ptr(sc), arr, len(segs),
                                              _lib.ptr(ws), _lib.stream_ptr()), 'gn shortcut')
    _lib.check(lib.geotr_group_norm_segmented(_lib.ptr(x), n, C, groups, _lib.ptr(w1), _lib.ptr(b1), 1e-5, _lib.ptr(sc), 2, _lib.ptr(two), arr,
                                              len(segs), _lib.ptr(ws), _lib.stream_ptr()), 'gn main')
    fused = kernels.group_norm_shortcut(x, t, groups, w1, b1, groups, w2, b2, act='leaky', seg_rows=segs)
    assert torch.equal(fused, two)
    want, r0 = [], 0
    for rows in segs:  # per segment: statistics over all its rows (modules.py:47-50)
        def gn(v, w, b):
            return F.group_norm(v.t().unsqueeze(0), groups, w, b, 1e-5).squeeze(0).t()
        want.append(F.leaky_relu(gn(x[r0:r0 + rows], w1, b1) + gn(t[r0:r0 + rows], w2, b2), 0.1))
        r0 += rows
    assert torch.allclose(fused, torch.cat(want), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize('N,K,segs', [(128, 32, [5000, 3333, 129, 128, 1, 77]), (32, 64, [2049, 4000]), (64, 64, [1500, 1500, 700]),
                                      (256, 128, [40000]), (96, 256, [1024, 255, 1300])])
def test_group_norm_statistics_out_of_the_gemm_epilogue(N, K, segs, matrix_precision):
    """geotr_gemm_packed_stats + geotr_group_norm_stats (round 3): row tiles aligned to the row segments, the GroupNorm statistics
    of the output written by the GEMM's epilogue.  (a) the output is bitwise the plain packed GEMM's; (b) every record holds the
    column sums / sums of squares of its rows; (c) GroupNorm from the records agrees with the statistics-pass GroupNorm of each
    segment to fp32 rounding; (d) a segment alone gives bitwise the records and the normalised rows it gives inside a stack."""
    from geotransformer_amd import _lib, kernels
    g = torch.Generator().manual_seed(N * 1000 + K + len(segs))
    M = sum(segs)
    a = (torch.randn(M, K, generator=g) * 1.5 + 0.3).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gamma, beta = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    groups = 8
    y, stats, rpr = kernels.linear_gn(a, w, bias, seg_rows=segs)
    assert stats is not None and rpr == (64 if N > 64 else 32)
    # (a) (the single-pass product: the exact-fp32 plan splits some of these narrow grids over K, which re-associates the sum)
    assert torch.equal(y, kernels.gemm_packed(a, kernels.gemm_pack(w), N, bias=bias, split_k=False))
    rec = stats.view(-1, 2, N).double().cpu()
    yd = y.double().cpu()
    r0, b0 = 0, 0
    for rows in segs:                                                                                                 # (b)
        nrec = (rows + 127) // 128 * (128 // rpr)
        for b in range(nrec):
            blk = yd[r0 + b * rpr: r0 + min(rows, (b + 1) * rpr)]
            assert torch.allclose(rec[b0 + b, 0], blk.sum(0), rtol=1e-5, atol=1e-4), (rows, b)
            assert torch.allclose(rec[b0 + b, 1], (blk * blk).sum(0), rtol=1e-5, atol=1e-4), (rows, b)
        r0, b0 = r0 + rows, b0 + nrec
    assert b0 == rec.shape[0]
    out = kernels.group_norm_stats(y, groups, gamma, beta, x_stats=stats, x_rpr=rpr, act='leaky', seg_rows=segs)
    r0 = 0
    for i, rows in enumerate(segs):
        want = kernels.group_norm(y[r0:r0 + rows].contiguous(), groups, gamma, beta, act='leaky')                    # (c)
        assert torch.allclose(out[r0:r0 + rows], want, rtol=2e-5, atol=2e-5), i
        if rows >= kernels.PACKED_MIN_ROWS:                                                                           # (d)
            ya, sa, _ = kernels.linear_gn(a[r0:r0 + rows].contiguous(), w, bias)
            alone = kernels.group_norm_stats(ya, groups, gamma, beta, x_stats=sa, x_rpr=rpr, act='leaky')
            assert torch.equal(ya, y[r0:r0 + rows]) and torch.equal(alone, out[r0:r0 + rows]), i
        r0 += rows
    # the residual with its own norm: both statistics from records == the two-pass composition
    t, st, _ = kernels.linear_gn(a, (w * 0.7 + 0.1).contiguous(), bias, seg_rows=segs)
    fused = kernels.group_norm_stats(y, groups, gamma, beta, x_stats=stats, x_rpr=rpr, residual=t, res_stats=st, res_rpr=rpr,
                                     res_norm=(4, beta, gamma, 1e-5), act='leaky', seg_rows=segs)
    tn = kernels.group_norm_stats(t, 4, beta, gamma, x_stats=st, x_rpr=rpr, seg_rows=segs)
    two = kernels.group_norm_stats(y, groups, gamma, beta, x_stats=stats, x_rpr=rpr, residual=tn, act='leaky', seg_rows=segs)
    assert torch.equal(fused, two)


@pytest.mark.parametrize('lat_ch,skip_ch,n_out,nc,m,segs', [(512, 256, 256, 3000, 9000, [4000, 5000]), (1024, 512, 512, 1100, 4200, None),
                                                            (256, 128, 96, 2048, 2048, [1024, 1024])])
def test_decoder_linear_without_the_concatenation(lat_ch, skip_ch, n_out, nc, m, segs, matrix_precision):
    """Linear(cat(nearest_upsample(latent), skip)) = up(latent W_latent^T) + skip W_skip^T + b (geotr_gemm_packed_gather): vs the
    concatenated form in fp64, pad indices (== number of coarse rows) contribute nothing, GroupNorm statistics of the sum."""
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(lat_ch + m)
    latent = torch.randn(nc, lat_ch, generator=g).cuda()
    skip = torch.randn(m, skip_ch, generator=g).cuda()
    w = (torch.randn(n_out, lat_ch + skip_ch, generator=g) * 0.05).cuda()
    bias = torch.randn(n_out, generator=g).cuda()
    up = torch.randint(0, nc, (m, 5), generator=g)
    up[::17, 0] = nc  # pad rows of the upsampling table
    up = up.cuda()
    y, stats, rpr = kernels.decoder_linear(latent, up, skip, w, bias, want_stats=True, seg_rows=segs)
    padded = torch.cat([latent, torch.zeros_like(latent[:1])]).double()
    cat = torch.cat([padded[up[:, 0]], skip.double()], dim=1)
    want = cat @ w.double().t() + bias.double()
    assert float((y.double() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    # the concatenated route through the same packed kernels agrees to fp32 rounding of the two partial sums
    ref = kernels.linear(kernels.upsample_concat(latent, up, skip), w, bias, packed=True)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    rec = stats.view(-1, 2, n_out).double().sum(0).cpu() if segs is None else None
    if rec is not None:
        assert torch.allclose(rec[0], y.double().sum(0).cpu(), rtol=1e-5, atol=1e-3)
        assert torch.allclose(rec[1], (y.double() ** 2).sum(0).cpu(), rtol=1e-5, atol=1e-3)
    gamma, beta = torch.ones(n_out).cuda(), torch.zeros(n_out).cuda()
    out = kernels.group_norm_stats(y, 8, gamma, beta, x_stats=stats, x_rpr=rpr, act='leaky', seg_rows=segs)
    r0 = 0
    for rows in (segs or [m]):
        assert torch.allclose(out[r0:r0 + rows], kernels.group_norm(y[r0:r0 + rows].contiguous(), 8, gamma, beta, act='leaky'), rtol=2e-5, atol=2e-5)
        r0 += rows


@pytest.mark.parametrize('mid,c_in,c_out,segs,linear_shortcut', [(32, 64, 128, [5000, 3333, 700], True), (64, 256, 256, [4000, 4100], False),
                                                                 (128, 256, 512, [2100], True), (32, 128, 128, [1500, 1501], False)])
def test_residual_tail_without_the_apply_pass_is_bitwise_the_apply_pass(mid, c_in, c_out, segs, linear_shortcut, matrix_precision):
    """geotr_gemm_packed_tail + geotr_group_norm_finalize (round 3): leaky(GN(unary2(y)) + shortcut) with every product launched once for
    its statistics and once more with the GroupNorm applied in its epilogue -- bit for bit what the statistics-epilogue + apply-pass
    path (linear_gn + group_norm_stats) stores, for an identity shortcut and for a shortcut with its own Linear + GroupNorm."""
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(mid + c_out + len(segs))
    M = sum(segs)
    y = torch.randn(M, mid, generator=g).cuda()
    w2, b2 = (torch.randn(c_out, mid, generator=g) * 0.2).cuda(), torch.randn(c_out, generator=g).cuda()
    g2, be2 = (torch.rand(c_out, generator=g) + 0.5).cuda(), torch.randn(c_out, generator=g).cuda()
    groups = 8
    z, sz, rpr = kernels.linear_gn(y, w2, b2, seg_rows=segs)
    if linear_shortcut:
        x = torch.randn(M, c_in, generator=g).cuda()
        ws, bs = (torch.randn(c_out, c_in, generator=g) * 0.1).cuda(), torch.randn(c_out, generator=g).cuda()
        gs, bes = (torch.rand(c_out, generator=g) + 0.5).cuda(), torch.randn(c_out, generator=g).cuda()
        t, st, _ = kernels.linear_gn(x, ws, bs, seg_rows=segs)
        want = kernels.group_norm_stats(z, groups, g2, be2, x_stats=sz, x_rpr=rpr, residual=t, res_stats=st, res_rpr=rpr,
                                        res_norm=(4, gs, bes, 1e-5), act='leaky', seg_rows=segs)
        got = kernels.residual_tail(y, w2, b2, (groups, g2, be2, 1e-5), x, ws, bs, (4, gs, bes, 1e-5), seg_rows=segs)
    else:
        x = torch.randn(M, c_out, generator=g).cuda()
        want = kernels.group_norm_stats(z, groups, g2, be2, x_stats=sz, x_rpr=rpr, residual=x, act='leaky', seg_rows=segs)
        got = kernels.residual_tail(y, w2, b2, (groups, g2, be2, 1e-5), x, seg_rows=segs)
    assert torch.equal(got, want)


def test_packed_weight_format_is_checked_against_the_arithmetic_mode():
    """the two pack layouts have the same size; a buffer packed for one arithmetic must not be consumed by the other.  The
    library records the layout per buffer address (geotr_gemm_pack_format) and every packed entry point refuses a mismatch."""
    from geotransformer_amd import _lib, kernels
    lib = _lib.load()
    w = torch.randn(64, 64).cuda()
    a = torch.randn(2048, 64).cuda()
    prev = kernels.set_precision('fp32')
    try:
        p32 = kernels.gemm_pack(w)
        assert lib.geotr_gemm_pack_format(_lib.ptr(p32)) == 2
        kernels.gemm_packed(a, p32, 64)  # matching mode: fine
        kernels.set_precision('bf16x3')
        with pytest.raises(RuntimeError, match='geotr_gemm_pack_f32'):
            kernels.gemm_packed(a, p32, 64)  # fp32 plane read as bf16 planes: refused
        p16 = kernels.gemm_pack(w)  # (the cache key carries the mode: a fresh buffer in the other layout)
        assert p16.data_ptr() != p32.data_ptr() and lib.geotr_gemm_pack_format(_lib.ptr(p16)) == 1
        kernels.set_precision('fp32')
        with pytest.raises(RuntimeError, match='geotr_gemm_pack'):
            kernels.gemm_packed(a, p16, 64)
        # (an address that was never packed reads 0; `a` may sit where an earlier test's packed buffer lived -- the record is per address and
        #  is overwritten by the next pack call on it, so only `packed` arguments are ever looked up)
    finally:
        kernels.set_precision(prev)
