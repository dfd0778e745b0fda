"""Binding of the native executor (geotr_model_forward, include/geotr.h): descriptor structs + one-call forward.

The ctypes Structures below mirror the C structs field for field.  `NativeModel` walks a `GeoTransformer` module once,
records the device pointers of its parameters (re-built automatically if a parameter is re-allocated or modified), and
`forward(data_dict)` allocates the output tensors, sizes the workspace and makes ONE asynchronous C-ABI call per pair.
"""
import ctypes
import threading

import torch

from . import _lib, kernels

import os

_POISON = os.environ.get('GEOTR_POISON_WS') == '1'  # fill workspaces / output buffers with 0xFF before every forward (debug)
# GEOTR_ALLOC_LOG=1 (hazard investigation, scripts/hazard_probe.py): (thread, what, first byte, bytes, host time) of every buffer the two
# native calls are handed, so that a stale word's address can be traced back to the buffers that owned it before
ALLOC_LOG = [] if os.environ.get('GEOTR_ALLOC_LOG') == '1' else None


def _log_buffers(what, tensors):
    if ALLOC_LOG is None:
        return
    import time
    now, who = time.perf_counter(), threading.current_thread().name
    for name, t in tensors:
        ALLOC_LOG.append((who, f'{what}.{name}', t.data_ptr(), t.numel() * t.element_size(), now))
MAX_STAGES = 5
MAX_PAIRS = 16  # GEOTR_MAX_PAIRS
P_F32 = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int32
F32 = ctypes.c_float


class Linear(ctypes.Structure):
    _fields_ = [('w', P_F32), ('b', P_F32), ('in_', I64), ('out', I64), ('packed', ctypes.c_void_p)]


class Norm(ctypes.Structure):
    _fields_ = [('gamma', P_F32), ('beta', P_F32), ('groups', I64), ('eps', F32), ('pad_', I32)]


class KPConvDesc(ctypes.Structure):
    _fields_ = [('weights', P_F32), ('bias', P_F32), ('kernel_points', P_F32), ('in_', I64), ('out', I64),
                ('num_kernel_points', I64), ('sigma', F32), ('pad_', I32), ('packed', ctypes.c_void_p)]


class Block(ctypes.Structure):
    _fields_ = [('is_conv_block', I32), ('has_unary1', I32), ('has_shortcut', I32), ('strided', I32),
                ('unary1', Linear), ('unary1_norm', Norm), ('conv', KPConvDesc), ('conv_norm', Norm),
                ('unary2', Linear), ('unary2_norm', Norm), ('shortcut', Linear), ('shortcut_norm', Norm)]


class Backbone(ctypes.Structure):
    _fields_ = [('num_stages', I32), ('fine_stage', I32), ('num_blocks', I32), ('num_decoders', I32),
                ('blocks', Block * (2 + 3 * (MAX_STAGES - 1))), ('decoder', Linear * MAX_STAGES),
                ('decoder_norm', Norm * MAX_STAGES), ('decoder_packed_latent', ctypes.c_void_p * MAX_STAGES),
                ('decoder_packed_skip', ctypes.c_void_p * MAX_STAGES)]


class Pyramid(ctypes.Structure):
    _fields_ = [('num_stages', I32), ('num_pairs', I32),
                ('points', P_F32 * MAX_STAGES), ('n', I64 * MAX_STAGES),
                ('neighbors', P_F32 * MAX_STAGES), ('neighbors_w', I64 * MAX_STAGES),
                ('subsampling', P_F32 * MAX_STAGES), ('subsampling_w', I64 * MAX_STAGES),
                ('upsampling', P_F32 * MAX_STAGES), ('upsampling_w', I64 * MAX_STAGES),
                ('cloud_n', (I64 * (2 * MAX_PAIRS)) * MAX_STAGES), ('order', P_F32 * MAX_STAGES)]


class PyramidBuffers(ctypes.Structure):
    _fields_ = [('points', P_F32 * MAX_STAGES), ('lengths', P_F32 * MAX_STAGES), ('neighbors', P_F32 * MAX_STAGES),
                ('subsampling', P_F32 * MAX_STAGES), ('upsampling', P_F32 * MAX_STAGES), ('order', P_F32 * MAX_STAGES)]


class AttnLayer(ctypes.Structure):
    _fields_ = [('is_self', I32), ('pad_', I32), ('q', Linear), ('k', Linear), ('v', Linear), ('p', Linear), ('out', Linear),
                ('expand', Linear), ('squeeze', Linear), ('norm', Norm), ('out_norm', Norm),
                ('qkv_w', P_F32), ('qkv_b', P_F32), ('kv_w', P_F32), ('kv_b', P_F32), ('qkv_packed', ctypes.c_void_p), ('kv_packed', ctypes.c_void_p)]


class Transformer(ctypes.Structure):
    _fields_ = [('num_layers', I32), ('num_heads', I32), ('angle_k', I32), ('reduction_a', I32), ('sigma_d', F32), ('sigma_a', F32),
                ('gse_precision', I32), ('pad2_', I32), ('gse_table_d', P_F32), ('gse_table_a', P_F32), ('gse_points_d', I64),
                ('gse_points_a', I64), ('div_term', P_F32), ('proj_d', Linear), ('proj_a', Linear), ('in_proj', Linear), ('out_proj', Linear),
                ('layers', AttnLayer * 8)]


class Model(ctypes.Structure):
    _fields_ = [('backbone', Backbone), ('transformer', Transformer), ('alpha', P_F32),
                ('num_points_in_patch', I64), ('num_correspondences', I64), ('num_sinkhorn_iterations', I64),
                ('dual_normalization', I32), ('topk', I32), ('mutual', I32), ('correspondence_threshold', I32),
                ('num_refinement_steps', I32), ('gemm_mode', I32), ('confidence_threshold', F32), ('acceptance_radius', F32)]


class Outputs(ctypes.Structure):
    _fields_ = [(name, P_F32) for name in (
        'feats_c', 'feats_f', 'ref_node_corr_indices', 'src_node_corr_indices', 'node_corr_scores', 'num_node_corr',
        'ref_knn_indices', 'src_knn_indices', 'ref_knn_masks', 'src_knn_masks', 'ref_knn_points', 'src_knn_points',
        'matching_scores', 'ref_corr_points', 'src_corr_points', 'corr_scores', 'num_corr', 'estimated_transform')]


_bound = False


def _bind():
    global _bound
    lib = _lib.load()
    if not _bound:
        lib.geotr_model_workspace_bytes.restype = ctypes.c_size_t
        lib.geotr_model_workspace_bytes.argtypes = [ctypes.POINTER(Model), ctypes.POINTER(Pyramid)]
        lib.geotr_model_forward.restype = ctypes.c_int
        lib.geotr_model_forward.argtypes = [ctypes.POINTER(Model), ctypes.POINTER(Pyramid), ctypes.c_void_p,
                                            ctypes.POINTER(Outputs), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _bound = True
    return lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _linear(mod, keep=None):
    """`keep` (a list) requests the packed split-bf16 weight for tall activations; it holds the tensor alive."""
    packed = None
    if keep is not None and kernels.GEMM_PACKED:
        keep.append(kernels.gemm_pack(mod.weight))
        packed = keep[-1].data_ptr()
    return Linear(_ptr(mod.weight), _ptr(mod.bias), mod.weight.shape[1], mod.weight.shape[0], packed)


def _norm(mod):
    """GroupNorm wrapper (modules.kpconv.GroupNorm), nn.LayerNorm, or None."""
    if mod is None:
        return Norm(None, None, 0, 0.0, 0)
    if hasattr(mod, 'norm'):  # kpconv.GroupNorm holds an nn.GroupNorm
        g = mod.norm
        return Norm(_ptr(g.weight), _ptr(g.bias), g.num_groups, g.eps, 0)
    return Norm(_ptr(mod.weight), _ptr(mod.bias), 0, mod.eps, 0)


def _kpconv(mod, keep=None):
    packed = None
    if keep is not None and kernels.GEMM_PACKED:
        keep.append(kernels.gemm_pack(mod.weights, b_is_kn=True, view=(mod.kernel_size * mod.in_channels, mod.out_channels)))
        packed = keep[-1].data_ptr()
    return KPConvDesc(_ptr(mod.weights), _ptr(mod.bias), _ptr(mod.kernel_points), mod.in_channels, mod.out_channels,
                      mod.kernel_size, float(mod.sigma), 0, packed)


def _block(mod, keep=None):
    from .modules.kpconv.modules import ConvBlock, UnaryBlock
    b = Block()
    if isinstance(mod, ConvBlock):
        b.is_conv_block = 1
        b.conv, b.conv_norm = _kpconv(mod.KPConv, keep), _norm(mod.norm)
        return b
    b.strided = int(mod.strided)
    if isinstance(mod.unary1, UnaryBlock):
        b.has_unary1 = 1
        b.unary1, b.unary1_norm = _linear(mod.unary1.mlp, keep), _norm(mod.unary1.norm)
    b.conv, b.conv_norm = _kpconv(mod.KPConv, keep), _norm(mod.norm_conv)
    b.unary2, b.unary2_norm = _linear(mod.unary2.mlp, keep), _norm(mod.unary2.norm)
    if isinstance(mod.unary_shortcut, UnaryBlock):
        b.has_shortcut = 1
        b.shortcut, b.shortcut_norm = _linear(mod.unary_shortcut.mlp, keep), _norm(mod.unary_shortcut.norm)
    return b


class NativeModel:
    """Descriptor of a GeoTransformer module for the native executor (pointers are re-read when parameters change)."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self._keep = []
        self._gen = None  # (descriptor, [tensors its raw pointers refer to]) -- published and retired as ONE object
        self._lock = threading.Lock()

    @property
    def desc(self):
        return None if self._gen is None else self._gen[0]

    def _version_key(self):
        """Everything whose raw device pointer is baked into the descriptor: parameters AND buffers (KPConv.kernel_points,
        the embedding's div_term), by (data_ptr, version)."""
        m = self.model
        return (kernels.GEMM_PACKED, kernels.GSE_PRECISION, kernels.DECODER_SPLIT) + tuple(
            (t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))

    def _build(self):
        m = self.model
        self._keep = []
        d = Model()
        bb, net = d.backbone, m.backbone
        S = net.num_stages
        bb.num_stages, bb.fine_stage = S, net.fine_stage
        blocks = [net.encoder1_1, net.encoder1_2]
        for s in range(2, S + 1):
            blocks += [getattr(net, f'encoder{s}_{i}') for i in (1, 2, 3)]
        bb.num_blocks = len(blocks)
        for i, blk in enumerate(blocks):
            bb.blocks[i] = _block(blk, self._keep)  # backbone weights also in packed split-bf16 form (tall activations)
        n_dec = 0
        for i in range(S - 2, net.fine_stage - 1, -1):
            dec = getattr(net, f'decoder{i + 1}')
            bb.decoder[n_dec] = _linear(dec.mlp, self._keep)
            bb.decoder_norm[n_dec] = _norm(getattr(dec, 'norm', None))
            if kernels.GEMM_PACKED and kernels.DECODER_SPLIT:  # W = [W_latent | W_skip] packed slice by slice (kernels.decoder_linear)
                lat, skp = kernels.decoder_packs(dec.mlp.weight, net.decoder_latent_channels(i))
                self._keep += [lat, skp]
                bb.decoder_packed_latent[n_dec], bb.decoder_packed_skip[n_dec] = lat.data_ptr(), skp.data_ptr()
            n_dec += 1
        bb.num_decoders = n_dec
        t, tr = d.transformer, m.transformer
        layers = tr.transformer.layers
        t.num_layers, t.num_heads = len(layers), layers[0].attention.attention.num_heads
        t.angle_k, t.sigma_d, t.sigma_a = tr.embedding.angle_k, float(tr.embedding.sigma_d), float(tr.embedding.sigma_a)
        t.gse_precision = int(kernels.GSE_PRECISION)
        t.reduction_a = int(tr.embedding.reduction_a == 'mean')  # (the executor refuses 'mean' with the MFMA forms of the embedding)
        if t.gse_precision == 5:  # lookup tables of proj_d / proj_a, built once per weight set (kept alive with the descriptor)
            tab_d, tab_a = tr.embedding.tables()
            self._keep += [tab_d, tab_a]
            t.gse_table_d, t.gse_table_a = tab_d.data_ptr(), tab_a.data_ptr()
            t.gse_points_d, t.gse_points_a = tab_d.shape[0], tab_a.shape[0]
        t.div_term = _ptr(tr.embedding.embedding.div_term)
        t.proj_d, t.proj_a = _linear(tr.embedding.proj_d), _linear(tr.embedding.proj_a)
        t.in_proj, t.out_proj = _linear(tr.in_proj, self._keep), _linear(tr.out_proj, self._keep)  # packed: used when pairs are stacked
        for i, (kind, layer) in enumerate(zip(tr.transformer.blocks, layers)):
            a = AttnLayer()
            att = layer.attention.attention
            a.is_self = int(kind == 'self')
            a.q, a.k, a.v = _linear(att.proj_q, self._keep), _linear(att.proj_k), _linear(att.proj_v)
            if a.is_self:
                a.p = _linear(att.proj_p)
                w = torch.cat([att.proj_q.weight, att.proj_k.weight, att.proj_v.weight], 0).detach().contiguous()
                b = torch.cat([att.proj_q.bias, att.proj_k.bias, att.proj_v.bias], 0).detach().contiguous()
                a.qkv_w, a.qkv_b = _ptr(w), _ptr(b)
                if kernels.GEMM_PACKED:
                    self._keep.append(kernels.gemm_pack(w))
                    a.qkv_packed = self._keep[-1].data_ptr()
            else:
                w = torch.cat([att.proj_k.weight, att.proj_v.weight], 0).detach().contiguous()
                b = torch.cat([att.proj_k.bias, att.proj_v.bias], 0).detach().contiguous()
                a.kv_w, a.kv_b = _ptr(w), _ptr(b)
                if kernels.GEMM_PACKED:
                    self._keep.append(kernels.gemm_pack(w))
                    a.kv_packed = self._keep[-1].data_ptr()
            self._keep += [w, b]
            a.out, a.norm = _linear(layer.attention.linear, self._keep), _norm(layer.attention.norm)
            a.expand, a.squeeze = _linear(layer.output.expand, self._keep), _linear(layer.output.squeeze, self._keep)
            a.out_norm = _norm(layer.output.norm)
            t.layers[i] = a
        d.alpha = _ptr(m.optimal_transport.alpha)
        d.num_points_in_patch = m.num_points_in_patch
        d.num_correspondences = m.coarse_matching.num_correspondences
        d.num_sinkhorn_iterations = m.optimal_transport.num_iterations
        d.dual_normalization = int(m.coarse_matching.dual_normalization)
        f = m.fine_matching
        if f.use_global_score or f.correspondence_limit is not None:
            raise NotImplementedError('use_global_score / correspondence_limit are module-level options (geotr_lgr_ex); the native executor '
                                      'carries the settings of the reference configs -- run this model with use_native = False')
        d.topk, d.mutual, d.correspondence_threshold = f.k, int(f.mutual), f.correspondence_threshold
        d.num_refinement_steps = f.num_refinement_steps
        d.gemm_mode = kernels.gemm_mode()
        d.confidence_threshold, d.acceptance_radius = float(f.confidence_threshold), float(f.acceptance_radius)
        return d, self._keep

    def generation(self):
        """The current (descriptor, keep-alive tensors) generation, re-built when a parameter / buffer changed.

        Lanes share one model.  A re-build never frees memory a concurrent lane may still use: the new generation is built
        completely (and its producing stream synchronised) before it is published with one reference assignment; the old
        generation is dropped only after a device-wide synchronise that FOLLOWS the publication, so every kernel launched
        from it before that point has finished; a lane that launches from an old generation after that point finds itself
        outdated when its call returns and pins the tensors to its stream (`forward_batch`)."""
        key = self._version_key()
        if key != self._key:
            with self._lock:
                if key != self._key:
                    old = self._gen
                    gen = self._build()
                    torch.cuda.current_stream().synchronize()  # fused / packed weights are produced on this thread's stream
                    self._gen = gen
                    self._key = key
                    if old is not None:
                        torch.cuda.synchronize()
                    del old
        return self._gen

    def descriptor(self):
        return self.generation()[0]

    @staticmethod
    def pyramid(data_dict, lengths_host):
        p = Pyramid()
        S = len(data_dict['points'])
        p.num_stages = S
        p.num_pairs = len(lengths_host[0]) // 2
        order = data_dict.get('_order')  # grid order of each stage's rows (build_pyramid); absent for a reference-built dict
        for i in range(S):
            if order is not None:
                assert order[i].dtype == torch.int32 and order[i].is_contiguous() and order[i].numel() == data_dict['points'][i].shape[0]
                p.order[i] = order[i].data_ptr()
            pts, nb = data_dict['points'][i], data_dict['neighbors'][i]
            assert pts.is_contiguous() and nb.is_contiguous()
            p.points[i], p.n[i] = pts.data_ptr(), pts.shape[0]
            p.neighbors[i], p.neighbors_w[i] = nb.data_ptr(), nb.shape[1]
            assert len(lengths_host[i]) % 2 == 0 and len(lengths_host[i]) <= 2 * MAX_PAIRS, 'clouds come in (ref, src) pairs'
            for q, l in enumerate(lengths_host[i]):
                p.cloud_n[i][q] = int(l)
            if i < S - 1:
                sub, up = data_dict['subsampling'][i], data_dict['upsampling'][i]
                assert sub.is_contiguous() and up.is_contiguous()
                p.subsampling[i], p.subsampling_w[i] = sub.data_ptr(), sub.shape[1]
                p.upsampling[i], p.upsampling_w[i] = up.data_ptr(), up.shape[1]
        return p

    @torch.no_grad()
    def forward(self, data_dict):
        """Whole forward of one pair in one native call.  Returns the output dict (same keys as GeoTransformer.forward).
        The dict must hold exactly one (ref, src) pair: multi-pair stacks have their own layout and entry point
        (RegistrationPipeline.register_batch -> forward_batch)."""
        lengths0 = data_dict['lengths_host'][0] if 'lengths_host' in data_dict else data_dict['lengths'][0]
        if len(lengths0) != 2 or int(data_dict.get('batch_size', 1)) != 1:
            raise ValueError(f'GeoTransformer.forward registers ONE pair per call (got batch_size={data_dict.get("batch_size", 1)}, '
                             f'{len(lengths0)} clouds), as the reference model does (experiments/*/model.py:76-83 reads lengths[.][0] '
                             f'only); use RegistrationPipeline.register_batch for several pairs')
        return self.forward_batch(data_dict)[0]

    @torch.no_grad()
    def forward_batch(self, data_dict):
        """One native call for B stacked pairs (clouds ordered ref_0, src_0, ref_1, src_1, ...; B = len(lengths[0]) / 2):
        the KPConv-FPN runs once over the whole stack, the heads pair by pair.  Returns B output dicts."""
        lib = _bind()
        m = self.model
        gen = self.generation()
        desc = gen[0]
        lengths = data_dict.get('lengths_host')
        if lengths is None:
            lengths = [l.tolist() for l in data_dict['lengths']]
        pyr = self.pyramid(data_dict, lengths)
        B = int(pyr.num_pairs)
        S, fine = m.backbone.num_stages, m.backbone.fine_stage
        feats = data_dict['features']
        dev = feats.device
        n_c, n_f = int(pyr.n[S - 1]), int(pyr.n[fine])
        P, K, topk = int(desc.num_correspondences), int(desc.num_points_in_patch), int(desc.topk)
        D = int(desc.transformer.out_proj.out)
        c_f = int(desc.backbone.decoder[desc.backbone.num_decoders - 1].out)
        cap = P * K * topk
        f32, i64 = torch.float32, torch.int64
        # one allocation per output key for the whole stack; pair b uses slice b
        o = {
            'feats_c': torch.empty((n_c, D), dtype=f32, device=dev), 'feats_f': torch.empty((n_f, c_f), dtype=f32, device=dev),
            'ref_node_corr_indices': torch.empty((B, P), dtype=i64, device=dev), 'src_node_corr_indices': torch.empty((B, P), dtype=i64, device=dev),
            'node_corr_scores': torch.empty((B, P), dtype=f32, device=dev), 'num_node_corr': torch.empty((B, 1), dtype=torch.int32, device=dev),
            'ref_knn_indices': torch.empty((B, P, K), dtype=i64, device=dev), 'src_knn_indices': torch.empty((B, P, K), dtype=i64, device=dev),
            'ref_knn_masks': torch.empty((B, P, K), dtype=torch.bool, device=dev), 'src_knn_masks': torch.empty((B, P, K), dtype=torch.bool, device=dev),
            'ref_knn_points': torch.empty((B, P, K, 3), dtype=f32, device=dev), 'src_knn_points': torch.empty((B, P, K, 3), dtype=f32, device=dev),
            'matching_scores': torch.empty((B, P, K + 1, K + 1), dtype=f32, device=dev),
            'ref_corr_points': torch.empty((B, cap, 3), dtype=f32, device=dev), 'src_corr_points': torch.empty((B, cap, 3), dtype=f32, device=dev),
            'corr_scores': torch.empty((B, cap), dtype=f32, device=dev), 'num_corr': torch.empty((B, 1), dtype=torch.int32, device=dev),
            'estimated_transform': torch.empty((B, 4, 4), dtype=f32, device=dev),
        }
        off = [[0] for _ in range(S)]  # cloud row offsets per stage
        for i in range(S):
            for l in lengths[i]:
                off[i].append(off[i][-1] + int(l))
        outs = (Outputs * B)()
        for b in range(B):
            vals = {}
            for name, _ in Outputs._fields_:
                if name == 'feats_c':
                    vals[name] = o[name][off[S - 1][2 * b]:].data_ptr()
                elif name == 'feats_f':
                    vals[name] = o[name][off[fine][2 * b]:].data_ptr()
                else:
                    vals[name] = o[name][b].data_ptr()
            outs[b] = Outputs(*[vals[name] for name, _ in Outputs._fields_])
        nbytes = lib.geotr_model_workspace_bytes(ctypes.byref(desc), ctypes.byref(pyr))
        if nbytes == 0:
            raise RuntimeError('geotr_model_workspace_bytes failed: ' + lib.geotr_last_error().decode('utf-8', 'replace'))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _log_buffers('forward', [('ws', ws)] + list(o.items()))
        if _POISON:  # debugging aid: a kernel that reads workspace / output memory it never wrote then shows up as NaNs
            ws.fill_(0xFF)
            for t in o.values():
                t.view(torch.uint8).fill_(0xFF)
        rc = lib.geotr_model_forward(ctypes.byref(desc), ctypes.byref(pyr), feats.data_ptr(), outs, ws.data_ptr(), nbytes,
                                     torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'geotr_model_forward')
        ws.record_stream(torch.cuda.current_stream())
        if gen is not self._gen:  # the model was re-packed while this call was being issued: see generation()
            for t in gen[1]:
                t.record_stream(torch.cuda.current_stream())
        points_c, points_f, points = data_dict['points'][-1], data_dict['points'][fine], data_dict['points'][0]
        results = []
        for b in range(B):
            c0, c1, c2 = off[S - 1][2 * b], off[S - 1][2 * b + 1], off[S - 1][2 * b + 2]
            f0, f1, f2 = off[fine][2 * b], off[fine][2 * b + 1], off[fine][2 * b + 2]
            r0, r1, r2 = off[0][2 * b], off[0][2 * b + 1], off[0][2 * b + 2]
            results.append({
                'ref_points_c': points_c[c0:c1], 'src_points_c': points_c[c1:c2], 'ref_points_f': points_f[f0:f1],
                'src_points_f': points_f[f1:f2], 'ref_points': points[r0:r1], 'src_points': points[r1:r2],
                'ref_feats_c': o['feats_c'][c0:c1], 'src_feats_c': o['feats_c'][c1:c2],
                'ref_feats_f': o['feats_f'][f0:f1], 'src_feats_f': o['feats_f'][f1:f2],
                'matching_scores': o['matching_scores'][b], 'estimated_transform': o['estimated_transform'][b],
                'ref_node_corr_knn_points': o['ref_knn_points'][b], 'src_node_corr_knn_points': o['src_knn_points'][b],
                'ref_node_corr_knn_masks': o['ref_knn_masks'][b], 'src_node_corr_knn_masks': o['src_knn_masks'][b],
                # data-dependent lengths stay on the device; `finalize` trims them with one host read
                '_ref_node_corr_indices': o['ref_node_corr_indices'][b], '_src_node_corr_indices': o['src_node_corr_indices'][b],
                '_ref_corr_points': o['ref_corr_points'][b], '_src_corr_points': o['src_corr_points'][b], '_corr_scores': o['corr_scores'][b],
                '_counts': (o['num_node_corr'][b], o['num_corr'][b]),
                '_counts_stack': (o['num_node_corr'], o['num_corr'], b),
            })
        return results

    @staticmethod
    def raise_on_overflow(worst):
        """The fixed-capacity radius search keeps the first `capacity` candidates of a ball in cell-scan order, not the
        nearest ones: a table built from an overflowed ball is wrong, so this is an error, never a warning."""
        if int(worst) > 0:
            raise RuntimeError(f'radius search row capacity exceeded ({int(worst)} neighbours in one ball); '
                               f'rebuild the pipeline with exact_width=True for such dense clouds')

    @staticmethod
    def finalize_stack(outs, overflow=None):
        """finalize() for all pairs of one forward_batch call with ONE host<-device read (all counts at once).
        `overflow`: the pyramid's device flag (native.build_pyramid); it rides on the same read and raises when set."""
        if not outs:
            return outs
        num_node, num_corr, _ = outs[0]['_counts_stack']
        cols = [num_node, num_corr]
        if overflow is not None:
            cols.append(overflow.view(1, 1).expand(num_node.shape[0], 1))
        counts = torch.cat(cols, dim=1).tolist()  # (B, 2 or 3)
        if overflow is not None:
            NativeModel.raise_on_overflow(counts[0][2])
        return [NativeModel.finalize(o, counts=counts[o['_counts_stack'][2]]) for o in outs]

    @staticmethod
    def counts_to_host_async(outs, overflow):
        """The data-dependent counts of a forward_batch call (and the pyramid's overflow flag) on their way to pinned host memory: an
        asynchronous copy on the current stream; `.tolist()` the result once the stream has been synchronised past this point."""
        num_node, num_corr, _ = outs[0]['_counts_stack']
        dev = torch.cat([num_node, num_corr, overflow.view(1, 1).expand(num_node.shape[0], 1)], dim=1)  # (B, 3) int32
        host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        return host

    @staticmethod
    def finalize_stack_counts(outs, counts):
        """finalize_stack with the counts already on the host (counts_to_host_async(...).tolist())."""
        NativeModel.raise_on_overflow(counts[0][2])
        return [NativeModel.finalize(o, counts=counts[o['_counts_stack'][2]]) for o in outs]

    @staticmethod
    def finalize(out, counts=None, overflow=None):
        """Trim the variable-length outputs to their true sizes (the one host<-device read of a pair); the pyramid's
        overflow flag, when given, is read together with the counts and raises when set."""
        num_node, num_corr = out.pop('_counts')
        out.pop('_counts_stack', None)
        if counts is None:
            counts = torch.cat([num_node, num_corr] + ([overflow.view(1)] if overflow is not None else [])).tolist()
            if overflow is not None:
                NativeModel.raise_on_overflow(counts[2])
        p, c = int(counts[0]), int(counts[1])
        out['ref_node_corr_indices'] = out.pop('_ref_node_corr_indices')[:p]
        out['src_node_corr_indices'] = out.pop('_src_node_corr_indices')[:p]
        for k in ('ref_node_corr_knn_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks', 'src_node_corr_knn_masks',
                  'matching_scores'):
            out[k] = out[k][:p]
        out['ref_corr_points'] = out.pop('_ref_corr_points')[:c]
        out['src_corr_points'] = out.pop('_src_corr_points')[:c]
        out['corr_scores'] = out.pop('_corr_scores')[:c]
        return out


class KernelProfiler:
    """HIP-event timing of the two heaviest kernel families (GSE embedding, packed GEMMs) for launches made by the native executor.

    Events are created here, handed to the library as raw handles and recorded by the executor on the launch stream; the first
    `capacity` such launches after arming are recorded.  `results()` -> [(seconds, kind, work)]: kind 'gse' with work = number of
    (i, j) superpoint pairs of the launch, kind 'gemm' with work = (m, n, k, epilogue flags: 1 residual read, 2 gathered coarse rows, 4 statistics records), or kind 'kpconv' (a fused KPConv layer) with
    work = (m, c_out, 15 c_in, h), or kind 'radius' (a radius search of the pyramid) with work = (query stage, support stage, table width, 1 = dense kernel)."""
    GEMM_TAG = 1 << 62
    KPCONV_TAG = 1 << 61
    RADIUS_TAG = 1 << 60

    def __init__(self, capacity, stride=1):
        """`stride`: bracket every stride-th eligible launch only -- a timed event pair keeps its launch from overlapping its
        neighbours in the stream, so a sample spread over the whole region disturbs it far less than every launch of its start."""
        _bind()
        self.capacity = capacity
        self.stride = max(1, int(stride))
        self.start = [torch.cuda.Event(enable_timing=True) for _ in range(capacity)]
        self.stop = [torch.cuda.Event(enable_timing=True) for _ in range(capacity)]
        for e in self.start + self.stop:
            e.record()  # forces creation of the underlying hipEvent_t
        torch.cuda.synchronize()
        self._start = (ctypes.c_void_p * capacity)(*[e.cuda_event for e in self.start])
        self._stop = (ctypes.c_void_p * capacity)(*[e.cuda_event for e in self.stop])
        self._sizes = (ctypes.c_int64 * capacity)()

    def __enter__(self):
        _lib.check(_lib.load().geotr_profile_stride(self.stride), 'geotr_profile_stride')
        _lib.check(_lib.load().geotr_profile_gse(self._start, self._stop, self._sizes, self.capacity), 'geotr_profile_gse')
        return self

    def __exit__(self, *exc):
        self.used = int(_lib.load().geotr_profile_gse_count())
        _lib.load().geotr_profile_gse(None, None, None, 0)
        _lib.load().geotr_profile_stride(1)

    def results(self):
        """Every recorded launch; call after torch.cuda.synchronize()."""
        out = []
        for i in range(self.used):
            sec, tag = self.start[i].elapsed_time(self.stop[i]) * 1e-3, int(self._sizes[i])
            if tag & self.GEMM_TAG and tag > 0:
                out.append((sec, 'gemm', ((tag >> 26) & 0xffffff, (tag >> 14) & 0xfff, tag & 0x3fff, (tag >> 50) & 7)))  # (m, n, k, epilogue flags)
            elif tag & self.KPCONV_TAG and tag > 0:
                out.append((sec, 'kpconv', ((tag >> 26) & 0xffffff, (tag >> 14) & 0xfff, tag & 0x3fff, (tag >> 50) & 0x7ff)))
            elif tag & self.RADIUS_TAG and tag > 0:  # a radius search of the pyramid: (query stage, support stage, table width, dense kernel?)
                out.append((sec, 'radius', ((tag >> 56) & 0xf, (tag >> 52) & 0xf, (tag >> 20) & 0xfffff, (tag >> 19) & 1)))
            elif tag != 0:
                out.append((sec, 'gse', tag * tag if tag > 0 else -tag))
        return out


GseProfiler = KernelProfiler  # former name


class PyramidPlan:
    """A pyramid whose kernels are enqueued but whose stage sizes are still on their way to the host (build_pyramid_async).
    `finish()` turns it into the reference's dict once the stream has passed the enqueue point (the caller synchronises: an event
    recorded after build_pyramid_async, or any later synchronisation of that stream)."""

    def __init__(self, pts, lens, nb, sub, up, order, overflow, host, B, S, ws):
        self.pts, self.lens, self.nb, self.sub, self.up, self.order = pts, lens, nb, sub, up, order
        self.overflow, self.host, self.B, self.S, self.ws = overflow, host, B, S, ws

    def finish(self):
        B, S = self.B, self.S
        flat = self.host.tolist()  # pinned host memory, written by a kernel of the stream (complete: the caller synchronised)
        lengths_host = [[int(flat[i * B + b]) for b in range(B)] for i in range(S)]
        n = [sum(l) for l in lengths_host]
        if min(n) < 1:
            raise RuntimeError(f'pyramid: an empty stage (sizes {n}) -- finish() called before the stream was synchronised?')
        self.ws = None
        return {
            'points': [self.pts[i][: n[i]] for i in range(S)],
            'lengths': self.lens,
            'neighbors': [self.nb[i][: n[i]] for i in range(S)],
            'subsampling': [self.sub[i][: n[i + 1]] for i in range(S - 1)],
            'upsampling': [self.up[i][: n[i]] for i in range(S - 1)],
            'lengths_host': lengths_host,
            '_overflow': self.overflow,
            '_order': [self.order[i][: n[i]] for i in range(S)],  # visiting order of the gather kernels (grid order of each stage)
        }


def _pyramid_buffers(points, lengths, S, neighbor_limits):
    dev = points.device
    n0, B = points.shape[0], lengths.shape[0]
    pts = [points] + [torch.empty((n0, 3), dtype=torch.float32, device=dev) for _ in range(S - 1)]
    lens = [lengths] + [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(S - 1)]
    nb = [torch.empty((n0, neighbor_limits[i]), dtype=torch.int64, device=dev) for i in range(S)]
    sub = [torch.empty((n0, neighbor_limits[i]), dtype=torch.int64, device=dev) for i in range(S - 1)]
    up = [torch.empty((n0, neighbor_limits[i + 1]), dtype=torch.int64, device=dev) for i in range(S - 1)]
    order = [torch.empty(n0, dtype=torch.int32, device=dev) for _ in range(S)]
    buf = PyramidBuffers()
    for i in range(S):
        buf.points[i], buf.lengths[i], buf.neighbors[i] = pts[i].data_ptr(), lens[i].data_ptr(), nb[i].data_ptr()
        buf.order[i] = order[i].data_ptr()
        if i < S - 1:
            buf.subsampling[i], buf.upsampling[i] = sub[i].data_ptr(), up[i].data_ptr()
    return pts, lens, nb, sub, up, order, buf


@torch.no_grad()
def build_pyramid_async(points, lengths, num_stages, voxel_size, radius, neighbor_limits):
    """The pyramid of `build_pyramid` enqueued on the current stream WITHOUT any host synchronisation (geotr_pyramid_build_async: every
    launch is sized from the row capacity, the stage sizes stay on the device and travel to pinned host memory by a kernel of the
    stream).  Returns a PyramidPlan; call `.finish()` after the stream has been synchronised past this point."""
    lib = _lib.load()
    assert points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()
    dev = points.device
    n0, B, S = points.shape[0], lengths.shape[0], int(num_stages)
    limits = (ctypes.c_int64 * S)(*[int(x) for x in neighbor_limits])
    pts, lens, nb, sub, up, order, buf = _pyramid_buffers(points, lengths, S, neighbor_limits)
    host = torch.zeros(S * B, dtype=torch.int64).pin_memory()
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = lib.geotr_pyramid_workspace_bytes(n0, B, S)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _log_buffers('pyramid', [('ws', ws), ('input', points)] + [(f'points{i}', t) for i, t in enumerate(pts[1:], 1)])
    if _POISON:
        ws.fill_(0xFF)
    rc = lib.geotr_pyramid_build_async(points.data_ptr(), lengths.data_ptr(), B, n0, S, float(voxel_size), float(radius), limits,
                                       ctypes.byref(buf), host.data_ptr(), overflow.data_ptr(), ws.data_ptr(), nbytes,
                                       torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'geotr_pyramid_build_async')
    ws.record_stream(torch.cuda.current_stream())
    return PyramidPlan(pts, lens, nb, sub, up, order, overflow, host, B, S, ws)


@torch.no_grad()
def build_pyramid(points, lengths, num_stages, voxel_size, radius, neighbor_limits):
    """precompute_data_stack_mode in one native call (fixed-width neighbour tables).  Device tensors in; returns the
    reference's dict (lists of device tensors) plus 'lengths_host' (python ints).  One host read at the end (the stage sizes);
    `build_pyramid_async` is the variant without any."""
    lib = _lib.load()
    assert points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()
    dev = points.device
    n0, B, S = points.shape[0], lengths.shape[0], int(num_stages)
    limits = (ctypes.c_int64 * S)(*[int(x) for x in neighbor_limits])
    pts, lens, nb, sub, up, order, buf = _pyramid_buffers(points, lengths, S, neighbor_limits)
    host = (ctypes.c_int64 * (S * B))()
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = lib.geotr_pyramid_workspace_bytes(n0, B, S)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _log_buffers('pyramid', [('ws', ws), ('input', points)] + [(f'points{i}', t) for i, t in enumerate(pts[1:], 1)] +
                 [(f'neighbors{i}', t) for i, t in enumerate(nb)] + [(f'subsampling{i}', t) for i, t in enumerate(sub)] +
                 [(f'upsampling{i}', t) for i, t in enumerate(up)] + [(f'order{i}', t) for i, t in enumerate(order)])
    if _POISON:
        ws.fill_(0xFF)
    rc = lib.geotr_pyramid_build(points.data_ptr(), lengths.data_ptr(), B, n0, S, float(voxel_size), float(radius), limits,
                                 ctypes.byref(buf), host, overflow.data_ptr(), ws.data_ptr(), nbytes,
                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'geotr_pyramid_build')
    ws.record_stream(torch.cuda.current_stream())
    plan = PyramidPlan(pts, lens, nb, sub, up, order, overflow, None, B, S, None)
    plan.host = torch.tensor(list(host), dtype=torch.int64)
    return plan.finish()
