// executor.hip -- native (host C++) executor of the whole inference forward (geotr_model_forward, include/geotr.h).
//
// Same kernels and the same arithmetic as the Python module mirror (geotransformer_amd/{backbone,model}.py and modules/*):
// experiments/<exp>/model.py:69-212 -> backbone.py -> modules/kpconv/modules.py, modules/geotransformer/*, modules/transformer/*,
// modules/sinkhorn, local_global_registration.  Why it exists: one pair used to be ~420 kernel launches; issued from Python they
// cost ~5 ms of interpreter time per pair, more than the kernels need, and on a 256-CU part a single small pair is launch-latency
// bound anyway.  Here one asynchronous host call runs a STACK of up to 16 pairs: the KPConv-FPN once over all stacked points
// (GroupNorm statistics segmented per pair), the geometric transformer once over all superpoints (row-wise ops one launch per
// layer, attention cores as ragged grouped launches), the matching heads once per stack with blockIdx.y = pair / cloud.  The
// intermediates live in a bump allocator over the caller's workspace, the data-dependent counts stay on the device.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace geotr {

// Optional timing of the dominant kernel: bench.py arms a pool of HIP events (geotr_profile_gse); every gse_embed launch
// made by the executor is then bracketed by hipEventRecord on the launch stream.  Disarmed (cap = 0) it costs nothing.
static void** g_prof_start = nullptr;
static void** g_prof_stop = nullptr;
static int64_t* g_prof_size = nullptr;
static int g_prof_cap = 0;
static std::atomic<int> g_prof_used{0};
static std::atomic<int> g_prof_seq{0};  // eligible launches seen since arming
static int g_prof_stride = 1;           // every g_prof_stride-th eligible launch is bracketed (geotr_profile_stride)

struct Ctx {
  char* base;
  size_t off = 0, peak = 0, cap;
  hipStream_t stream;
  bool dry;  // size-query pass: account for allocations, launch nothing
  int rc = GEOTR_OK;
  int gemm_mode = 0;  // geotr_model.gemm_mode: arithmetic of the packed GEMMs / fused KPConv -- 0 split-bf16, 1 plain bf16, 2 exact fp32
  int nseg = 1;                                             // stacked pairs: GroupNorm statistics stay inside a pair
  int64_t seg_rows[GEOTR_MAX_STAGES][GEOTR_MAX_PAIRS] = {};  // rows of pair b at stage s (ref + src)
  const int32_t* order[GEOTR_MAX_STAGES] = {};               // grid order of each stage's rows (geotr_pyramid.order; null = row order)

  template <typename T>
  T* alloc(size_t count) {
    off = align_up(off);
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    if (off > peak) peak = off;
    if (!dry && off > cap && rc == GEOTR_OK) rc = fail(GEOTR_E_WORKSPACE, "model_forward: workspace exhausted (%zu > %zu)", off, cap);
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  bool live() const { return !dry && rc == GEOTR_OK; }
  void check(int code) {
    if (code != GEOTR_OK && rc == GEOTR_OK) rc = code;
  }
};

// packed split-bf16 path: tall activations, 16-byte aligned rows, K a multiple of 32 (geotr_gemm_packed's contract)
static inline bool use_packed(const void* packed, const float* a, int64_t lda, int64_t m, int64_t k) {
  return packed && m >= GEOTR_PACKED_MIN_ROWS && k % 32 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
}

// Event bracket of one launch for bench.py's live roofline (armed by geotr_profile_gse; free when disarmed).  `tag` goes into the
// size slot: n > 0 = a per-cloud GSE launch over n superpoints; -pairs = a ragged GSE launch over that many (i, j) pairs;
// kProfGemm | flags << 50 | m << 26 | n << 14 | k = a packed GEMM of that shape (m < 2^24, n < 2^12, k < 2^14; larger shapes are not
// recorded); flags = what its epilogue moves besides A, W and C: 1 a residual tensor, 2 gathered rows of a coarser product (split
// decoder), 4 GroupNorm statistics records (bench.py gemm_bytes counts them as algorithmic bytes of the launch).
constexpr int64_t kProfGemm = 1ll << 62;
constexpr int64_t kProfResidual = 1ll << 50, kProfGather = 2ll << 50, kProfStats = 4ll << 50;
constexpr int64_t kProfRadius = 1ll << 60;  // kProfRadius | query stage << 56 | support stage << 52 | table width << 20 | dense << 19 = a radius search of the pyramid
constexpr int64_t kProfKpconv = 1ll << 61;  // kProfKpconv | h << 50 | m << 26 | c_out << 14 | 15 c_in = a fused KPConv layer (kpconv_fused.hip)
struct ProfScope {
  int slot = -1;
  hipStream_t stream;
  explicit ProfScope(hipStream_t st) : stream(st) {
    if (g_prof_cap > 0 && g_prof_seq.fetch_add(1) % g_prof_stride == 0) {
      slot = g_prof_used.fetch_add(1);
      if (slot >= g_prof_cap) slot = -1;
    }
    if (slot >= 0) (void)hipEventRecord((hipEvent_t)g_prof_start[slot], stream);
  }
  void done(int64_t tag) {
    if (slot < 0) return;
    (void)hipEventRecord((hipEvent_t)g_prof_stop[slot], stream);
    g_prof_size[slot] = tag;
  }
};

// packed GEMM in the model's precision: split-bf16 (default) or plain bf16 operands
// (narrow, deep launches are split over K: the partial tiles live in the bump allocator for the duration of the call)
static int packed_gemm(Ctx& c, const float* a, int64_t lda, const void* packed, float* out, int64_t ldc, int64_t m, int64_t n, int64_t k,
                       const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act, hipStream_t stream) {
  const size_t mk = c.mark();
  const size_t sk_bytes = geotr_gemm_packed_splitk_workspace_bytes_mode(m, n, k, c.gemm_mode);
  char* sk = sk_bytes ? c.alloc<char>(sk_bytes) : nullptr;
  c.release(mk);  // stream order keeps the scratch valid until the reduce kernel has run: later allocations are written by later launches
  if (!c.live()) return GEOTR_OK;
  ProfScope prof(stream);
  const int rc = geotr_gemm_packed_splitk(a, lda, packed, out, ldc, m, n, k, bias, row_div, residual, ldr, alpha, act, c.gemm_mode, sk,
                                          sk_bytes, stream);
  if (m < (1 << 24) && n < (1 << 12) && k < (1 << 14)) prof.done(kProfGemm | (residual ? kProfResidual : 0) | (m << 26) | (n << 14) | k);
  else prof.done(0);
  return rc;
}

static float* linear(Ctx& c, const geotr_linear& l, const float* x, int64_t lda, int64_t m, int act, const float* residual = nullptr,
                     int64_t ldr = 0) {
  float* y = c.alloc<float>((size_t)m * l.out);
  // (packed_gemm also runs in the size-query pass: it accounts for its split-K scratch and launches only when live)
  if (use_packed(l.packed, x, lda, m, l.in))
    c.check(packed_gemm(c, x, lda, l.packed, y, l.out, m, l.out, l.in, l.b, nullptr, residual, ldr, 1.0f, act, c.stream));
  else if (c.live())
    c.check(geotr_gemm(x, lda, l.w, l.in, 0, y, l.out, m, l.out, l.in, 1, 0, 0, 0, l.b, nullptr, residual, ldr, 1.0f, act, c.stream));
  return y;
}

// GroupNorm statistics out of the producing GEMM's epilogue (round 3; geotr_gemm_packed_stats): the records of one Linear output
struct GnStats {
  const float* rec = nullptr;  // null: not produced (shape not on the packed path, K-split launch, switch off) -> the norm computes its own
  int64_t rpr = 0;             // rows per record
};
static const bool gn_epilogue_stats = [] {
  const char* e = std::getenv("GEOTR_GN_EPILOGUE_STATS");  // A/B switch for measurements: 0 = every GroupNorm runs its own statistics pass
  return !(e && e[0] == '0');
}();
// y = x W^T + b for a Linear whose output feeds a GroupNorm over the row segments of `stage`: the statistics records ride in `st`
static float* linear_gn(Ctx& c, const geotr_linear& l, const float* x, int64_t lda, int64_t m, int stage, GnStats& st) {
  st = GnStats();
  // unsplit packed launches only: a launch that is split over K for occupancy keeps its split (the statistics pass of such a narrow
  // output is small) -- judged by the split-bf16 rule (mode 0) in every arithmetic mode, so that which norms take their statistics from
  // the epilogue does not depend on the mode (the exact-fp32 plan splits more shapes)
  if (!(gn_epilogue_stats && use_packed(l.packed, x, lda, m, l.in) && geotr_gemm_packed_splits(m, l.out, l.in, 0) == 1))
    return linear(c, l, x, lda, m, 0);
  float* y = c.alloc<float>((size_t)m * l.out);
  float* rec = c.alloc<float>(geotr_gemm_packed_stats_floats(c.seg_rows[stage], c.nseg, l.out));
  if (c.live()) {
    ProfScope prof(c.stream);
    c.check(geotr_gemm_packed_stats(x, lda, l.packed, y, l.out, m, l.out, l.in, l.b, nullptr, 0, c.gemm_mode, c.seg_rows[stage], c.nseg, rec,
                                    c.stream));
    if (m < (1 << 24) && l.out < (1 << 12) && l.in < (1 << 14)) prof.done(kProfGemm | kProfStats | (m << 26) | (l.out << 14) | l.in);
    else prof.done(0);
  }
  st.rec = rec;
  st.rpr = geotr_gemm_packed_stats_rows_per_record(l.out);
  return y;
}

// GroupNorm (groups > 0) or LayerNorm (groups == 0) with optional residual / activation; returns a new (n, ch) buffer
// `stage` selects the pair segments of the row set (GroupNorm only)
// `row_flags` (GroupNorm only, optional): receives (row sum of the output > 0) per row -- what the KPConv fed by this output
// needs for its neighbour count -- when the width allows it (geotr_group_norm_flags_supported), else it is left untouched and
// *row_flags_done stays false
// `st` (GroupNorm only, optional): x's statistics records from its producing GEMM (linear_gn)
static float* norm(Ctx& c, const geotr_norm& nm, const float* x, int64_t n, int64_t ch, const float* residual, int act, int stage = 0,
                   float* y = nullptr, uint8_t* row_flags = nullptr, bool* row_flags_done = nullptr, const GnStats* st = nullptr) {
  if (!y) y = c.alloc<float>((size_t)n * ch);
  if (nm.groups > 0) {
    const size_t m = c.mark();
    double* ws = reinterpret_cast<double*>(c.alloc<char>(geotr_group_norm_workspace_bytes(n, ch)));
    const bool flags = row_flags && geotr_group_norm_flags_supported(ch);
    if (row_flags_done) *row_flags_done = flags;
    if (c.live()) {
      if (st && st->rec)
        c.check(geotr_group_norm_stats(x, n, ch, nm.groups, nm.gamma, nm.beta, nm.eps, st->rec, st->rpr, residual, nullptr, 0, 0, nullptr, nullptr,
                                       0.f, act, y, c.seg_rows[stage], c.nseg, ws, flags ? row_flags : nullptr, c.stream));
      else
        c.check(geotr_group_norm_segmented_flags(x, n, ch, nm.groups, nm.gamma, nm.beta, nm.eps, residual, act, y, c.seg_rows[stage], c.nseg, ws,
                                                 flags ? row_flags : nullptr, c.stream));
    }
    c.release(m);
  } else {
    if (c.live()) c.check(geotr_layer_norm(x, residual, n, ch, nm.gamma, nm.beta, nm.eps, y, c.stream));
  }
  return y;
}

// `s_flags` (optional): (row sum > 0) of s_feats, already produced by the GroupNorm that wrote them; else computed here
// `order`: visiting order of the m query rows (the grid order of their stage) or null
static float* kpconv(Ctx& c, const geotr_kpconv& kp, const float* s_feats, int64_t ns, const float* q_pts, int64_t m,
                     const float* s_pts, const int64_t* nb, int64_t h, const int32_t* order, const uint8_t* s_flags = nullptr) {
  float* out = c.alloc<float>((size_t)m * kp.out);
  const size_t mk = c.mark();
  const int64_t kdim = kp.num_kernel_points * kp.in;
  const uint8_t* flag = s_flags;
  if (kp.in > 1 && !flag) {
    uint8_t* f = c.alloc<uint8_t>((size_t)ns);
    if (c.live()) c.check(geotr_row_positive(s_feats, ns, kp.in, f, c.stream));
    flag = f;
  }
  // one kernel for the whole layer where the shape allows it: the (m, 15 c_in) operand stays in LDS (kpconv_fused.hip)
  static const bool fused_enabled = [] {
    const char* e = std::getenv("GEOTR_KPCONV_FUSED");  // A/B switch for measurements: GEOTR_KPCONV_FUSED=0 keeps the two-kernel path
    return !(e && e[0] == '0');
  }();
  if (fused_enabled && kp.in == 1 && kp.num_kernel_points == 15 && h <= 64) {  // first layer: exact fp32, bitwise the two-kernel result
    if (c.live())
      c.check(geotr_kpconv_c1_fused(s_feats, q_pts, s_pts, nb, kp.kernel_points, m, ns, h, kp.out, kp.num_kernel_points, kp.sigma, kp.weights,
                                    kp.bias, order, out, c.stream));
    c.release(mk);
    return out;
  }
  // (C_in <= 64 only: the deep layers are matrix-bound and lost 3.7 % fused -- profiles/r05_ab_runs.md section 3; that form was removed in round 6)
  if (fused_enabled && kp.packed && flag && kp.num_kernel_points == 15 && geotr_kpconv_fused_supported(kp.in, kp.out, h) &&
      (reinterpret_cast<uintptr_t>(s_feats) & 15) == 0) {
    if (c.live()) {
      ProfScope prof(c.stream);
      c.check(geotr_kpconv_fused(s_feats, q_pts, s_pts, nb, kp.kernel_points, flag, m, ns, h, kp.in, kp.out, kp.num_kernel_points, kp.sigma,
                                 kp.packed, kp.bias, c.gemm_mode, order, out, c.stream));
      prof.done(kProfKpconv | (h << 50) | (m << 26) | (kp.out << 14) | kdim);
    }
    c.release(mk);
    return out;
  }
  // two-kernel path (the deep layers, and shapes the fused kernel does not take): the (m, 15 c_in) operand is written once and read by the GEMM.  (Bounding
  // it to an Infinity-Cache-sized chunk per round was measured not to pay -- profiles/r02_ab_runs.md -- and the switch is gone.)
  float* weighted = c.alloc<float>((size_t)m * kdim);
  int32_t* nnum = c.alloc<int32_t>((size_t)m);
  if (c.live())
    c.check(geotr_kpconv_gather(s_feats, q_pts, s_pts, nb, kp.kernel_points, flag, m, ns, h, kp.in, kp.num_kernel_points, kp.sigma, weighted, nnum,
                                c.stream));
  if (use_packed(kp.packed, weighted, kdim, m, kdim))
    c.check(packed_gemm(c, weighted, kdim, kp.packed, out, kp.out, m, kp.out, kdim, kp.bias, nnum, nullptr, 0, 1.0f, 0, c.stream));
  else if (c.live())
    c.check(geotr_gemm(weighted, kdim, kp.weights, kp.out, 1, out, kp.out, m, kp.out, kdim, 1, 0, 0, 0, kp.bias, nnum, nullptr, 0, 1.0f, 0, c.stream));
  c.release(mk);
  return out;
}

// ConvBlock / ResidualBlock (modules/kpconv/modules.py:105-225); returns (m, out) features
// s_stage / q_stage: pyramid stages of the support rows (ns) and the query rows (m)
static float* block(Ctx& c, const geotr_block& b, const float* s_feats, int64_t ns, const float* q_pts, int64_t m, const float* s_pts,
                    const int64_t* nb, int64_t h, int s_stage, int q_stage) {
  if (b.is_conv_block) {
    float* x = kpconv(c, b.conv, s_feats, ns, q_pts, m, s_pts, nb, h, c.order[q_stage]);
    return norm(c, b.conv_norm, x, m, b.conv.out, nullptr, 2, q_stage);
  }
  const float* x = s_feats;
  const uint8_t* x_flags = nullptr;
  if (b.has_unary1) {
    GnStats st1;
    float* t = b.unary1_norm.groups > 0 ? linear_gn(c, b.unary1, s_feats, b.unary1.in, ns, s_stage, st1) : linear(c, b.unary1, s_feats, b.unary1.in, ns, 0);
    uint8_t* f = c.alloc<uint8_t>((size_t)ns);
    bool done = false;
    x = norm(c, b.unary1_norm, t, ns, b.unary1.out, nullptr, 2, s_stage, nullptr, f, &done, &st1);
    if (done) x_flags = f;  // the KPConv below needs no separate pass over its input
  }
  float* y = kpconv(c, b.conv, x, ns, q_pts, m, s_pts, nb, h, c.order[q_stage], x_flags);
  y = norm(c, b.conv_norm, y, m, b.conv.out, nullptr, 2, q_stage);
  const float* sc = s_feats;  // shortcut branch
  const int64_t in_ch = b.has_unary1 ? b.unary1.in : b.conv.in;
  if (b.strided) {
    float* pooled = c.alloc<float>((size_t)m * in_ch);
    if (c.live()) c.check(geotr_maxpool_ordered(s_feats, nb, m, ns, h, in_ch, c.order[q_stage], pooled, c.stream));
    sc = pooled;
  }
  static const bool fuse_shortcut_norm = [] {
    const char* e = std::getenv("GEOTR_GN_SHORTCUT_FUSED");  // A/B switch for measurements: 0 = normalise the shortcut in its own pass
    return !(e && e[0] == '0');
  }();
  // (The block's tail without an apply pass -- each small-K product launched twice, once for its statistics and once with the finalised
  // scale / shift in its epilogue -- was built in round 3, measured 2-3 % slower than the path below and removed from the executor in
  // round 5; its entry points geotr_gemm_packed_tail / geotr_group_norm_finalize stay in the ABI with their bit-identity test.)
  if (b.has_shortcut) {
    GnStats st_sc;
    float* t = b.shortcut_norm.groups > 0 ? linear_gn(c, b.shortcut, sc, b.shortcut.in, m, q_stage, st_sc) : linear(c, b.shortcut, sc, b.shortcut.in, m, 0);
    if (fuse_shortcut_norm && b.shortcut_norm.groups > 0 && b.unary2_norm.groups > 0 && b.shortcut.out == b.unary2.out) {
      // leaky_relu(GN(unary2(x)) + GN(shortcut Linear)): the shortcut's affine rides in the apply pass of the main branch (bit-identical);
      // its normalised (m, out) tensor -- 328 MB per 16-pair stack at stage 0 -- is neither written nor re-read
      GnStats st_z;
      float* z = linear_gn(c, b.unary2, y, b.unary2.in, m, q_stage, st_z);
      float* out = c.alloc<float>((size_t)m * b.unary2.out);
      const size_t mk = c.mark();
      double* ws = reinterpret_cast<double*>(c.alloc<char>(geotr_group_norm_workspace_bytes(m, b.unary2.out)));
      if (c.live())
        c.check(geotr_group_norm_stats(z, m, b.unary2.out, b.unary2_norm.groups, b.unary2_norm.gamma, b.unary2_norm.beta, b.unary2_norm.eps, st_z.rec,
                                       st_z.rpr, t, st_sc.rec, st_sc.rpr, b.shortcut_norm.groups, b.shortcut_norm.gamma, b.shortcut_norm.beta,
                                       b.shortcut_norm.eps, 2, out, c.seg_rows[q_stage], c.nseg, ws, nullptr, c.stream));
      c.release(mk);
      return out;
    }
    sc = norm(c, b.shortcut_norm, t, m, b.shortcut.out, nullptr, 0, q_stage, nullptr, nullptr, nullptr, &st_sc);
  }
  GnStats st_z;
  float* z = b.unary2_norm.groups > 0 ? linear_gn(c, b.unary2, y, b.unary2.in, m, q_stage, st_z) : linear(c, b.unary2, y, b.unary2.in, m, 0);
  return norm(c, b.unary2_norm, z, m, b.unary2.out, sc, 2, q_stage, nullptr, nullptr, nullptr, &st_z);  // leaky_relu(unary2(x) + shortcut)
}

struct BackboneOut {
  const float* feats_c;
  int64_t c_dim;
};

static BackboneOut backbone_forward(Ctx& c, const geotr_backbone& net, const geotr_pyramid& p, const float* feats, float* feats_f_out) {
  const int S = net.num_stages;
  const float* enc[GEOTR_MAX_STAGES];
  int64_t enc_ch[GEOTR_MAX_STAGES];
  const float* x = block(c, net.blocks[0], feats, p.n[0], p.points[0], p.n[0], p.points[0], p.neighbors[0], p.neighbors_w[0], 0, 0);
  x = block(c, net.blocks[1], x, p.n[0], p.points[0], p.n[0], p.points[0], p.neighbors[0], p.neighbors_w[0], 0, 0);
  enc[0] = x;
  enc_ch[0] = net.blocks[1].unary2.out;
  int bi = 2;
  for (int s = 1; s < S; ++s) {
    x = block(c, net.blocks[bi++], x, p.n[s - 1], p.points[s], p.n[s], p.points[s - 1], p.subsampling[s - 1], p.subsampling_w[s - 1], s - 1, s);
    x = block(c, net.blocks[bi++], x, p.n[s], p.points[s], p.n[s], p.points[s], p.neighbors[s], p.neighbors_w[s], s, s);
    x = block(c, net.blocks[bi++], x, p.n[s], p.points[s], p.n[s], p.points[s], p.neighbors[s], p.neighbors_w[s], s, s);
    enc[s] = x;
    enc_ch[s] = net.blocks[bi - 1].unary2.out;
  }
  const float* latent = enc[S - 1];
  int64_t lat_ch = enc_ch[S - 1];
  int d = 0;
  static const bool decoder_split = [] {
    const char* e = std::getenv("GEOTR_DECODER_SPLIT");  // A/B switch for measurements: 0 = concatenate, then one Linear (the reference's form)
    return !(e && e[0] == '0');
  }();
  for (int i = S - 2; i >= net.fine_stage; --i, ++d) {
    const int64_t tot = lat_ch + enc_ch[i];
    const geotr_linear& l = net.decoder[d];
    const bool last = i == net.fine_stage;  // LastUnaryBlock: straight into the caller's buffer, no norm
    const bool gn = !last && net.decoder_norm[d].groups > 0;
    // Linear(cat(up(latent), skip)) = up(latent W_latent^T) + skip W_skip^T + b  (round 3): the coarse-level product is gathered into the
    // fine-level GEMM's epilogue, so the (rows, latent + skip channels) concatenation -- 553 MB per 16-pair stack at the fine level of the
    // 3DMatch model, written once and read once -- never exists, and the fine-level contraction is over the skip channels only.
    // Both products must be on the packed path (>= GEOTR_PACKED_MIN_ROWS rows each), else the reference's form below.
    if (decoder_split && net.decoder_packed_latent[d] && net.decoder_packed_skip[d] && l.in == tot &&
        use_packed(net.decoder_packed_latent[d], latent, lat_ch, p.n[i + 1], lat_ch) &&
        use_packed(net.decoder_packed_skip[d], enc[i], enc_ch[i], p.n[i], enc_ch[i])) {
      float* coarse = c.alloc<float>((size_t)p.n[i + 1] * l.out);
      c.check(packed_gemm(c, latent, lat_ch, net.decoder_packed_latent[d], coarse, l.out, p.n[i + 1], l.out, lat_ch, nullptr, nullptr, nullptr, 0,
                          1.0f, 0, c.stream));
      float* t = last ? feats_f_out : c.alloc<float>((size_t)p.n[i] * l.out);
      GnStats st_d;
      float* rec = nullptr;
      if (gn && gn_epilogue_stats) {
        rec = c.alloc<float>(geotr_gemm_packed_stats_floats(c.seg_rows[i], c.nseg, l.out));
        st_d.rec = rec, st_d.rpr = geotr_gemm_packed_stats_rows_per_record(l.out);
      }
      if (c.live()) {
        ProfScope prof(c.stream);
        c.check(geotr_gemm_packed_gather(enc[i], enc_ch[i], net.decoder_packed_skip[d], t, l.out, p.n[i], l.out, enc_ch[i], l.b, 0,
                                         c.gemm_mode, coarse, l.out, p.n[i + 1], p.upsampling[i], p.upsampling_w[i], c.seg_rows[i], c.nseg, rec,
                                         c.stream));
        if (p.n[i] < (1 << 24) && l.out < (1 << 12) && enc_ch[i] < (1 << 14)) prof.done(kProfGemm | kProfGather | (rec ? kProfStats : 0) | (p.n[i] << 26) | (l.out << 14) | enc_ch[i]);
        else prof.done(0);
      }
      latent = last ? t : norm(c, net.decoder_norm[d], t, p.n[i], l.out, nullptr, 2, i, nullptr, nullptr, nullptr, &st_d);
      lat_ch = l.out;
      continue;
    }
    float* cat = c.alloc<float>((size_t)p.n[i] * tot);
    if (c.live())
      c.check(geotr_upsample_concat(latent, p.n[i + 1], lat_ch, p.upsampling[i], p.upsampling_w[i], enc[i], enc_ch[i], p.n[i], cat, c.stream));
    if (last) {  // LastUnaryBlock: straight into the caller's buffer
      if (use_packed(l.packed, cat, tot, p.n[i], l.in))
        c.check(packed_gemm(c, cat, tot, l.packed, feats_f_out, l.out, p.n[i], l.out, l.in, l.b, nullptr, nullptr, 0, 1.0f, 0, c.stream));
      else if (c.live())
        c.check(geotr_gemm(cat, tot, l.w, l.in, 0, feats_f_out, l.out, p.n[i], l.out, l.in, 1, 0, 0, 0, l.b, nullptr, nullptr, 0, 1.0f, 0,
                           c.stream));
      latent = feats_f_out;
    } else {
      GnStats st_d;
      float* t = net.decoder_norm[d].groups > 0 ? linear_gn(c, l, cat, tot, p.n[i], i, st_d) : linear(c, l, cat, tot, p.n[i], 0);
      latent = norm(c, net.decoder_norm[d], t, p.n[i], l.out, nullptr, 2, i, nullptr, nullptr, nullptr, &st_d);
    }
    lat_ch = l.out;
  }
  return {enc[S - 1], enc_ch[S - 1]};
}

static float* attn_tail(Ctx& c, const geotr_attn_layer& L, const float* hidden, const float* x, int64_t n, int64_t C, float* out = nullptr) {
  float* h2 = linear(c, L.out, hidden, C, n, 0);
  float* y = norm(c, L.norm, h2, n, C, x, 0);                       // LN(linear(attn) + x)
  float* e = linear(c, L.expand, y, C, n, 1);                       // ReLU fused
  float* s = linear(c, L.squeeze, e, L.expand.out, n, 0);
  return norm(c, L.out_norm, s, n, C, y, 0, 0, out);                // LN(y + W2 relu(W1 y))
}

// `shared_ws` (optional): split-weight workspace shared by all clouds of a stack; `first` = this call fills it
static float* gse(Ctx& c, const geotr_transformer& t, const float* pts, int64_t n, char* shared_ws = nullptr, bool first = true) {
  const int64_t D = t.proj_d.out;
  float* emb = c.alloc<float>((size_t)n * n * D);
  int32_t* knn = c.alloc<int32_t>((size_t)n * t.angle_k);
  const size_t gws_bytes = geotr_gse_embed_workspace_bytes(D, t.gse_precision);
  char* gws = shared_ws ? shared_ws : c.alloc<char>(gws_bytes + 16);
  const int precision = ((t.gse_precision == 1 || t.gse_precision == 3) && shared_ws && !first) ? t.gse_precision + 1 : t.gse_precision;
  if (c.live()) {
    c.check(geotr_gse_knn(pts, n, t.angle_k, knn, c.stream));
    ProfScope prof(c.stream);
    c.check(geotr_gse_embed(pts, knn, n, t.angle_k, D, t.div_term, t.proj_d.w, t.proj_d.b, t.proj_a.w, t.proj_a.b, t.sigma_d, t.sigma_a,
                            precision, gws, gws_bytes, emb, c.stream));
    prof.done(n);
  }
  return emb;
}

// ---- geometric transformer over a stack of pairs (model.py:133-145, conditional_transformer.py:97-117) --------------------
// Working layout "refs first": rows [ref_0 .. ref_{B-1} | src_0 .. src_{B-1}], so that every row-wise op (projections, output
// linear, LayerNorm, FFN) is ONE launch over all clouds (self layers) or over all reference / all source clouds (the two
// sequential halves of a cross layer); only the attention core runs per cloud.
struct RowMove {
  int n;
  int64_t src0[2 * GEOTR_MAX_PAIRS], dst0[2 * GEOTR_MAX_PAIRS], rows[2 * GEOTR_MAX_PAIRS];
};
__global__ __launch_bounds__(256) void move_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int C4, RowMove mv) {
  const int seg = blockIdx.y;
  const float4* s = reinterpret_cast<const float4*>(src) + mv.src0[seg] * C4;
  float4* d = reinterpret_cast<float4*>(dst) + mv.dst0[seg] * C4;
  const int64_t total = mv.rows[seg] * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) d[e] = s[e];
}
static void move_rows(Ctx& c, const float* src, float* dst, int64_t C, const RowMove& mv) {
  if (!c.live()) return;
  int64_t mx = 0;
  for (int i = 0; i < mv.n; ++i) mx = std::max(mx, mv.rows[i]);
  const unsigned gx = (unsigned)std::min<int64_t>((mx * (C / 4) + 255) / 256, 64);
  move_rows_kernel<<<dim3(gx, (unsigned)mv.n), dim3(256), 0, c.stream>>>(src, dst, (int)(C / 4), mv);
  if (hipGetLastError() != hipSuccess) c.check(fail(GEOTR_E_LAUNCH, "model_forward: move_rows launch failed"));
}

// qt[:, h, :] = q_h W_p[h], qb[:, h] = q_h . b_p[h] for the query rows of ALL groups at once: proj_p collapsed into the queries
// (rpe_transformer.py:54-56)
static void positional_queries(Ctx& c, int H, int64_t C, const float* q, int64_t ldq, int64_t rows, const geotr_linear* proj_p, float* qt,
                               float* qb) {
  const int64_t ch = C / H;
  c.check(geotr_gemm(q, ldq, proj_p->w, C, 1, qt, H * C, rows, C, ch, H, ch, ch * C, C, nullptr, nullptr, nullptr, 0, 1.0f, 0, c.stream));
  c.check(geotr_gemm(q, ldq, proj_p->b, 1, 1, qb, H, rows, 1, ch, H, ch, ch, 1, nullptr, nullptr, nullptr, 0, 1.0f, 0, c.stream));
}

// Attention cores of G (target, memory) cloud pairs in three grouped launches (+ two for the positional collapse):
// q rows of group g start at q + qrow[g] * ldq (n[g] rows), k / v rows at k + krow[g] * ldkv (m[g] rows); hidden (rows, C) gets the
// result at row qrow[g].  emb != nullptr (self-attention): emb[g] = (n[g], n[g], C) embedding, proj_p collapsed into qt / qb.
// pos_pre / qb_pre (self-attention, optional): the positional term already computed by the embedding launch (geotr_gse_embed_table_ex;
// same layout as the scores built here) and its q . b_p part -- then the embedding is not read at all.
static void attention_groups(Ctx& c, int H, int64_t C, int G, const int64_t* n, const int64_t* m, const float* q, int64_t ldq,
                             const int64_t* qrow, int64_t q_rows_total, const float* k, const float* v, int64_t ldkv, const int64_t* krow,
                             const float* const* emb, const geotr_linear* proj_p, float* hidden, const float* pos_pre = nullptr,
                             const float* qb_pre = nullptr) {
  const int64_t ch = C / H;
  const size_t mk = c.mark();
  geotr_gemm_groups g1, g2;
  geotr_attn_groups ga;
  std::memset(&g1, 0, sizeof(g1)), std::memset(&g2, 0, sizeof(g2)), std::memset(&ga, 0, sizeof(ga));
  g1.count = g2.count = ga.count = G;
  int64_t total = 0;
  for (int g = 0; g < G; ++g) {
    const int64_t mp = (m[g] + 3) / 4 * 4;
    // scores_g (H, n, mp) = q_h k_h^T
    g1.m[g] = n[g], g1.n[g] = m[g], g1.k[g] = ch;
    g1.lda[g] = ldq, g1.ldb[g] = ldkv, g1.ldc[g] = mp;
    g1.a_off[g] = qrow[g] * ldq, g1.b_off[g] = krow[g] * ldkv, g1.c_off[g] = total;
    g1.a_head_stride[g] = ch, g1.b_head_stride[g] = ch, g1.c_head_stride[g] = n[g] * mp;
    // hidden_g (n, C) = softmax(scores_g) v_h
    g2.m[g] = n[g], g2.n[g] = ch, g2.k[g] = m[g];
    g2.lda[g] = mp, g2.ldb[g] = ldkv, g2.ldc[g] = C;
    g2.a_off[g] = total, g2.b_off[g] = krow[g] * ldkv, g2.c_off[g] = qrow[g] * C;
    g2.a_head_stride[g] = n[g] * mp, g2.b_head_stride[g] = ch, g2.c_head_stride[g] = ch;
    ga.n[g] = n[g], ga.m[g] = m[g], ga.ld[g] = mp, ga.scores_off[g] = total, ga.q_row0[g] = qrow[g];
    ga.emb[g] = emb ? emb[g] : nullptr;
    total += (int64_t)H * n[g] * mp;
  }
  float* scores = c.alloc<float>((size_t)total);
  const bool own_pos = emb && !pos_pre;
  float* qt = own_pos ? c.alloc<float>((size_t)q_rows_total * H * C) : nullptr;
  float* qb = own_pos ? c.alloc<float>((size_t)q_rows_total * H) : nullptr;
  if (c.live()) {
    c.check(geotr_gemm_grouped(q, k, 0, scores, &g1, H, 1.0f, c.stream));
    if (pos_pre) {
      c.check(geotr_attn_softmax_grouped_pos(scores, &ga, pos_pre, qb_pre, H, 1.0f / sqrtf((float)ch), c.stream));
    } else {
      if (emb) positional_queries(c, H, C, q, ldq, q_rows_total, proj_p, qt, qb);
      c.check(geotr_attn_softmax_grouped(scores, &ga, qt, qb, C, H, 1.0f / sqrtf((float)ch), c.stream));
    }
    c.check(geotr_gemm_grouped(scores, v, 1, hidden, &g2, H, 1.0f, c.stream));
  }
  c.release(mk);
}

// feats_bb: (n_c, c_dim) coarse backbone features in stack order (ref_0, src_0, ref_1, ...); cloud_n[2B] superpoints per cloud;
// feats_out: (n_c, D) L2-normalised transformer features, stack order (what the matching heads read).
static void transformer_stack(Ctx& c, const geotr_transformer& t, int B, const int64_t* cloud_n, const float* pts_c, const float* feats_bb,
                              int64_t c_dim, float* feats_out) {
  const int64_t C = t.in_proj.out, D = t.out_proj.out;
  const int H = t.num_heads;
  int64_t stack0[2 * GEOTR_MAX_PAIRS], work0[2 * GEOTR_MAX_PAIRS];  // first row of cloud q in the stack / working layout
  int64_t NR = 0, NS = 0, N = 0;
  for (int b = 0; b < B; ++b) NR += cloud_n[2 * b], NS += cloud_n[2 * b + 1];
  {
    int64_t r = 0, s = NR;
    for (int q = 0; q < 2 * B; ++q) {
      stack0[q] = N;
      N += cloud_n[q];
      if (q % 2 == 0) work0[q] = r, r += cloud_n[q];
      else work0[q] = s, s += cloud_n[q];
    }
  }
  RowMove to_work, to_stack;
  to_work.n = to_stack.n = 2 * B;
  for (int q = 0; q < 2 * B; ++q) {
    to_work.src0[q] = stack0[q], to_work.dst0[q] = work0[q], to_work.rows[q] = cloud_n[q];
    to_stack.src0[q] = work0[q], to_stack.dst0[q] = stack0[q], to_stack.rows[q] = cloud_n[q];
  }
  float* xin = c.alloc<float>((size_t)N * c_dim);
  move_rows(c, feats_bb, xin, c_dim, to_work);
  const float* x = linear(c, t.in_proj, xin, c_dim, N, 0);
  // q / k / v of a self-attention layer: one fused (N, C) x (C, 3C) GEMM where the descriptor carries the concatenated weight
  auto project_qkv = [&](const geotr_attn_layer& L, const float* xl, const float*& q, const float*& k, const float*& v, int64_t& ld) {
    if (L.qkv_w) {
      float* qkv = c.alloc<float>((size_t)N * 3 * C);
      if (use_packed(L.qkv_packed, xl, C, N, C))
        c.check(packed_gemm(c, xl, C, L.qkv_packed, qkv, 3 * C, N, 3 * C, C, L.qkv_b, nullptr, nullptr, 0, 1.0f, 0, c.stream));
      else if (c.live())
        c.check(geotr_gemm(xl, C, L.qkv_w, C, 0, qkv, 3 * C, N, 3 * C, C, 1, 0, 0, 0, L.qkv_b, nullptr, nullptr, 0, 1.0f, 0, c.stream));
      q = qkv, k = qkv + C, v = qkv + 2 * C, ld = 3 * C;
    } else {
      q = linear(c, L.q, xl, C, N, 0), k = linear(c, L.k, xl, C, N, 0), v = linear(c, L.v, xl, C, N, 0), ld = C;
    }
  };
  // Round 5: the FIRST self-attention layer's positional scores come out of the embedding launch (gse_embed_table_kernel<.., POS>): its
  // queries are projected before the embedding exists, W_p is collapsed into them, and the table kernel contracts every e[i, j, :] with
  // qt[i, h, :] while it holds the row in registers -- that layer's softmax then never streams the (n, n, D) tensor.
  static const bool pos_fused_enabled = [] {
    const char* e = std::getenv("GEOTR_GSE_POS_FUSED");  // A/B switch for measurements: 0 = every self layer reads the embedding
    return !(e && e[0] == '0');
  }();
  const bool pos_fused = pos_fused_enabled && t.gse_precision == 5 && t.proj_d.out == 256 && C == 256 && H == 4 && t.num_layers > 0 &&
                         t.layers[0].is_self;
  const float *q0 = nullptr, *k0 = nullptr, *v0 = nullptr, *pos0 = nullptr, *qb0 = nullptr;
  int64_t ld0 = 0;
  // geometric structure embeddings, one (n, n, D) tensor per cloud
  const float* emb[2 * GEOTR_MAX_PAIRS];
  if (t.gse_precision == 5) {  // by table: all clouds of the stack in one ragged launch (+ one for the k nearest superpoints)
    geotr_gse_clouds gc;
    geotr_gse_pos gp;
    std::memset(&gc, 0, sizeof(gc)), std::memset(&gp, 0, sizeof(gp));
    gc.count = 2 * B;
    int64_t tot = 0, sq = 0, pos_tot = 0;
    for (int q = 0; q < 2 * B; ++q) {
      gc.n[q] = (int32_t)cloud_n[q], gc.row0[q] = (int32_t)stack0[q], gc.emb_off[q] = tot;
      tot += cloud_n[q] * cloud_n[q] * t.proj_d.out;
      sq += cloud_n[q] * cloud_n[q];
      const int64_t mp = (cloud_n[q] + 3) / 4 * 4;  // = the score layout attention_groups builds for the self layers
      gp.q_row0[q] = (int32_t)work0[q], gp.ld[q] = (int32_t)mp, gp.pos_off[q] = pos_tot;
      pos_tot += (int64_t)H * cloud_n[q] * mp;
    }
    float* qt_first = nullptr;
    float* pos_first = nullptr;
    if (pos_fused) {
      project_qkv(t.layers[0], x, q0, k0, v0, ld0);
      qt_first = c.alloc<float>((size_t)N * H * C);
      float* qb_first = c.alloc<float>((size_t)N * H);
      pos_first = c.alloc<float>((size_t)pos_tot);
      if (c.live()) positional_queries(c, H, C, q0, ld0, N, &t.layers[0].p, qt_first, qb_first);
      pos0 = pos_first, qb0 = qb_first;
    }
    float* emb_all = c.alloc<float>((size_t)tot);
    int32_t* knn = c.alloc<int32_t>((size_t)N * t.angle_k);
    for (int q = 0; q < 2 * B; ++q) emb[q] = emb_all + gc.emb_off[q];
    if (c.live()) {
      c.check(geotr_gse_knn_clouds(pts_c, &gc, t.angle_k, knn, c.stream));
      ProfScope prof(c.stream);
      c.check(geotr_gse_embed_table_ex(pts_c, knn, &gc, t.angle_k, t.proj_d.out, t.gse_table_d, t.gse_points_d, t.gse_table_a, t.gse_points_a,
                                       t.proj_d.w, t.proj_d.b, t.proj_a.w, t.proj_a.b, t.div_term, t.sigma_d, t.sigma_a, t.reduction_a,
                                       qt_first, pos_fused ? &gp : nullptr, pos_first, emb_all, c.stream));
      prof.done(-sq);
    }
  } else {  // fused sinusoid -> MFMA kernels, one launch per cloud, shared split-weight workspace
    char* gws = c.alloc<char>(geotr_gse_embed_workspace_bytes(t.proj_d.out, t.gse_precision) + 16);
    for (int q = 0; q < 2 * B; ++q) emb[q] = gse(c, t, pts_c + 3 * stack0[q], cloud_n[q], gws, q == 0);
  }

  for (int l = 0; l < t.num_layers; ++l) {
    const geotr_attn_layer& L = t.layers[l];
    float* y = c.alloc<float>((size_t)N * C);
    if (L.is_self) {
      const float *q, *k, *v;
      int64_t ld;
      const bool first_fused = l == 0 && pos_fused;
      if (first_fused) q = q0, k = k0, v = v0, ld = ld0;
      else project_qkv(L, x, q, k, v, ld);
      float* hidden = c.alloc<float>((size_t)N * C);
      attention_groups(c, H, C, 2 * B, cloud_n, cloud_n, q, ld, work0, N, k, v, ld, work0, emb, &L.p, hidden, first_fused ? pos0 : nullptr,
                       first_fused ? qb0 : nullptr);
      attn_tail(c, L, hidden, x, N, C, y);
    } else {
      // sequential cross-attention (conditional_transformer.py:110-111): refs attend to the sources, then the sources to the
      // UPDATED refs; half 0: targets = refs (rows [0, NR)), memory = sources; half 1: the other way round on y's new ref rows
      for (int half = 0; half < 2; ++half) {
        const int64_t t0 = half == 0 ? 0 : NR, nt = half == 0 ? NR : NS;
        const int64_t m0 = half == 0 ? NR : 0, nm = half == 0 ? NS : NR;
        const float* xt = x + t0 * C;
        const float* xm = half == 0 ? x + m0 * C : y + m0 * C;
        const float* q = linear(c, L.q, xt, C, nt, 0);
        const float *k, *v;
        int64_t ld;
        if (L.kv_w) {
          float* kv = c.alloc<float>((size_t)nm * 2 * C);
          if (use_packed(L.kv_packed, xm, C, nm, C))
            c.check(packed_gemm(c, xm, C, L.kv_packed, kv, 2 * C, nm, 2 * C, C, L.kv_b, nullptr, nullptr, 0, 1.0f, 0, c.stream));
          else if (c.live())
            c.check(geotr_gemm(xm, C, L.kv_w, C, 0, kv, 2 * C, nm, 2 * C, C, 1, 0, 0, 0, L.kv_b, nullptr, nullptr, 0, 1.0f, 0, c.stream));
          k = kv, v = kv + C, ld = 2 * C;
        } else {
          k = linear(c, L.k, xm, C, nm, 0), v = linear(c, L.v, xm, C, nm, 0), ld = C;
        }
        float* hidden = c.alloc<float>((size_t)nt * C);
        int64_t gn[GEOTR_MAX_PAIRS], gmm[GEOTR_MAX_PAIRS], qrow[GEOTR_MAX_PAIRS], krow[GEOTR_MAX_PAIRS];
        for (int b = 0; b < B; ++b) {
          const int gt = 2 * b + half, gm = 2 * b + 1 - half;  // target / memory cloud of pair b
          gn[b] = cloud_n[gt], gmm[b] = cloud_n[gm], qrow[b] = work0[gt] - t0, krow[b] = work0[gm] - m0;
        }
        attention_groups(c, H, C, B, gn, gmm, q, C, qrow, nt, k, v, ld, krow, nullptr, nullptr, hidden);
        attn_tail(c, L, hidden, xt, nt, C, y + t0 * C);
      }
    }
    x = y;
  }
  float* g = linear(c, t.out_proj, x, C, N, 0);
  float* gn = c.alloc<float>((size_t)N * D);
  if (c.live()) c.check(geotr_l2_normalize(g, N, D, gn, c.stream));
  move_rows(c, gn, feats_out, D, to_stack);
}

// heads of one pair: superpoint patches, geometric transformer, coarse matching, patch OT, LGR.  All pointers are the pair's
// own slices of the stacked arrays (reference cloud first).
// node_masks / node_knn_idx / node_knn_mask: the pair's slices of the stack-wide superpoint partition (reference cloud first)
static void run_pair(Ctx& c, const geotr_model& net, const float* pts_c, int64_t nr_c, int64_t ns_c, const float* pts_f, int64_t nr_f,
                     int64_t ns_f, const float* feats_f, int64_t c_f, const uint8_t* node_masks, const int64_t* node_knn_idx,
                     const uint8_t* node_knn_mask, const geotr_outputs& o, bool per_pair_tail, bool coarse_done) {
  (void)pts_c;
  const int64_t K = net.num_points_in_patch, P = net.num_correspondences;

  const int64_t D = net.transformer.out_proj.out;  // o.feats_c already holds the L2-normalised transformer features (transformer_stack)

  // 4. coarse matching (model.py:153-160) -- unless it already ran for the whole stack
  float* sim = coarse_done ? nullptr : c.alloc<float>((size_t)nr_c * ns_c);
  const size_t spm_bytes = coarse_done ? 0 : geotr_superpoint_match_workspace_bytes(nr_c, ns_c);
  char* spm_ws = coarse_done ? nullptr : c.alloc<char>(spm_bytes);
  if (c.live()) {
    if (!coarse_done) {
      c.check(geotr_gemm(o.feats_c, D, o.feats_c + nr_c * D, D, 0, sim, ns_c, nr_c, ns_c, D, 1, 0, 0, 0, nullptr, nullptr, nullptr, 0, 1.0f, 0,
                         c.stream));
      c.check(geotr_superpoint_match(sim, nr_c, ns_c, node_masks, node_masks + nr_c, net.dual_normalization, P, spm_ws, spm_bytes,
                                     o.ref_node_corr_indices, o.src_node_corr_indices, o.node_corr_scores, o.num_node_corr, c.stream));
    }
    // 5. patches of the selected pairs (model.py:169-179); rows >= *num_node_corr are neutral (pad index, mask False)
    c.check(geotr_patch_gather(node_knn_idx, node_knn_mask, pts_f, nr_f, o.ref_node_corr_indices, node_knn_idx + nr_c * K,
                               node_knn_mask + nr_c * K, pts_f + 3 * nr_f, ns_f, o.src_node_corr_indices, P, K, o.num_node_corr,
                               o.ref_knn_indices, o.ref_knn_masks, o.ref_knn_points, o.src_knn_indices, o.src_knn_masks,
                               o.src_knn_points, c.stream));
    if (per_pair_tail)  // 6. patch scores + optimal transport (model.py:187-191)
      c.check(geotr_patch_sinkhorn(feats_f, nr_f, feats_f + nr_f * c_f, ns_f, c_f, o.ref_knn_indices, o.src_knn_indices, o.ref_knn_masks,
                                   o.src_knn_masks, P, K, net.alpha, net.num_sinkhorn_iterations, nullptr, o.num_node_corr,
                                   o.matching_scores, c.stream));
  }
  if (!per_pair_tail) return;
  // 7. local-to-global registration on the dustbin-free block (model.py:195-210)
  const size_t lgr_bytes = geotr_lgr_workspace_bytes(P, K, net.topk);
  char* lgr_ws = c.alloc<char>(lgr_bytes);
  if (c.live())
    c.check(geotr_lgr(o.ref_knn_points, o.src_knn_points, o.ref_knn_masks, o.src_knn_masks, o.matching_scores, (K + 1) * (K + 1), K + 1, P,
                      K, net.topk, net.confidence_threshold, net.mutual, net.acceptance_radius, net.correspondence_threshold,
                      net.num_refinement_steps, o.num_node_corr, o.ref_corr_points, o.src_corr_points, o.corr_scores, o.num_corr,
                      o.estimated_transform, lgr_ws, lgr_bytes, c.stream));
}

// byte distance outs[1].x - outs[0].x if every pair b has x at outs[0].x + b * that distance, else -1
template <typename F>
static int64_t uniform_stride(const geotr_outputs* outs, int B, F field) {
  if (B < 2) return 0;
  const int64_t d = reinterpret_cast<const char*>(field(outs[1])) - reinterpret_cast<const char*>(field(outs[0]));
  for (int b = 2; b < B; ++b)
    if (reinterpret_cast<const char*>(field(outs[b])) - reinterpret_cast<const char*>(field(outs[0])) != d * b) return -1;
  return d;
}

// One forward over `p.num_pairs` stacked pairs (clouds ordered ref_0, src_0, ref_1, src_1, ...): the KPConv-FPN runs once over
// the whole stack (row-wise kernels; GroupNorm statistics per pair), the per-pair heads run pair after pair on slices.
static int run(Ctx& c, const geotr_model& net, const geotr_pyramid& p, const float* features, const geotr_outputs* outs) {
  const int S = net.backbone.num_stages, fine = net.backbone.fine_stage, B = p.num_pairs;
  c.nseg = B;
  static const bool spatial_order = [] {
    const char* e = std::getenv("GEOTR_SPATIAL_ORDER");  // A/B switch for measurements: GEOTR_SPATIAL_ORDER=0 visits rows in row order
    return !(e && e[0] == '0');
  }();
  for (int s = 0; s < p.num_stages; ++s) c.order[s] = spatial_order ? p.order[s] : nullptr;
  for (int s = 0; s < S; ++s)
    for (int b = 0; b < B; ++b) c.seg_rows[s][b] = p.cloud_n[s][2 * b] + p.cloud_n[s][2 * b + 1];
  const int64_t n_c = p.n[S - 1];
  const int64_t c_dim = net.backbone.blocks[net.backbone.num_blocks - 1].unary2.out;
  const int64_t c_f = net.backbone.decoder[net.backbone.num_decoders - 1].out;
  float* feats_f = c.dry ? nullptr : outs[0].feats_f;  // stacked (n_f, c_f): pair b's rows follow pair b-1's

  // KPConv-FPN (model.py:127-130); only the coarse features outlive the backbone's intermediates
  float* feats_c = c.alloc<float>((size_t)n_c * c_dim);
  const size_t mk = c.mark();
  BackboneOut bb = backbone_forward(c, net.backbone, p, features, feats_f);
  if (c.live()) c.check(copy_async(feats_c, bb.feats_c, sizeof(float) * (size_t)n_c * c_dim, c.stream));
  c.release(mk);

  // geometric transformer over the whole stack -> L2-normalised superpoint features in the caller's stack-ordered buffer
  {
    const size_t mt = c.mark();
    float* feats_c_out = c.dry ? nullptr : outs[0].feats_c;
    transformer_stack(c, net.transformer, B, p.cloud_n[S - 1], p.points[S - 1], feats_c, c_dim, feats_c_out);
    c.release(mt);
  }

  // The patch Sinkhorn and the local-to-global registration have the same shapes for every pair, so they run ONCE for the whole
  // stack (blockIdx.y = pair) when the caller laid the per-pair outputs out at a constant stride (native.py does); else per pair.
  const int64_t K = net.num_points_in_patch, P = net.num_correspondences;
  LgrBatch lb;
  std::memset(&lb, 0, sizeof(lb));
  bool batched_tail = !c.dry && B > 1;
  int64_t idx_stride = 0;
  if (batched_tail) {
    lb.knn_pts = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.ref_knn_points; });
    lb.knn_mask = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.ref_knn_masks; });
    lb.score = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.matching_scores; });
    lb.pcount = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.num_node_corr; });
    lb.corr_pts = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.ref_corr_points; });
    lb.corr_score = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.corr_scores; });
    lb.total = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.num_corr; });
    lb.transform = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.estimated_transform; });
    idx_stride = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.ref_knn_indices; });
    batched_tail = lb.knn_pts > 0 && lb.knn_mask > 0 && lb.score > 0 && lb.pcount > 0 && lb.corr_pts > 0 && lb.corr_score > 0 &&
                   lb.total > 0 && lb.transform > 0 && idx_stride > 0 &&
                   lb.knn_pts == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.src_knn_points; }) &&
                   lb.knn_mask == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.src_knn_masks; }) &&
                   lb.corr_pts == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.src_corr_points; }) &&
                   idx_stride == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.src_knn_indices; });
  }
  if (c.dry) batched_tail = B > 1;  // size query: reserve the batched workspace

  // superpoint patches (model.py:98-108) of every cloud of the stack in one launch pair
  const int64_t n_f_all = p.n[fine];
  int64_t* p2n = c.alloc<int64_t>((size_t)n_f_all);
  uint8_t* node_masks = c.alloc<uint8_t>((size_t)n_c);
  int64_t* node_knn_idx = c.alloc<int64_t>((size_t)n_c * K);
  uint8_t* node_knn_mask = c.alloc<uint8_t>((size_t)n_c * K);
  int32_t* scratch_flag = c.alloc<int32_t>(4);
  if (c.live()) {
    int64_t f0[2 * GEOTR_MAX_PAIRS + 1], c0[2 * GEOTR_MAX_PAIRS + 1];
    f0[0] = c0[0] = 0;
    for (int q = 0; q < 2 * B; ++q) f0[q + 1] = f0[q] + p.cloud_n[fine][q], c0[q + 1] = c0[q] + p.cloud_n[S - 1][q];
    c.check(zero_async(scratch_flag, 16, c.stream));
    c.check(p2n_launch(p.points[fine], p.points[S - 1], 2 * B, f0, c0, K, p2n, node_masks, node_knn_idx, node_knn_mask, scratch_flag, c.stream));
  }

  // coarse matching (model.py:153-160) of all pairs in one launch sequence when the per-pair outputs are uniformly strided
  bool coarse_stack = batched_tail;
  int64_t corr_stride = 0;
  if (coarse_stack && !c.dry) {
    corr_stride = uniform_stride(outs, B, [](const geotr_outputs& o) { return o.ref_node_corr_indices; });
    coarse_stack = corr_stride > 0 && corr_stride == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.src_node_corr_indices; }) &&
                   corr_stride / 2 == uniform_stride(outs, B, [](const geotr_outputs& o) { return o.node_corr_scores; });
  }
  if (coarse_stack) {
    const int64_t D = net.transformer.out_proj.out;
    int64_t pn[GEOTR_MAX_PAIRS], pm[GEOTR_MAX_PAIRS], s_off[GEOTR_MAX_PAIRS], m_off[GEOTR_MAX_PAIRS];
    geotr_gemm_groups gg;
    std::memset(&gg, 0, sizeof(gg));
    gg.count = B;
    int64_t tot = 0, oc = 0;
    for (int b = 0; b < B; ++b) {
      pn[b] = p.cloud_n[S - 1][2 * b], pm[b] = p.cloud_n[S - 1][2 * b + 1], s_off[b] = tot, m_off[b] = oc;
      gg.m[b] = pn[b], gg.n[b] = pm[b], gg.k[b] = D, gg.lda[b] = D, gg.ldb[b] = D, gg.ldc[b] = pm[b];
      gg.a_off[b] = oc * D, gg.b_off[b] = (oc + pn[b]) * D, gg.c_off[b] = tot;
      tot += pn[b] * pm[b], oc += pn[b] + pm[b];
    }
    float* sims = c.alloc<float>((size_t)tot);
    const size_t wsb = spm_stack_workspace_bytes(B, pn, pm);
    char* ws2 = c.alloc<char>(wsb);
    if (c.live()) {
      const geotr_outputs& o = outs[0];
      c.check(geotr_gemm_grouped(o.feats_c, o.feats_c, 0, sims, &gg, 1, 1.0f, c.stream));
      c.check(spm_stack_launch(sims, B, pn, pm, s_off, node_masks, m_off, net.dual_normalization, P, ws2, wsb, o.ref_node_corr_indices,
                               o.src_node_corr_indices, o.node_corr_scores, o.num_node_corr, corr_stride / 8, lb.pcount / 4, c.stream));
    }
  }

  int64_t off_c = 0, off_f = 0;
  const float* tail_ref_feats[GEOTR_MAX_PAIRS];
  const float* tail_src_feats[GEOTR_MAX_PAIRS];
  int64_t tail_nr[GEOTR_MAX_PAIRS], tail_ns[GEOTR_MAX_PAIRS];
  for (int b = 0; b < B; ++b) {
    const int64_t nr_c = p.cloud_n[S - 1][2 * b], ns_c = p.cloud_n[S - 1][2 * b + 1];
    const int64_t nr_f = p.cloud_n[fine][2 * b], ns_f = p.cloud_n[fine][2 * b + 1];
    geotr_outputs o;
    if (c.dry) std::memset(&o, 0, sizeof(o));
    else o = outs[b];
    if (!c.dry && (o.feats_f != feats_f + off_f * c_f || o.feats_c != outs[0].feats_c + off_c * net.transformer.out_proj.out) &&
        c.rc == GEOTR_OK)
      c.rc = fail(GEOTR_E_INVALID, "model_forward: outputs[%d].feats_f / feats_c must be the pair's rows of the stacked buffers", b);
    const size_t mp = c.mark();
    run_pair(c, net, p.points[S - 1] + 3 * off_c, nr_c, ns_c, p.points[fine] + 3 * off_f, nr_f, ns_f, feats_f + off_f * c_f, c_f,
             node_masks + off_c, node_knn_idx + off_c * K, node_knn_mask + off_c * K, o, !batched_tail, coarse_stack);
    tail_ref_feats[b] = feats_f + off_f * c_f, tail_src_feats[b] = feats_f + (off_f + nr_f) * c_f, tail_nr[b] = nr_f, tail_ns[b] = ns_f;
    c.release(mp);
    off_c += nr_c + ns_c;
    off_f += nr_f + ns_f;
  }
  if (batched_tail) {
    const size_t lgr_bytes = align_up(geotr_lgr_workspace_bytes(P, K, net.topk));
    char* lgr_ws = c.alloc<char>(lgr_bytes * (size_t)B);
    if (c.live()) {
      const geotr_outputs& o = outs[0];
      lb.ws = (int64_t)lgr_bytes;
      c.check(sinkhorn_launch(B, tail_ref_feats, tail_nr, tail_src_feats, tail_ns, c_f, o.ref_knn_indices, o.src_knn_indices, o.ref_knn_masks,
                              o.src_knn_masks, idx_stride / 8, lb.knn_mask, P, K, net.alpha,
                              net.num_sinkhorn_iterations, o.num_node_corr, lb.pcount / 4, o.matching_scores, lb.score / 4, c.stream));
      c.check(lgr_launch(o.ref_knn_points, o.src_knn_points, o.ref_knn_masks, o.src_knn_masks, o.matching_scores, (K + 1) * (K + 1), K + 1, P, K,
                         net.topk, net.confidence_threshold, net.mutual, net.acceptance_radius, net.correspondence_threshold,
                         net.num_refinement_steps, o.num_node_corr, o.ref_corr_points, o.src_corr_points, o.corr_scores, o.num_corr,
                         o.estimated_transform, lgr_ws, lgr_bytes, c.stream, B, lb));
    }
  }
  return c.rc;
}

}  // namespace geotr

using namespace geotr;

static int validate(const geotr_model* net, const geotr_pyramid* pyr) {
  GEOTR_CHECK_ARG(net && pyr, "model_forward: null descriptor");
  const int S = net->backbone.num_stages;
  GEOTR_CHECK_ARG(S >= 3 && S <= GEOTR_MAX_STAGES && pyr->num_stages == S, "model_forward: %d stages (pyramid has %d)", S, pyr->num_stages);
  GEOTR_CHECK_ARG(net->backbone.num_blocks == 2 + 3 * (S - 1), "model_forward: backbone block count does not match the depth");
  GEOTR_CHECK_ARG(net->transformer.num_layers >= 1 && net->transformer.num_layers <= 8, "model_forward: 1..8 transformer layers");
  GEOTR_CHECK_ARG(net->transformer.reduction_a == 0 || (net->transformer.reduction_a == 1 && net->transformer.gse_precision == 5),
                  "model_forward: reduction_a must be 0 (max) or 1 (mean; with the table embedding, gse_precision 5, only)");
  GEOTR_CHECK_ARG(pyr->num_pairs >= 1 && pyr->num_pairs <= GEOTR_MAX_PAIRS, "model_forward: 1..%d stacked pairs", GEOTR_MAX_PAIRS);
  for (int s = 0; s < S; ++s) {
    int64_t tot = 0;
    for (int q = 0; q < 2 * pyr->num_pairs; ++q) {
      GEOTR_CHECK_ARG(pyr->cloud_n[s][q] > 0, "model_forward: empty cloud %d at stage %d", q, s);
      tot += pyr->cloud_n[s][q];
    }
    GEOTR_CHECK_ARG(tot == pyr->n[s], "model_forward: stage %d has %lld rows but its clouds sum to %lld", s, (long long)pyr->n[s], (long long)tot);
  }
  return GEOTR_OK;
}

extern "C" {

int geotr_profile_gse(void** start_events, void** stop_events, int64_t* sizes, int64_t capacity) {
  GEOTR_CHECK_ARG(capacity >= 0 && (capacity == 0 || (start_events && stop_events && sizes)), "profile_gse: bad arguments");
  g_prof_cap = 0;
  g_prof_start = start_events;
  g_prof_stop = stop_events;
  g_prof_size = sizes;
  g_prof_used.store(0);
  g_prof_seq.store(0);
  g_prof_cap = (int)capacity;
  return GEOTR_OK;
}

int geotr_profile_stride(int64_t stride) {
  GEOTR_CHECK_ARG(stride >= 1 && stride < (1 << 20), "profile_stride: stride must be >= 1");
  g_prof_stride = (int)stride;
  return GEOTR_OK;
}

int64_t geotr_profile_gse_count(void) {
  const int used = g_prof_used.load();
  return used < g_prof_cap ? used : g_prof_cap;
}

size_t geotr_model_workspace_bytes(const geotr_model* net, const geotr_pyramid* pyr) {
  if (validate(net, pyr) != GEOTR_OK) return 0;
  Ctx c;
  c.base = nullptr;
  c.cap = 0;
  c.stream = nullptr;
  c.dry = true;
  c.gemm_mode = net->gemm_mode;
  run(c, *net, *pyr, nullptr, nullptr);
  return c.peak + 4096;
}

int geotr_model_forward(const geotr_model* net, const geotr_pyramid* pyr, const float* features, const geotr_outputs* out, void* ws,
                        size_t ws_bytes, void* stream) {
  const int v = validate(net, pyr);
  if (v != GEOTR_OK) return v;
  GEOTR_CHECK_ARG(features && out && ws, "model_forward: null pointer");
  GEOTR_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "model_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.base = reinterpret_cast<char*>(ws);
  c.cap = ws_bytes;
  c.stream = (hipStream_t)stream;
  c.dry = false;
  c.gemm_mode = net->gemm_mode;
  return run(c, *net, *pyr, features, out);
}

size_t geotr_pyramid_workspace_bytes(int64_t n0, int64_t batch, int64_t num_stages) {
  return align_up(geotr_grid_subsample_workspace_bytes(n0, batch)) +
         (size_t)num_stages * align_up(geotr_radius_grid_workspace_bytes(n0, batch)) + 4096;
}

// The whole pyramid is enqueued without a single host read (round 3): every launch is sized from the row capacity n0, the kernels read
// the stage sizes from the device-resident lengths (neighbors.hip), and the cell budget of a stage's grid comes from a hint.
static int pyramid_enqueue(const float* points, const int64_t* lengths, int64_t batch, int64_t n0, int64_t num_stages, float voxel_size,
                           float radius, const int64_t* limits_host, const geotr_pyramid_buffers* buf, int32_t* overflow, void* ws,
                           size_t ws_bytes, hipStream_t stream) {
  GEOTR_CHECK_ARG(points && lengths && limits_host && buf && ws, "pyramid_build: null pointer");
  GEOTR_CHECK_ARG(num_stages >= 1 && num_stages <= GEOTR_MAX_STAGES && batch >= 1 && n0 >= 1, "pyramid_build: bad sizes");
  if (ws_bytes < geotr_pyramid_workspace_bytes(n0, batch, num_stages)) return fail(GEOTR_E_WORKSPACE, "pyramid_build: workspace too small");
  const int S = (int)num_stages;
  char* base = reinterpret_cast<char*>(ws);
  const size_t gs_bytes = align_up(geotr_grid_subsample_workspace_bytes(n0, batch));
  const size_t grid_bytes = align_up(geotr_radius_grid_workspace_bytes(n0, batch));
  void* gs_ws = base;
  const float* pts[GEOTR_MAX_STAGES];
  const int64_t* len[GEOTR_MAX_STAGES];
  int64_t hint[GEOTR_MAX_STAGES];
  pts[0] = points;
  len[0] = lengths;
  hint[0] = n0;
  float v = voxel_size;
  for (int i = 1; i < S; ++i) {  // data.py:23-28: stage i is the grid subsample of stage i-1 at voxel * 2^i
    v *= 2.0f;
    // (n0 is the capacity of every stage: a subsample never has more points than its input)
    int rc = geotr_grid_subsample(pts[i - 1], len[i - 1], batch, n0, v, buf->points[i], buf->lengths[i], gs_ws, gs_bytes, stream);
    if (rc != GEOTR_OK) return rc;
    pts[i] = buf->points[i];
    len[i] = buf->lengths[i];
    hint[i] = std::max<int64_t>(hint[i - 1] / 3, 1024);  // a voxel twice as large keeps between a quarter and a third of a surface's points
  }
  // one uniform grid per stage serves the three searches against that stage (data.py:31-69)
  float r = radius;
  void* grids[GEOTR_MAX_STAGES];
  for (int i = 0; i < S; ++i) {
    grids[i] = base + gs_bytes + (size_t)i * grid_bytes;
    int rc = radius_grid_build_hinted(pts[i], len[i], batch, n0, hint[i], r, grids[i], grid_bytes, stream);
    if (rc == GEOTR_OK && buf->order[i]) rc = radius_grid_order_hinted(grids[i], n0, hint[i], batch, buf->order[i], stream);
    if (rc != GEOTR_OK) return rc;
    r *= 2.0f;
  }
  r = radius;
  // every search of the pyramid has radius / (voxel of the support stage) = radius / voxel_size; <= 3 voxels = a sparse candidate set
  const int sparse = radius <= 3.0f * voxel_size ? 1 : 0;
  // Row capacity of the fixed-width searches: after grid subsampling a ball of 2.5 voxels holds at most the voxels within
  // 2.5 + sqrt(3) voxel sizes of its centre (~320), so 512 cannot overflow from stage 1 on; stage 0 (raw input) raises if it does.
  const int64_t kRowCap = 512;
  for (int i = 0; i < S; ++i) {
    GEOTR_CHECK_ARG(limits_host[i] >= 1 && limits_host[i] < (1 << 20), "pyramid_build: bad neighbour limit at stage %d", i);
    // (the visiting order of the QUERY rows -- their own stage's grid order -- selects the LDS-staged tile kernel: neighbours in that
    // order share their candidate cells)
    // (every search is one launch: bracketed for bench.py's live roofline of the radius family when the profiler is armed)
    auto search = [&](int qs, int ss, const void* grid, float rad, int64_t width, int64_t* table) -> int {
      ProfScope prof(stream);
      const int rc = radius_query_hinted(false, grid, pts[qs], len[qs], batch, n0, hint[qs], n0, hint[ss], rad, width, kRowCap, table, nullptr, nullptr,
                                         overflow, stream, buf->order[qs], sparse);
      prof.done(kProfRadius | ((int64_t)qs << 56) | ((int64_t)ss << 52) | (width << 20) | ((int64_t)(sparse ? 0 : 1) << 19));
      return rc;
    };
    int rc = search(i, i, grids[i], r, limits_host[i], buf->neighbors[i]);
    if (rc != GEOTR_OK) return rc;
    if (i < S - 1) {
      rc = search(i + 1, i, grids[i], r, limits_host[i], buf->subsampling[i]);
      if (rc != GEOTR_OK) return rc;
      rc = search(i, i + 1, grids[i + 1], 2.0f * r, limits_host[i + 1], buf->upsampling[i]);
      if (rc != GEOTR_OK) return rc;
    }
    r *= 2.0f;
  }
  return GEOTR_OK;
}

int geotr_pyramid_build(const float* points, const int64_t* lengths, int64_t batch, int64_t n0, int64_t num_stages, float voxel_size,
                        float radius, const int64_t* limits_host, const geotr_pyramid_buffers* buf, int64_t* lengths_host,
                        int32_t* overflow, void* ws, size_t ws_bytes, void* stream_) {
  GEOTR_CHECK_ARG(lengths_host, "pyramid_build: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = pyramid_enqueue(points, lengths, batch, n0, num_stages, voxel_size, radius, limits_host, buf, overflow, ws, ws_bytes, stream);
  if (rc != GEOTR_OK) return rc;
  // the stage sizes, read ONCE after everything is enqueued (round 2 read them stage by stage: num_stages - 1 stream synchronisations)
  for (int i = 0; i < (int)num_stages; ++i)
    if (hipMemcpyAsync(lengths_host + (size_t)i * batch, i == 0 ? lengths : buf->lengths[i], sizeof(int64_t) * batch, hipMemcpyDeviceToHost,
                       stream) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "pyramid_build: memcpy failed");
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(GEOTR_E_LAUNCH, "pyramid_build: reading the stage sizes failed");
  // an empty stage is an error of THIS call, as before the device-resident sizes of round 3 (C / C++ callers got GEOTR_OK and
  // fed a zero-row stage to geotr_model_forward); the async variant leaves this check to the caller, who reads the sizes later
  for (int i = 0; i < (int)num_stages; ++i) {
    int64_t rows = 0;
    for (int64_t b = 0; b < batch; ++b) rows += lengths_host[(size_t)i * batch + b];
    if (rows < 1) return fail(GEOTR_E_INVALID, "pyramid_build: stage %d is empty", i);
  }
  return GEOTR_OK;
}

// The same without any host synchronisation (and therefore WITHOUT the empty-stage check: the caller examines the sizes when it reads
// them): `lengths_pinned` must be device-accessible host memory (hipHostMalloc / a pinned torch
// tensor) of num_stages x batch int64; it holds the stage sizes once the stream has passed this call -- the caller reads it after its
// next synchronisation of the stream (e.g. together with the previous stack's result counts: ONE host wait per stack).
int geotr_pyramid_build_async(const float* points, const int64_t* lengths, int64_t batch, int64_t n0, int64_t num_stages, float voxel_size,
                              float radius, const int64_t* limits_host, const geotr_pyramid_buffers* buf, int64_t* lengths_pinned,
                              int32_t* overflow, void* ws, size_t ws_bytes, void* stream_) {
  GEOTR_CHECK_ARG(lengths_pinned, "pyramid_build_async: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = pyramid_enqueue(points, lengths, batch, n0, num_stages, voxel_size, radius, limits_host, buf, overflow, ws, ws_bytes, stream);
  if (rc != GEOTR_OK) return rc;
  for (int i = 0; i < (int)num_stages; ++i)  // written by a kernel on the stream: an ordinary in-order dispatch, no blit path
    if (copy_async(lengths_pinned + (size_t)i * batch, i == 0 ? lengths : buf->lengths[i], sizeof(int64_t) * batch, stream) != GEOTR_OK)
      return GEOTR_E_LAUNCH;
  return GEOTR_OK;
}

}  // extern "C"
