// kpconv_fused.hip -- K1 as ONE kernel: KPConv (geotransformer/modules/kpconv/kpconv.py:79-121) without the (M, 15 C_in) operand
// in HBM.
//
// The reference computes, per query point m with neighbours h = 0 .. H-1 (pad index -> zero row, zero weight):
//     w[k, h]   = max(0, 1 - |(s_h - q_m) - kp_k| / sigma)                 15 kernel points            (:91-99)
//     g[k, c]   = sum_h w[k, h] f[h, c]                                     (15, C_in)  "weighted"      (:102-105)
//     out[m, :] = sum_{k, c} g[k, c] W[k, c, :]  / max(#{h : sum_c f[h, c] > 0}, 1) + bias            (:108-117)
// The two-kernel path (kpconv.hip gather -> gemm.hip packed GEMM) writes g for every point -- 15 C_in floats, 250 MB per 20k+20k
// pair over the backbone -- and reads it back: the deep-K GEMMs it feeds are bandwidth-bound on exactly that operand (47 TFLOP/s
// alone at 12 288 x 64 x 960, profiles/r02_bench_n1.json).  Here a workgroup owns 32 query points and keeps g in LDS:
//
//   phase 1  (matrix pipe, exact fp32)  one wave per point:  g (16 x C_in) = w (16 x H) . f (H x C_in)  as v_mfma_f32_16x16x4_f32
//            steps over 4 neighbours.  A operand = the lane's own influence w[k = lane & 15][h = 4 s + (lane >> 4)], computed in
//            registers from the relative position; B operand = the neighbour's feature channels, fetched with ONE 8 / 16-byte
//            load per lane and step that feeds VEC MFMAs (channel <-> column map: column n of tile j in group g is channel
//            16 VEC g + VEC n + j, so 16 lanes read a neighbour's row as one contiguous segment).  The MFMA is a k-ordered
//            fmaf chain (bitwise the VALU kernel's sum over h).  The result is split into bf16 hi / lo and stored to the LDS
//            tile A[point][k C_in + c] (rows padded by 16 B: conflict-free ds_read_b128 fragments).
//   phase 2  (matrix pipe, split-bf16)   out (32 x C_out) = A (32 x 15 C_in) . W  with v_mfma_f32_32x32x16_bf16, three products
//            per step (a_hi b_lo + a_lo b_hi + a_hi b_hi) against the SAME packed weight planes geotr_gemm_pack builds for the
//            two-kernel path, streamed fragment by fragment from L2.  Waves split (column tile, K range); the K partials meet in
//            LDS, the epilogue (/ count + bias) writes whole rows.
// Wider layers (C_in >= 128) stay on the two-kernel path: round 5 built them as channel blocks of 64 through this tile and measured
// 3.7 % SLOWER end to end (profiles/r05_ab_runs.md section 3: they are matrix-bound, phase 2 streams the 1-4 MB weight from L2 once per
// 32-row tile); round 6 removed those instantiations (git: 2b000bd has them).
// Supported: C_in = 32 or 64, C_out a multiple of 32 with C_out / 32 dividing the wave count (32 .. 256), H <= 40.
// Everything else stays on the two-kernel path.
#include <cstdlib>

#include "common.h"

namespace geotr {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifdef GEOTR_KPF_STAMPS  // measurement build only (scripts/abi_bench.cpp `kpconv`): cycles of wave 0 per section, summed over tiles and blocks
__device__ unsigned long long g_kpf_stamps[8];
#define KPF_STAMP(i)                                             \
  do {                                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += now_ - stamp_t;                              \
    stamp_t = now_;                                              \
  } while (0)
#else
#define KPF_STAMP(i) do {} while (0)
#endif

constexpr int kFusedRows = 32;  // query points per workgroup tile = one 32-row MFMA tile
constexpr int kMaxSteps = 10;   // neighbour steps of 4 held in registers: H <= 40 (every reference config: 24 .. 40)

template <int V>
struct FVec;
template <>
struct FVec<2> {
  using T = float2;
};
template <>
struct FVec<4> {
  using T = float4;
};

// C = C_in; WAVES = waves per workgroup; TERMS = 3 split-bf16 / 1 plain bf16 (hi planes only) / 0 exact fp32: the tile A stays fp32
// in LDS (same bytes as its hi + lo halves) and phase 2 runs v_mfma_f32_32x32x2_f32 against the weight packed by geotr_gemm_pack_f32
// (gemm.hip: four steps of an 8-deep group per 16-byte fragment) -- the reference's own arithmetic end to end (round 4).
// C = 32 tiles (78 KB of LDS) are meant to share a CU two by two: 4 waves per SIMD, i.e. at most 128 VGPRs (C = 64: one workgroup, 256)
template <int C, int WAVES, int TERMS>
__global__ __launch_bounds__(64 * WAVES, C == 32 ? 2 * WAVES / 4 : WAVES / 4) void kpconv_fused_kernel(const float* __restrict__ feats, const float* __restrict__ qp,
                                                                  const float* __restrict__ sp, const int64_t* __restrict__ nb,
                                                                  const float* __restrict__ kp, const unsigned char* __restrict__ pos,
                                                                  int64_t M, int64_t Ns, int H, float sigma, int c_out, int KS, int NT,
                                                                  const unsigned short* __restrict__ Bhi, const unsigned short* __restrict__ Blo,
                                                                  const float* __restrict__ bias, const int* __restrict__ order,
                                                                  float* __restrict__ out) {
  constexpr int VEC = C >= 64 ? 4 : 2;          // feature channels per lane and load
  constexpr int G = C / (16 * VEC);             // loads (column-tile groups) per neighbour step
  constexpr int K = 15 * C;                     // contraction depth of phase 2
  constexpr int RS = K + 8;                     // LDS row stride in bf16 elements (16 B pad)
  constexpr int RS32 = K + 4;                   // fp32 tile: row stride in floats (16 B pad; the tile has the bytes of the hi + lo halves)
  constexpr bool F32 = TERMS == 0;
  constexpr int PPW = kFusedRows / WAVES;       // points per wave in phase 1
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  unsigned short* A_hi = reinterpret_cast<unsigned short*>(fsm);
  unsigned short* A_lo = A_hi + kFusedRows * RS;
  float* A_32 = reinterpret_cast<float*>(fsm);  // TERMS == 0: [32][RS32] fp32
  // (rel.xyz, neighbour index bits) per (point, neighbour) live in the first 640 bytes of the point's own tile row until its MFMAs have
  // consumed them (relw_of below) -- the tile is ALL the kernel's LDS: 62 KB at C = 32 (two workgroups per CU with room to spare),
  // 124 KB at C = 64
  int* cnt_s = reinterpret_cast<int*>(A_lo + kFusedRows * RS);          // [32] neighbours with a positive feature sum
  int* row_s = cnt_s + kFusedRows;                                       // [32] the tile's query rows (visiting order applied)
  float* part = reinterpret_cast<float*>(fsm);                           // [WAVES][16][64] K partials (reuses the A tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, q4 = lane >> 4;
  const float inv_sigma = 1.f / sigma;
  // the feature rows as a raw buffer (Ns C floats < 4 GB: checked by the host): loads past its end return zeros
  const __amdgpu_buffer_rsrc_t feat_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats), 0, (int)((unsigned)Ns * (unsigned)C * 4u), 0x00020000);
  // the lane's kernel point (row of the phase-1 A operand); row 15 is padding
  const bool kp_ok = n16 < 15;
  const float kx = kp_ok ? kp[3 * n16] : 0.f, ky = kp_ok ? kp[3 * n16 + 1] : 0.f, kz = kp_ok ? kp[3 * n16 + 2] : 0.f;
  const int steps = (H + 3) >> 2;
  const int CT = c_out >> 5;            // 32-column tiles of the output
  const int KPARTS = WAVES / CT;        // K ranges of phase 2
  const int ct = wave % CT, kpart = wave / CT;
  const int nkk = F32 ? K / 8 : K / 16;  // 16-deep steps of phase 2 (K % 32 == 0); fp32: 8-deep groups of four 32x32x2 steps
  const int kk_per = (nkk + KPARTS - 1) / KPARTS;
  const int kk0 = kpart * kk_per, kk1 = min(nkk, kk0 + kk_per);

  // Tile t holds the query rows order[32 t .. 32 t + 31] (the grid order of the pyramid: spatial neighbours, which share most of
  // their neighbour rows) or rows 32 t .. when no order is given; outputs land in their own rows either way, and a row's result never
  // depends on its tile mates.  Blocks take CONTIGUOUS tile ranges, XCD by XCD (block b runs on XCD b % 8): the blocks resident on one
  // XCD work through one stretch of the order, so the neighbour rows they share are hits in that XCD's L2.
  // Rows past M are computed on a clamped index, never stored.
  auto load_rows = [&](int64_t tile) -> int {  // lane i < PPW: the wave's i-th point of that tile
    const int64_t t = min(tile * kFusedRows + wave * PPW + min(lane, PPW - 1), M - 1);
    return order ? order[t] : (int)t;
  };
  // Round 6: the dependent chain  order -> neighbour index -> support position  of a tile is off the tile's critical path.  Section
  // stamps of round 5's kernel (profiles/r06_kpconv_sections.md): a wave spent ~6 000 cycles per point in phase 1 against 640 / 1 152
  // cycles of matrix-pipe work (C_in = 32 / 64) -- every point paid two dependent global round trips (index, then position) that a
  // two-deep software pipeline over only FOUR points per wave could not hide, and the influence arithmetic of a point ran as one block
  // ahead of its MFMAs.  Now: the rows of a tile are fetched TWO tiles ahead, the neighbour indices of its four points one tile ahead
  // (at the start of the previous tile's phase 1: four loads in flight together), their positions and flags at the start of the
  // previous tile's phase 2 (sixteen loads in flight together, landing under the weight stream) -- so a tile starts with everything
  // but the feature rows in registers; and inside phase 1 the influence arithmetic + feature loads of point i + 1 are interleaved step
  // by step with the MFMAs of point i.  Same operations in the same order per output element: bit-identical results.
  auto load_idx = [&](int rows, int (&jx)[PPW]) {
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      const int64_t m = __builtin_amdgcn_readlane(rows, p);
      jx[p] = lane < H ? (int)nb[m * H + lane] : (int)Ns;  // (Ns < 2^31: checked by the host)
    }
  };
  auto load_rel = [&](int rows, const int (&jx)[PPW], float (&rx)[PPW][3], int (&cn)[PPW]) {
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      const int64_t m = __builtin_amdgcn_readlane(rows, p);
      const int64_t j = jx[p];
      bool counted = false;
      rx[p][0] = rx[p][1] = rx[p][2] = 0.f;
      if (j < Ns) {
        rx[p][0] = sp[3 * j] - qp[3 * m], rx[p][1] = sp[3 * j + 1] - qp[3 * m + 1], rx[p][2] = sp[3 * j + 2] - qp[3 * m + 2];
        counted = pos[j] != 0;
      }
      cn[p] = __popcll(__ballot(counted));
    }
  };
  // (relative position, neighbour index bits) of point p: the first 40 float4 of the point's OWN row of the A tile (nothing is stored
  // there before the point's MFMAs have consumed them)
  auto relw_of = [&](int p) -> float4* {
    if constexpr (F32) return reinterpret_cast<float4*>(A_32 + (wave * PPW + p) * RS32);
    else return reinterpret_cast<float4*>(A_hi + (wave * PPW + p) * RS);
  };
  const int64_t tiles = (M + kFusedRows - 1) / kFusedRows;
  const int lblock = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);  // gridDim.x % 8 == 0 (host)
  const int64_t per_block = (tiles + gridDim.x - 1) / gridDim.x;
  const int64_t tile_begin = (int64_t)lblock * per_block, tile_end = min(tiles, (int64_t)(lblock + 1) * per_block);
  float a_r[2][kMaxSteps];            // influences (A operands) of the point in flight / the next one
  float b_r[2][kMaxSteps][G][VEC];    // their neighbours' channels (B operands)
  // (C = 32 runs two workgroups per CU at <= 128 VGPRs: the other workgroup covers the round trip and the 20 registers are not there)
  constexpr bool PRE_B0 = C >= 64;
  bool b0_loaded = false;             // set 0 of b_r already holds point 0 of the tile about to start (loaded under the previous tile's epilogue)
  // neighbour (4 u + q4)'s channels of one point: ONE bounds-checked buffer load per group -- an absent neighbour (pad index, lanes past
  // H, steps past `steps`) gets an out-of-range offset, for which the hardware returns zeros
  auto load_feats = [&](int id, bool ok, float (&b)[G][VEC]) {
    const unsigned off = ok ? 4u * ((unsigned)id * (unsigned)C + (unsigned)(VEC * n16)) : 0xffffffffu;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if constexpr (VEC == 4) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(feat_rsrc, off, 64 * VEC * g, 0));
#pragma unroll
        for (int j = 0; j < VEC; ++j) b[g][j] = v[j];
      } else {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(feat_rsrc, off, 64 * VEC * g, 0));
#pragma unroll
        for (int j = 0; j < VEC; ++j) b[g][j] = v[j];
      }
    }
  };
  int rows_cur = 0, rows_nxt = 0;
  int jx_cur[PPW], cn_cur[PPW];
  float rx_cur[PPW][3];
#pragma unroll
  for (int p = 0; p < PPW; ++p) jx_cur[p] = 0, cn_cur[p] = 0, rx_cur[p][0] = rx_cur[p][1] = rx_cur[p][2] = 0.f;
  if (tile_begin < tile_end) {
    rows_cur = load_rows(tile_begin);
    rows_nxt = tile_begin + 1 < tile_end ? load_rows(tile_begin + 1) : 0;
    load_idx(rows_cur, jx_cur);
    load_rel(rows_cur, jx_cur, rx_cur, cn_cur);
  }
#ifdef GEOTR_KPF_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
    const int64_t m0 = tile * kFusedRows;
    const int rows_nn = tile + 2 < tile_end ? load_rows(tile + 2) : 0;  // in flight under this tile's work
    if (lane < PPW) row_s[wave * PPW + lane] = rows_cur;                // read by the epilogue, three barriers later
    // the next tile's neighbour indices: issued now, consumed (position loads) at the start of this tile's phase 2
    int jx_nxt[PPW], cn_nxt[PPW];
    float rx_nxt[PPW][3];
    const bool have_next = tile + 1 < tile_end;
    if (have_next) load_idx(rows_nxt, jx_nxt);
    else {
#pragma unroll
      for (int p = 0; p < PPW; ++p) jx_nxt[p] = (int)Ns;
    }
    f32x16 acc2;  // phase 2's accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    // ------------------------------------------------------------------ phase 1: g = w . f per point
#if defined(GEOTR_KPF_PRIO) && GEOTR_KPF_PRIO == 1
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);  // experiment: the younger half loses every arbitration against the older one
#elif defined(GEOTR_KPF_PRIO) && GEOTR_KPF_PRIO == 2
    __builtin_amdgcn_s_setprio(wave & 3);
#endif
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      if (lane < 40) relw_of(p)[lane] = make_float4(rx_cur[p][0], rx_cur[p][1], rx_cur[p][2], __int_as_float(jx_cur[p] < Ns ? jx_cur[p] : -1));
      if (lane == 0) cnt_s[wave * PPW + p] = cn_cur[p];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one neighbour step of point i: the lane's influence (A operand) and its neighbour's channels (B operands, one load per group)
    // Branch-free: one 16-byte LDS read, selects, one buffer load per group (the plain-load form compiled to two branches and two
    // dependent LDS reads per step, which also pinned the MFMAs between them).  `with_feats` false: the channels are already in b.
    auto b1_step = [&](int i, int u, float& a, float (&b)[G][VEC], bool with_feats) {
      const float4 rv = relw_of(i)[4 * u + q4];  // lanes >= H of the row hold index -1
      const int id = __float_as_int(rv.w);
      const bool ok = u < steps && id >= 0;
      const float dx = rv.x - kx, dy = rv.y - ky, dz = rv.z - kz;
      const float w = fmaxf(1.f - __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz) * inv_sigma, 0.f);  // as kpconv_gather_kernel
      a = w * ((ok && kp_ok) ? 1.f : 0.f);  // (w is finite and >= 0: the product is w or +0 exactly; a select here is compiled to a branch around the arithmetic)
      if (with_feats) load_feats(id, ok, b);
    };
    if (PRE_B0 && b0_loaded) {  // (uniform)
#pragma unroll
      for (int u = 0; u < kMaxSteps; ++u) b1_step(0, u, a_r[0][u], b_r[0][u], false);
    } else {
#pragma unroll
      for (int u = 0; u < kMaxSteps; ++u) b1_step(0, u, a_r[0][u], b_r[0][u], true);
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int cur = i & 1, nxt = cur ^ 1;
      f32x4 acc[G][VEC];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < kMaxSteps; ++u) {  // steps past `steps` multiply zeros (their operands are 0): the order of the sum over h is kept
        if (i + 1 < PPW) b1_step(i + 1, u, a_r[nxt][u], b_r[nxt][u], true);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[cur][u], b_r[cur][u][g][j], acc[g][j], 0, 0, 0);
      }
      // accumulator (16 kernel points x 16 columns per tile): lane holds rows 4 q4 + r, column n16 -> A[row][k C + channel]
      const int row = wave * PPW + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kpt = 4 * q4 + r;
        if (kpt >= 15) continue;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if constexpr (F32) {  // the fp32 sums as they are
            float* d = A_32 + row * RS32 + kpt * C + 16 * VEC * g + VEC * n16;
            if constexpr (VEC == 2) *reinterpret_cast<float2*>(d) = make_float2(acc[g][0][r], acc[g][1][r]);
            else *reinterpret_cast<float4*>(d) = make_float4(acc[g][0][r], acc[g][1][r], acc[g][2][r], acc[g][3][r]);
            continue;
          }
          unsigned h[VEC], l[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) {  // hi = bf16(x), lo = bf16(x - hi), round to nearest even (v_cvt_pk_bf16_f32, as gemm.hip)
            const float x = acc[g][j][r];
            const __bf16 hb_ = (__bf16)x;
            const __bf16 lb_ = (__bf16)(x - (float)hb_);
            h[j] = (unsigned)__builtin_bit_cast(unsigned short, hb_);
            l[j] = (unsigned)__builtin_bit_cast(unsigned short, lb_);
          }
          const int e = row * RS + kpt * C + 16 * VEC * g + VEC * n16;  // VEC consecutive bf16: 4- or 8-byte aligned
          if constexpr (VEC == 2) {
            *reinterpret_cast<unsigned*>(A_hi + e) = h[0] | (h[1] << 16);
            if constexpr (TERMS == 3) *reinterpret_cast<unsigned*>(A_lo + e) = l[0] | (l[1] << 16);
          } else {
            *reinterpret_cast<uint2*>(A_hi + e) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            if constexpr (TERMS == 3) *reinterpret_cast<uint2*>(A_lo + e) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
          }
        }
      }
    }
#ifdef GEOTR_KPF_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    KPF_STAMP(0);     // phase 1 (this wave's points)
    // the next tile's positions and flags: sixteen loads in flight together, landing under the weight stream of phase 2
    if (have_next) load_rel(rows_nxt, jx_nxt, rx_nxt, cn_nxt);
    else {
#pragma unroll
      for (int p = 0; p < PPW; ++p) cn_nxt[p] = 0, rx_nxt[p][0] = rx_nxt[p][1] = rx_nxt[p][2] = 0.f;
    }
    __syncthreads();  // A tile complete
    KPF_STAMP(1);     // wait for the other waves' points
    // ------------------------------------------------------------------ phase 2: out += A . W  (this wave: column tile ct, steps kk0 .. kk1)
    if constexpr (F32) {
      // group q: lane (fr, fk) holds A[fr][8 q + 4 fk + e] and W[8 q + 4 fk + e][32 ct + fr], e = 0 .. 3 = its operands of four steps
      const float* a32 = A_32 + (lane & 31) * RS32 + 4 * (lane >> 5);
      const float* b32 = reinterpret_cast<const float*>(Bhi) + ((int64_t)min(ct, NT - 1) * 2 * KS * 64 + lane) * 4;
      // weight fragments three groups ahead (an L2 round trip is 2-3 groups of MFMAs long), the tile's own fragment one group ahead
      constexpr int WD = 3;
      f32x4 bq[WD];
#pragma unroll
      for (int d = 0; d < WD; ++d) bq[d] = *reinterpret_cast<const f32x4*>(b32 + (int64_t)min(kk0 + d, max(kk1 - 1, kk0)) * 256);
      f32x4 av_n = *reinterpret_cast<const f32x4*>(a32 + 8 * min(kk0, max(kk1 - 1, 0)));
      for (int q0 = kk0; q0 < kk1; q0 += WD) {
#pragma unroll
        for (int d = 0; d < WD; ++d) {
          const int q = q0 + d;
          if (q >= kk1) break;  // (uniform)
          const f32x4 av = av_n;
          const f32x4 bv = bq[d];
          av_n = *reinterpret_cast<const f32x4*>(a32 + 8 * min(q + 1, kk1 - 1));
          bq[d] = *reinterpret_cast<const f32x4*>(b32 + (int64_t)min(q + WD, kk1 - 1) * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc2, 0, 0, 0);
        }
      }
    } else {
      const unsigned short* a_hi = A_hi + (lane & 31) * RS + 8 * (lane >> 5);
      const unsigned short* a_lo = A_lo + (lane & 31) * RS + 8 * (lane >> 5);
#pragma unroll 2
      for (int kk = kk0; kk < kk1; ++kk) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_hi + 16 * kk);
        bf16x8 al;
        if constexpr (TERMS == 3) al = *reinterpret_cast<const bf16x8*>(a_lo + 16 * kk);
        const int64_t off = ((int64_t)min(ct, NT - 1) * KS * 64 + lane) * 8 + (int64_t)kk * 512;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bhi + off);
        if constexpr (TERMS == 3) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Blo + off);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
        }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
      }
    }
    // the channels of the NEXT tile's first point (set 0 of b_r is free since point PPW - 2): its loads fly under the partial sums, the
    // epilogue and two barriers instead of opening the next tile with a bare round trip.  The neighbour index of lane (n16, q4) at
    // step u is the index lane 4 u + q4 fetched (jx_nxt[0]): one bpermute per step.
    if (PRE_B0) {
      b0_loaded = have_next;
      if (have_next) {
#pragma unroll
        for (int u = 0; u < kMaxSteps; ++u) {
          const int id = __builtin_amdgcn_ds_bpermute(4 * (4 * u + q4), jx_nxt[0]);
          load_feats(id, u < steps && 4 * u + q4 < H && id < Ns, b_r[0][u]);
        }
      }
    }
    KPF_STAMP(2);     // phase 2 (this wave's K range)
    __syncthreads();  // every wave has read its A fragments: the tile's memory takes the next channel block / the K partials
    KPF_STAMP(3);     // wait for the other waves' K ranges
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc2[r];  // (tile ct of K range kpart = partial index kpart CT + ct = wave)
    __syncthreads();
    KPF_STAMP(4);     // partials to LDS + barrier
    // ------------------------------------------------------------------ epilogue: sum the K partials, / count + bias, whole rows out
    for (int e = tid; e < kFusedRows * c_out; e += 64 * WAVES) {
      const int row = e / c_out, col = e - row * c_out;
      if (m0 + row >= M) continue;
      const int64_t m = row_s[row];
      const int t = col >> 5, cc = col & 31;
      // element (row, cc) of a 32 x 32 accumulator: lane = cc + 32 ((row >> 2) & 1), register = (row & 3) + 4 (row >> 3)
      const int src = ((row & 3) + 4 * (row >> 3)) * 64 + cc + 32 * ((row >> 2) & 1);
      float v = 0.f;
      for (int p = 0; p < KPARTS; ++p) v += part[(p * CT + t) * 16 * 64 + src];
      out[m * c_out + col] = v / (float)max(cnt_s[row], 1) + (bias ? bias[col] : 0.f);
    }
    KPF_STAMP(5);     // epilogue
    __syncthreads();  // partials, counts and rows are consumed before the next tile overwrites them
    KPF_STAMP(6);
    rows_cur = rows_nxt, rows_nxt = rows_nn;
#pragma unroll
    for (int p = 0; p < PPW; ++p) jx_cur[p] = jx_nxt[p], cn_cur[p] = cn_nxt[p], rx_cur[p][0] = rx_nxt[p][0], rx_cur[p][1] = rx_nxt[p][1], rx_cur[p][2] = rx_nxt[p][2];
  }
#ifdef GEOTR_KPF_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_kpf_stamps[i], stamp_acc[i]);
    atomicAdd(&g_kpf_stamps[7], (unsigned long long)(tile_end > tile_begin ? tile_end - tile_begin : 0));
  }
#endif
}

// ---- first layer (C_in = 1, kpconv.py:79-121 with a one-channel feature): the whole layer per point is 15 influence-weighted sums
// and a 15 x C_out product.  The two-kernel path ran a 16-lanes-per-point gather whose inner loop waited on two dependent global loads
// per neighbour, wrote (M, 15) to HBM and launched an exact-fp32 GEMM over it.  Here a wave takes 4 points at a time:
//   (1) all 4 H neighbour records (relative position, feature) are loaded with lanes <-> neighbours and parked in LDS;
//   (2) lanes <-> (point, kernel point): g[p][k] = sum_h w f as an fmaf chain over h in order (bitwise the gather kernel's sum);
//   (3) lanes <-> output channels: out[c] = (sum_k g[k] W[k][c] as an fmaf chain over k in order -- bitwise the fp32 MFMA GEMM's)
//       / max(#positive neighbours, 1) + bias, one coalesced row store per point.
constexpr int kC1Points = 4;  // points per wave iteration (16 lanes each in step 2)
__global__ __launch_bounds__(256) void kpconv_c1_fused_kernel(const float* __restrict__ feats, const float* __restrict__ qp,
                                                              const float* __restrict__ sp, const int64_t* __restrict__ nb,
                                                              const float* __restrict__ kp, int64_t M, int64_t Ns, int H, float sigma,
                                                              const float* __restrict__ W, const float* __restrict__ bias, int c_out,
                                                              const int* __restrict__ order, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c1sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int recs = kC1Points * H;                                   // <= 256
  float4* rec = reinterpret_cast<float4*>(c1sm) + wave * (256 + 16 + 4);  // [recs] (relative position, feature)
  float* g_s = reinterpret_cast<float*>(rec + 256);                 // [4][16] weighted sums
  int* cnt_s = reinterpret_cast<int*>(g_s + 64);                    // [4] positive-feature neighbour counts
  const float inv_sigma = 1.f / sigma;
  const int p16 = lane >> 4, k16 = lane & 15;
  const bool is_kp = k16 < 15;
  const float kx = is_kp ? kp[3 * k16] : 0.f, ky = is_kp ? kp[3 * k16 + 1] : 0.f, kz = is_kp ? kp[3 * k16 + 2] : 0.f;
  // groups of 4 consecutive points of the visiting order (see kpconv_fused_kernel); blocks take contiguous ranges, XCD by XCD
  auto load_rows = [&](int64_t grp) -> int {  // lane p < 4: the row of the group's p-th point, fetched one group ahead
    const int64_t t = min(grp * kC1Points + min(lane, kC1Points - 1), M - 1);
    return order ? order[t] : (int)t;
  };
  const int64_t groups = (M + kC1Points - 1) / kC1Points;
  const int lblock = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);  // gridDim.x % 8 == 0 (host)
  const int64_t per_block = (groups + gridDim.x - 1) / gridDim.x;
  const int64_t grp_end = min(groups, (int64_t)(lblock + 1) * per_block);
  int rows_cur = (int64_t)lblock * per_block + wave < grp_end ? load_rows((int64_t)lblock * per_block + wave) : 0;
  for (int64_t grp = (int64_t)lblock * per_block + wave; grp < grp_end; grp += 4) {
    const int64_t m0 = grp * kC1Points;
    const int rows_nxt = grp + 4 < grp_end ? load_rows(grp + 4) : 0;
    const int row4[kC1Points] = {__builtin_amdgcn_readlane(rows_cur, 0), __builtin_amdgcn_readlane(rows_cur, 1),
                                 __builtin_amdgcn_readlane(rows_cur, 2), __builtin_amdgcn_readlane(rows_cur, 3)};
    // (1) neighbour records; an absent neighbour (pad index, or a point past M) is marked by an infinite x offset
    for (int e = lane; e < recs; e += 64) {
      const int p = e / H, h = e - p * H;
      float4 r = make_float4(__int_as_float(0x7f800000), 0.f, 0.f, 0.f);
      if (m0 + p < M) {
        const int64_t m = p == 0 ? row4[0] : (p == 1 ? row4[1] : (p == 2 ? row4[2] : row4[3]));
        const int64_t j = nb[m * H + h];
        if (j < Ns) {
          r.x = sp[3 * j] - qp[3 * m], r.y = sp[3 * j + 1] - qp[3 * m + 1], r.z = sp[3 * j + 2] - qp[3 * m + 2];
          r.w = feats[j];
        }
      }
      rec[e] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (2) lane (p16, k16): kernel point k16 of point p16; lane k16 == 15 counts the neighbours with a positive feature instead
    {
      float acc = 0.f;
      int cnt = 0;
      const float4* rp = rec + p16 * H;
      for (int h = 0; h < H; ++h) {
        const float4 r = rp[h];
        if (r.x == __int_as_float(0x7f800000)) continue;  // absent: skipped as in the gather kernel, the order of the others is kept
        cnt += r.w > 0.f;
        const float dx = r.x - kx, dy = r.y - ky, dz = r.z - kz;
        const float w = fmaxf(1.f - __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz) * inv_sigma, 0.f);
        acc = fmaf(w, r.w, acc);
      }
      if (is_kp) g_s[p16 * 16 + k16] = acc;
      else cnt_s[p16] = cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) lanes <-> output channels
    for (int c = lane; c < c_out; c += 64) {
      float wk[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) wk[k] = W[k * c_out + c];
      const float b = bias ? bias[c] : 0.f;
#pragma unroll
      for (int p = 0; p < kC1Points; ++p) {
        if (m0 + p >= M) break;
        const int64_t m = row4[p];
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 15; ++k) v = fmaf(g_s[p * 16 + k], wk[k], v);
        out[m * c_out + c] = v / (float)max(cnt_s[p], 1) + b;
      }
    }
    __builtin_amdgcn_wave_barrier();
    rows_cur = rows_nxt;
  }
}

static inline int64_t pad32(int64_t x) { return (x + 31) / 32 * 32; }

}  // namespace geotr

using namespace geotr;

extern "C" {

#ifdef GEOTR_KPF_STAMPS
// measurement build only: reads and clears the section counters (7 sections of wave 0 in cycles, [7] = tiles)
int geotr_debug_kpf_stamps(unsigned long long* out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kpf_stamps), sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_kpf_stamps), zero, sizeof(zero)) != hipSuccess;
}
#endif

int geotr_kpconv_fused_supported(int64_t c_in, int64_t c_out, int64_t h) {
  if (!(c_in == 32 || c_in == 64) || c_out < 32 || c_out % 32 != 0 || h < 1 || h > 4 * kMaxSteps) return 0;
  const int waves = 8;
  const int64_t ct = c_out / 32;
  return ct <= waves && waves % ct == 0;
}

int geotr_kpconv_c1_fused(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                          const float* kernel_points, int64_t m, int64_t ns, int64_t h, int64_t c_out, int64_t num_kernel_points, float sigma,
                          const float* weights, const float* bias, const int32_t* order, float* out, void* stream_) {
  GEOTR_CHECK_ARG(m >= 0 && m < (1ll << 31) && ns >= 0 && h >= 1 && h <= 64 && c_out >= 1, "kpconv_c1_fused: bad sizes (h <= 64)");
  GEOTR_CHECK_ARG(num_kernel_points == 15, "kpconv_c1_fused: only 15 kernel points are supported (got %lld)", (long long)num_kernel_points);
  if (m == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(s_feats && q_points && s_points && neighbors && kernel_points && weights && out, "kpconv_c1_fused: null pointer");
  const size_t lds = 4 * (256 + 16 + 4) * sizeof(float4);
  const int64_t groups = (m + kC1Points - 1) / kC1Points;
  const unsigned grid = (unsigned)((std::min<int64_t>((groups + 3) / 4, 256 * 16) + 7) / 8 * 8);  // a multiple of 8: one share per XCD
  kpconv_c1_fused_kernel<<<dim3(grid), dim3(256), lds, (hipStream_t)stream_>>>(s_feats, q_points, s_points, neighbors, kernel_points, m, ns, (int)h,
                                                                             sigma, weights, bias, (int)c_out, order, out);
  GEOTR_CHECK_LAUNCH("kpconv_c1_fused");
  return GEOTR_OK;
}

int geotr_kpconv_fused(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                       const float* kernel_points, const uint8_t* pos_flag, int64_t m, int64_t ns, int64_t h, int64_t c_in, int64_t c_out,
                       int64_t num_kernel_points, float sigma, const void* packed, const float* bias, int bf16_operands,
                       const int32_t* order, float* out, void* stream_) {
  GEOTR_CHECK_ARG(m >= 0 && m < (1ll << 31) && ns >= 0, "kpconv_fused: bad sizes");
  GEOTR_CHECK_ARG(num_kernel_points == 15, "kpconv_fused: only 15 kernel points are supported (got %lld)", (long long)num_kernel_points);
  GEOTR_CHECK_ARG(geotr_kpconv_fused_supported(c_in, c_out, h), "kpconv_fused: unsupported shape (c_in %lld, c_out %lld, h %lld)",
                  (long long)c_in, (long long)c_out, (long long)h);
  if (m == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(s_feats && q_points && s_points && neighbors && kernel_points && pos_flag && packed && out, "kpconv_fused: null pointer");
  GEOTR_CHECK_ARG((reinterpret_cast<uintptr_t>(s_feats) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed) & 15) == 0,
                  "kpconv_fused: features and packed weights must be 16-byte aligned");
  GEOTR_CHECK_ARG(ns * c_in < (1ll << 30), "kpconv_fused: more than 2^30 feature elements (the feature rows are read as one 4 GB buffer)");
  GEOTR_CHECK_ARG(bf16_operands >= 0 && bf16_operands <= 2, "kpconv_fused: arithmetic mode must be 0 (split-bf16), 1 (bf16) or 2 (fp32)");
  if (const int rc = pack_format_check(packed, bf16_operands, "kpconv_fused")) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t kdim = 15 * c_in, kp_pad = pad32(kdim), np_pad = pad32(c_out);
  const unsigned short* bhi = reinterpret_cast<const unsigned short*>(packed);
  const unsigned short* blo = bhi + np_pad * kp_pad;
  const int KS = (int)(kp_pad / 16), NT = (int)(np_pad / 32);
  // (C_in = 64 as two channel blocks of 32 -- a 78 KB tile, two workgroups per CU -- was measured in round 6: 2 942 vs 2 815 us per
  // 16-pair stack alone, nothing end to end: profiles/r06_ab_runs.md section 1)
  const size_t lds = (size_t)2 * kFusedRows * (15 * c_in + 8) * 2 + 2 * kFusedRows * 4;
  const int64_t tiles = (m + kFusedRows - 1) / kFusedRows;
  // persistent: a few tiles per resident workgroup; a multiple of 8 blocks, one share of the tile range per XCD
  const unsigned grid = (unsigned)((std::min<int64_t>(tiles, 256 * (c_in == 32 ? 8 : 4)) + 7) / 8 * 8);
#define GEOTR_KPF(CC, WW, TT)                                                                                                          \
  do {                                                                                                                             \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kpconv_fused_kernel<CC, WW, TT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds) != hipSuccess)                                                                               \
      return fail(GEOTR_E_LAUNCH, "kpconv_fused: cannot reserve %zu B of LDS", lds);                                               \
    kpconv_fused_kernel<CC, WW, TT><<<dim3(grid), dim3(64 * WW), lds, stream>>>(s_feats, q_points, s_points, neighbors, kernel_points, pos_flag, m, \
                                                                               ns, (int)h, sigma, (int)c_out, KS, NT, bhi, blo, bias, order, out); \
  } while (0)
  if (c_in == 32) {  // 8 waves in both widths: two resident workgroups at c_in = 32 give 4 waves per SIMD, the LDS tile allows no more
    if (bf16_operands == 2) GEOTR_KPF(32, 8, 0);
    else if (bf16_operands == 1) GEOTR_KPF(32, 8, 1);
    else GEOTR_KPF(32, 8, 3);
  } else {
    if (bf16_operands == 2) GEOTR_KPF(64, 8, 0);
    else if (bf16_operands == 1) GEOTR_KPF(64, 8, 1);
    else GEOTR_KPF(64, 8, 3);
  }
#undef GEOTR_KPF
  GEOTR_CHECK_LAUNCH("kpconv_fused");
  return GEOTR_OK;
}

}  // extern "C"
