// HBM-bound elementwise / reduction kernels around the GEMM/SpMM pairs of the TeacherGNN step
// (TricksComb.forward GNN_model/GCN.py:103-138; run_trainSet trainer_node_classification.py:386-430).
// All fp32, 16-byte vector accesses when the data allows it, grid-stride loops capped at
// 256 CUs x 8 blocks, reductions in a fixed two-stage order (no float atomics -> bit-reproducible).
#include "cb_common.h"
#include "cb_philox.h"

namespace cb {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256 * 8;

static inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

// out[i] = x[i] * keep(offset + i) / (1 - p).  `offset` is the flat index of x[0] in the logical
// (unsharded) tensor, so a row shard draws the same mask as the full tensor would.
__global__ void __launch_bounds__(kBlock) k_dropout(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                    uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                    int64_t offset, int vec_ok) {
  if (seed_dev) seed += *seed_dev;   // hipGraph mode: the per-step part of the seed lives in device memory
  const int64_t nq = (n + 3) / 4;
  const int sub = (int)(offset & 3);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    float m[4];
    const int64_t gq = (offset >> 2) + q;
    keep4(seed, gq, thresh, scale, m);
    if (sub) {  // local quad straddles two global quads (compile-time shifts: no dynamically indexed register arrays)
      float m2[4];
      keep4(seed, gq + 1, thresh, scale, m2);
      const float e[8] = {m[0], m[1], m[2], m[3], m2[0], m2[1], m2[2], m2[3]};
      if (sub == 1) { m[0] = e[1]; m[1] = e[2]; m[2] = e[3]; m[3] = e[4]; }
      else if (sub == 2) { m[0] = e[2]; m[1] = e[3]; m[2] = e[4]; m[3] = e[5]; }
      else { m[0] = e[3]; m[1] = e[4]; m[2] = e[5]; m[3] = e[6]; }
    }
    const int64_t i = q * 4;
    if (vec_ok && i + 4 <= n) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      float4 o = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
      *reinterpret_cast<float4*>(out + i) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < n) out[i + k] = x[i + k] * m[k];
    }
  }
}

// out = a*x + b*y   (res_tricks.py:14,23: (1-alpha)*Xs[-1] + alpha*Xs[k])
__global__ void __launch_bounds__(kBlock) k_axpby(float a, const float* __restrict__ x, float b, const float* __restrict__ y,
                                                  float* __restrict__ out, int64_t n, int vec_ok) {
  const int64_t nq = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    if (vec_ok && i + 4 <= n) {
      float4 u = *reinterpret_cast<const float4*>(x + i);
      float4 w = *reinterpret_cast<const float4*>(y + i);
      *reinterpret_cast<float4*>(out + i) = make_float4(mix2(a, u.x, b, w.x), mix2(a, u.y, b, w.y), mix2(a, u.z, b, w.z), mix2(a, u.w, b, w.w));
    } else {
      for (int k = 0; k < 4; ++k)
        if (i + k < n) out[i + k] = mix2(a, x[i + k], b, y[i + k]);
    }
  }
}

// Backward of  Y = act(b * R + bias):  gm = g * (act > 0);  colsum(gm) -> dbias partials;  out = gm * row_scale.
// Block = 256 threads laid out as (rows_per_iter = 256 / tx) x (tx column groups of 4); each thread owns 4
// fixed columns per column pass so the column sums stay in registers; partial[blockIdx][d] is reduced by k_colsum_finish.
__global__ void __launch_bounds__(kBlock) k_act_bwd(const float* __restrict__ g, const float* __restrict__ act,
                                                    const float* __restrict__ row_scale, float* __restrict__ out,
                                                    int64_t rows, int d, float* __restrict__ partial) {
  extern __shared__ float s_red[];  // [kBlock][4]
  const int tx = min(64, (d + 3) / 4);       // threads along columns per pass
  const int ty = kBlock / tx;                // rows per iteration
  const int cx = threadIdx.x % tx, ry = threadIdx.x / tx;
  const bool live = ry < ty;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(rows, r_begin + rows_per_block);
  const bool vec_ok = (d % 4 == 0);
  for (int c_base = 0; c_base < d; c_base += tx * 4) {
    const int c = c_base + cx * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && c < d) {
      for (int64_t r = r_begin + ry; r < r_end; r += ty) {
        const int64_t off = r * d + c;
        const float sc = row_scale ? row_scale[r] : 1.f;
        float gv[4], av[4];
        if (vec_ok) {
          float4 t = *reinterpret_cast<const float4*>(g + off);
          gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
          if (act) {
            float4 u = *reinterpret_cast<const float4*>(act + off);
            av[0] = u.x; av[1] = u.y; av[2] = u.z; av[3] = u.w;
          }
        } else {
          for (int k = 0; k < 4; ++k) {
            gv[k] = (c + k < d) ? g[off + k] : 0.f;
            av[k] = (act && c + k < d) ? act[off + k] : 1.f;
          }
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gm = (!act || av[k] > 0.f) ? gv[k] : 0.f;
          s[k] += gm;
          o[k] = gm * sc;
        }
        if (out) {
          if (vec_ok) *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
          else
            for (int k = 0; k < 4; ++k)
              if (c + k < d) out[off + k] = o[k];
        }
      }
    }
    if (partial) {  // reduce over ry in a fixed order
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[threadIdx.x * 4 + k] = s[k];
      __syncthreads();
      if (ry == 0 && c < d) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < ty; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * tx + cx) * 4 + k];
        for (int k = 0; k < 4; ++k)
          if (c + k < d) partial[(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Backward of the fused aggregation epilogue of the residual trunk, one pass over [rows, d], d % 256 == 0:
//   gm = g * keep(seed, row0 + r, c) / (1 - p)            dropout backward (thresh == 0: gm = g)
//   gx0 = (accumulate ? gx0 : 0) + c_mix * gm             gradient flowing to the mixed-in tensor (X0)
//   gy  = c_act * gm * relu_bit(r, c)                      mix + ReLU backward (mask bits written by the forward)
//   colsum(gy) -> dbias partials;  out = gy * row_scale[r] (input of the reverse-graph aggregation)
// One wavefront per row per iteration, lane l owns columns 4l..4l+3 of each 256-wide tile (same mapping as
// the forward epilogue, so mask word k is tested at bit l).
// MODE 0: the layer kernel above.  MODE 1: trunk input stage  gy = (add + gm) * (act > 0); out = gy; colsum(gy).
template <int MODE, bool OUT_BF16, bool STORE = true, bool RIDX = false>      // STORE = false: column sums only (no output row is written); RIDX: compact rows (ridx)
__global__ void __launch_bounds__(kBlock) k_trunk_bwd(const float* __restrict__ g, const unsigned long long* __restrict__ bits,
                                                      const float* __restrict__ act, const float* __restrict__ row_scale,
                                                      void* __restrict__ outv, float* __restrict__ gx0, int accumulate,
                                                      int64_t rows, int d, uint32_t thresh, float keep_scale, uint64_t seed,
                                                      const uint64_t* __restrict__ seed_dev, int64_t row0, float c_act, float c_mix,
                                                      float* __restrict__ partial, const int64_t* __restrict__ ridx,
                                                      const float* __restrict__ g2, uint64_t seed2, float c2, const int* __restrict__ g2_pos) {
  // g2_pos (may be null; int32 per row of the FULL matrix): g2 is a COMPACT matrix — row rr of the full matrix sits at g2_pos[rr], absent (zero)
  // where that is negative (a row-sparse backward: g2 lives on the previous level's support)
  // g2 (MODE 0, dense rows; may be null): the 'Residual' connection (res_tricks.py:7-14) — this layer's ReLU output A_l is also the mix source
  // of layer l+1, so dL/dA_l = c_act * dropout_bwd_seed(g) + c2 * dropout_bwd_seed2(g2), g2 = the gradient w.r.t. layer l+1's stored (dropped)
  // output; `bits` must then be the ReLU mask alone (bits_relu_only of the forward store)
  // ridx (MODE 0, gx0 == NULL): g / out hold only the rows ridx[0 .. rows) of the matrix (the loss rows of a row-sparse backward); mask words,
  // row scale and the dropout mask are those of row ridx[r]
  extern __shared__ float s_red[];  // [4 waves][256 cols] per tile pass
  if (seed_dev) { seed += *seed_dev; seed2 += *seed_dev; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tiles = d >> 8;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(rows, r_begin + rows_per_block);
  for (int tile = 0; tile < tiles; ++tile) {
    const int c = tile * 256 + lane * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t rr_next = 0;      // RIDX: the index of the NEXT row is requested one iteration ahead (mask words and row scale hang on it)
    if constexpr (RIDX) {
      if (r_begin + w < r_end) rr_next = ridx[r_begin + w];
    }
    // g is read once (streaming); the NEXT row's 1 KiB is requested before this row is worked on: at full occupancy (8 wavefronts per SIMD) one
    // row in flight per wavefront keeps only 32 KB per CU outstanding — 4 TB/s at the ~2 us these loads take; two rows double that
    float gn[4] = {0.f, 0.f, 0.f, 0.f};
    if (r_begin + w < r_end) {
      const int64_t o0 = (r_begin + w) * d + c;
      gn[0] = __builtin_nontemporal_load(g + o0); gn[1] = __builtin_nontemporal_load(g + o0 + 1);
      gn[2] = __builtin_nontemporal_load(g + o0 + 2); gn[3] = __builtin_nontemporal_load(g + o0 + 3);
    }
    for (int64_t r = r_begin + w; r < r_end; r += kBlock / kWave) {
      const int64_t off = r * d + c;
      int64_t rr = r;      // the row of the full matrix this row is
      if constexpr (RIDX) {
        rr = rr_next;
        if (r + kBlock / kWave < r_end) rr_next = ridx[r + kBlock / kWave];
      }
      float gm[4] = {gn[0], gn[1], gn[2], gn[3]};
      // (this row's mask words and scale are requested BEFORE the next row's gradient: loads return in order, so waiting for them must not
      // mean waiting for the prefetch)
      unsigned long long bwr[4] = {0ull, 0ull, 0ull, 0ull};
      float sc_r = 1.f;
      if (MODE == 0) {
        const unsigned long long* bwp = bits + (rr * tiles + tile) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) bwr[k] = bwp[k];
        sc_r = row_scale ? row_scale[rr] : 1.f;
      }
      if (r + kBlock / kWave < r_end) {
        const int64_t o1 = off + (int64_t)(kBlock / kWave) * d;
        gn[0] = __builtin_nontemporal_load(g + o1); gn[1] = __builtin_nontemporal_load(g + o1 + 1);
        gn[2] = __builtin_nontemporal_load(g + o1 + 2); gn[3] = __builtin_nontemporal_load(g + o1 + 3);
      }
      if (thresh) {
        float m[4];
        keep4(seed, ((row0 + rr) * d + c) >> 2, thresh, keep_scale, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) gm[k] *= m[k];
      }
      float gy[4];
      if (MODE == 0) {
        if (gx0) {
          float4 a = accumulate ? *reinterpret_cast<const float4*>(gx0 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          a.x += c_mix * gm[0]; a.y += c_mix * gm[1]; a.z += c_mix * gm[2]; a.w += c_mix * gm[3];
          *reinterpret_cast<float4*>(gx0 + off) = a;
        }
        const unsigned long long* bw = bwr;
        if (g2) {      // (uniform) second gradient through the same ReLU, under the next layer's dropout mask
          const int64_t p2 = g2_pos ? (int64_t)__builtin_amdgcn_readfirstlane(g2_pos[rr]) : (RIDX ? rr : r);      // (wave-uniform row)
          const int64_t off2 = (p2 < 0 ? 0 : p2) * d + c;
          float g2m[4] = {__builtin_nontemporal_load(g2 + off2), __builtin_nontemporal_load(g2 + off2 + 1), __builtin_nontemporal_load(g2 + off2 + 2),
                          __builtin_nontemporal_load(g2 + off2 + 3)};
          if (p2 < 0) { g2m[0] = g2m[1] = g2m[2] = g2m[3] = 0.f; }
          if (thresh) {
            float m2[4];
            keep4(seed2, ((row0 + rr) * d + c) >> 2, thresh, keep_scale, m2);
#pragma unroll
            for (int k = 0; k < 4; ++k) g2m[k] *= m2[k];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) gy[k] = ((bw[k] >> lane) & 1ull) ? c_act * gm[k] + c2 * g2m[k] : 0.f;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) gy[k] = ((bw[k] >> lane) & 1ull) ? c_act * gm[k] : 0.f;
        }
      } else {
        const float4 a = *reinterpret_cast<const float4*>(gx0 + off);
        const float4 x = *reinterpret_cast<const float4*>(act + off);
        gy[0] = x.x > 0.f ? a.x + gm[0] : 0.f;
        gy[1] = x.y > 0.f ? a.y + gm[1] : 0.f;
        gy[2] = x.z > 0.f ? a.z + gm[2] : 0.f;
        gy[3] = x.w > 0.f ? a.w + gm[3] : 0.f;
      }
      const float sc = MODE == 0 ? sc_r : (row_scale ? row_scale[rr] : 1.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += gy[k];
      if constexpr (!STORE) continue;
      else if constexpr (OUT_BF16)
        *reinterpret_cast<uint2*>((bf16_t*)outv + off) = pack4_bf16(gy[0] * sc, gy[1] * sc, gy[2] * sc, gy[3] * sc);
      else
      {      // written once, gathered by the next kernel: streaming store
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t q = {gy[0] * sc, gy[1] * sc, gy[2] * sc, gy[3] * sc};
        __builtin_nontemporal_store(q, reinterpret_cast<f4_t*>((float*)outv + off));
      }
    }
    if (partial) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[(w * 64 + lane) * 4 + k] = s[k];
      __syncthreads();
      if (w == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < kBlock / kWave; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * 64 + lane) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) partial[(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
  }
}

// The layer kernel above (MODE 0, all rows, fp32 out) for LAYER 0 of the 'Initial' trunk, which also FOLDS the gradients that reach X0 through the mixes
// (round 6; the elementwise form of cb_spmm_csr_store_bwd_mix_f32's epilogue):
//   out_m = c_mix * ( keep(seed) * g  +  sum_q keep(seed_q) * g_q[pos_q[r] | r] )        (pos_q null: a dense operand; pos < 0: the row is absent)
// g is read here anyway; the input stage (cb_gemm_tn_instage_f32) then reads out_m instead of g and every g_q.
struct FoldOps {
  int n;
  const float* g[2];
  const int* pos[2];
  uint64_t seed[2];
  float* out_m;
  // optional second column sum (cs_partial non-null): over the rows, cs_c * dropout_bwd(g[cs_src]) where cs_bits (mask words of ANOTHER store, indexed by the
  // node row) has the element's bit — the bias gradient of the store whose backward left a reverse aggregation's epilogue (cb_spmm_csr_store_bwd_f32), which
  // cb_trunk_input_bwd_multi_cs_f32 took while the input stage was a pass
  int cs_src;
  const unsigned long long* cs_bits;
  float cs_c;
  float* cs_partial;
};
__global__ void __launch_bounds__(kBlock) k_trunk_bwd_fold(const float* __restrict__ g, const unsigned long long* __restrict__ bits, const float* __restrict__ row_scale,
                                                           float* __restrict__ out, FoldOps fo, int64_t rows, int d, uint32_t thresh, float keep_scale, uint64_t seed,
                                                           const uint64_t* __restrict__ seed_dev, int64_t row0, float c_act, float c_mix, float* __restrict__ partial) {
  extern __shared__ float s_red[];
  const uint64_t sd = seed_dev ? *seed_dev : 0ull;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tiles = d >> 8;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(rows, r_begin + rows_per_block);
  for (int tile = 0; tile < tiles; ++tile) {
    const int c = tile * 256 + lane * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float gn[4] = {0.f, 0.f, 0.f, 0.f};
    if (r_begin + w < r_end) {
      const int64_t o0 = (r_begin + w) * d + c;
      gn[0] = __builtin_nontemporal_load(g + o0); gn[1] = __builtin_nontemporal_load(g + o0 + 1);
      gn[2] = __builtin_nontemporal_load(g + o0 + 2); gn[3] = __builtin_nontemporal_load(g + o0 + 3);
    }
    for (int64_t r = r_begin + w; r < r_end; r += kBlock / kWave) {
      const int64_t off = r * d + c;
      float gm[4] = {gn[0], gn[1], gn[2], gn[3]};
      const unsigned long long* bwp = bits + (r * tiles + tile) * 4;
      unsigned long long bw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) bw[k] = bwp[k];
      const float sc = row_scale ? row_scale[r] : 1.f;
      int pq[2] = {-1, -1};      // (wave-uniform) row of operand q that holds node row r, or < 0
      float u[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < fo.n) {
          const int64_t p = fo.pos[q] ? (int64_t)__builtin_amdgcn_readfirstlane(fo.pos[q][r]) : r;
          pq[q] = p < 0 ? -1 : 0;
          if (p >= 0) {
            const float* gq = fo.g[q] + p * d + c;
#pragma unroll
            for (int k = 0; k < 4; ++k) u[q][k] = __builtin_nontemporal_load(gq + k);
          }
        }
      }
      if (r + kBlock / kWave < r_end) {
        const int64_t o1 = off + (int64_t)(kBlock / kWave) * d;
        gn[0] = __builtin_nontemporal_load(g + o1); gn[1] = __builtin_nontemporal_load(g + o1 + 1);
        gn[2] = __builtin_nontemporal_load(g + o1 + 2); gn[3] = __builtin_nontemporal_load(g + o1 + 3);
      }
      const int64_t quad = ((row0 + r) * d + c) >> 2;
      if (thresh) {
        float m[4];
        keep4(seed + sd, quad, thresh, keep_scale, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) gm[k] *= m[k];
      }
      float mm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < fo.n && pq[q] >= 0) {
          float mq[4] = {1.f, 1.f, 1.f, 1.f};
          if (thresh) keep4(fo.seed[q] + sd, quad, thresh, keep_scale, mq);
#pragma unroll
          for (int k = 0; k < 4; ++k) mm[k] += c_mix * (u[q][k] * mq[k]);
          if (fo.cs_partial && q == fo.cs_src) {      // (wave-uniform)
            const unsigned long long* bw2 = fo.cs_bits + (r * tiles + tile) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) s2[k] += ((bw2[k] >> lane) & 1ull) ? fo.cs_c * (u[q][k] * mq[k]) : 0.f;
          }
        }
      }
      float gy[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm[k] += c_mix * gm[k];
        gy[k] = ((bw[k] >> lane) & 1ull) ? c_act * gm[k] : 0.f;
        s[k] += gy[k];
      }
      typedef float f4_t __attribute__((ext_vector_type(4)));
      const f4_t qo = {gy[0] * sc, gy[1] * sc, gy[2] * sc, gy[3] * sc};
      __builtin_nontemporal_store(qo, reinterpret_cast<f4_t*>(out + off));
      const f4_t qm = {mm[0], mm[1], mm[2], mm[3]};
      __builtin_nontemporal_store(qm, reinterpret_cast<f4_t*>(fo.out_m + off));
    }
    if (partial) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[(w * 64 + lane) * 4 + k] = s[k];
      __syncthreads();
      if (w == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < kBlock / kWave; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * 64 + lane) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) partial[(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
    if (fo.cs_partial) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[(w * 64 + lane) * 4 + k] = s2[k];
      __syncthreads();
      if (w == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < kBlock / kWave; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * 64 + lane) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) fo.cs_partial[(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
  }
}

// Input stage of the trunk backward with the X0-gradient gathered in ONE pass instead of accumulated layer by layer:
//   gy = ( keep(seed, r, c) * g  +  c_mix * sum_l keep(seed_l, r, c) * g_l ) / (1 - p)  *  (act > 0)
// g = gradient w.r.t. the dropped X0 that feeds layer 0; g_l = gradient w.r.t. the output of layer l's fused store (the mix
// (1-a) relu(Y_l) + a X0 sits under that store's dropout).  Replaces n_mix read-modify-write passes over a [rows, d]
// accumulator (20 B/element each) by n_mix streaming reads (4 B/element each); masks are regenerated, never stored.
constexpr int kMixMax = 7;
struct MixTable {
  const float* g[kMixMax];
  uint64_t seed[kMixMax];
  int n;
  const int* pos[kMixMax];      // null, or [rows]: g[l] is a COMPACT matrix that holds only some rows (a row-sparse backward's support rows) — row r
                                // sits at pos[l][r], absent (zero) where that is negative
  // second column sum (cs_partial non-null): sum over the rows of cs_c * dropout_bwd(g[cs_src]) where cs_bits has the element's bit — the bias gradient of
  // the store whose backward left the reverse aggregation's own epilogue (cb_spmm_csr_store_bwd_f32)
  // (up to two such sums per launch; a compact operand's mask words are still indexed by the node row)
  int cs_src[2];
  const unsigned long long* cs_bits[2];
  float cs_c[2];
  float* cs_partial[2];
};

template <int NMIX>   // number of mixed-in gradients, compile-time so that all row loads are issued before the first Philox round
__global__ void __launch_bounds__(kBlock) k_trunk_input_bwd_multi(const float* __restrict__ g, MixTable mt, const float* __restrict__ act,
                                                                  const unsigned long long* __restrict__ act_bits,
                                                                  float* __restrict__ out, int64_t rows, int d, uint32_t thresh,
                                                                  float keep_scale, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                                  int64_t row0, float c_mix, float* __restrict__ partial) {
  extern __shared__ float s_red[];
  const uint64_t sd = seed_dev ? *seed_dev : 0ull;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tiles = d >> 8;
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = min(rows, r_begin + rows_per_block);
  for (int tile = 0; tile < tiles; ++tile) {
    const int c = tile * 256 + lane * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, s2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int64_t r = r_begin + w; r < r_end; r += kBlock / kWave) {
      const int64_t off = r * d + c;
      const int64_t quad = ((row0 + r) * d + c) >> 2;
      float t[4] = {__builtin_nontemporal_load(g + off), __builtin_nontemporal_load(g + off + 1), __builtin_nontemporal_load(g + off + 2),
                    __builtin_nontemporal_load(g + off + 3)};
      float u[NMIX > 0 ? NMIX : 1][4];
      int pl[NMIX > 0 ? NMIX : 1];      // (wave-uniform) position of row r in a compact operand, or < 0; 0 for a dense operand
#pragma unroll
      for (int l = 0; l < NMIX; ++l) pl[l] = mt.pos[l] ? __builtin_amdgcn_readfirstlane(mt.pos[l][r]) : 0;
#pragma unroll
      for (int l = 0; l < NMIX; ++l) {
        const float* gl = mt.pos[l] ? mt.g[l] + ((int64_t)max(pl[l], 0) * d + c) : mt.g[l] + off;
#pragma unroll
        for (int k = 0; k < 4; ++k) u[l][k] = __builtin_nontemporal_load(gl + k);
      }
#pragma unroll
      for (int l = 0; l < NMIX; ++l) {
        if (pl[l] < 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) u[l][k] = 0.f;
        }
      }
      if (thresh) {
        float m[4];
        keep4(seed + sd, quad, thresh, keep_scale, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] *= m[k];
      }
#pragma unroll
      for (int l = 0; l < NMIX; ++l) {
        if (thresh && pl[l] >= 0) {      // (an absent row of a compact operand is zero whatever its mask)
          float m[4];
          keep4(mt.seed[l] + sd, quad, thresh, keep_scale, m);
#pragma unroll
          for (int k = 0; k < 4; ++k) u[l][k] *= m[k];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (mt.cs_partial[q] && l == mt.cs_src[q]) {      // (wave-uniform) this operand's masked gradient through the store's mask words
            const unsigned long long* bw2 = mt.cs_bits[q] + (r * tiles + tile) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) s2[q][k] += ((bw2[k] >> lane) & 1ull) ? mt.cs_c[q] * u[l][k] : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += c_mix * u[l][k];
      }
      float gy[4];
      if (act_bits) {      // mask words of (act > 0) instead of act itself: word k of (row, tile), bit `lane` <-> column 256 tile + 4 lane + k
        const unsigned long long* bw = act_bits + (r * tiles + tile) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) gy[k] = ((bw[k] >> lane) & 1ull) ? t[k] : 0.f;
      } else {
        const float4 x = *reinterpret_cast<const float4*>(act + off);
        gy[0] = x.x > 0.f ? t[0] : 0.f; gy[1] = x.y > 0.f ? t[1] : 0.f; gy[2] = x.z > 0.f ? t[2] : 0.f; gy[3] = x.w > 0.f ? t[3] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += gy[k];
      {      // written once, streamed by the weight-gradient GEMM that follows
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t q = {gy[0], gy[1], gy[2], gy[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<f4_t*>(out + off));
      }
    }
    if (partial) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[(w * 64 + lane) * 4 + k] = s[k];
      __syncthreads();
      if (w == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < kBlock / kWave; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * 64 + lane) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) partial[(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!mt.cs_partial[q]) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) s_red[(w * 64 + lane) * 4 + k] = s2[q][k];
      __syncthreads();
      if (w == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < kBlock / kWave; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += s_red[(j * 64 + lane) * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) mt.cs_partial[q][(int64_t)blockIdx.x * d + c + k] = t[k];
      }
      __syncthreads();
    }
  }
}

// out[i, :] = src[idx[i], :]  — packs the rows a peer rank asked for (halo exchange of the node-sharded path)
__global__ void __launch_bounds__(kBlock) k_gather_rows(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ idx,
                                                        int64_t n_idx, int d, float* __restrict__ out, int vec_ok) {
  if (vec_ok) {
    const int q = d >> 2;
    const int64_t total = n_idx * q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / q;
      const int c = (int)(i - r * q) * 4;
      *reinterpret_cast<float4*>(out + r * d + c) = *reinterpret_cast<const float4*>(src + idx[r] * ld + c);
    }
  } else {
    const int64_t total = n_idx * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / d;
      out[i] = src[idx[r] * ld + (i - r * d)];
    }
  }
}

// out[r, :] = pos[r] >= 0 ? src[pos[r], :] : fill — the inverse of the row pack: a compact [n, d] matrix over a row subset written back to all
// N rows in one pass (fill = 0: the table gradient dL/dZ_l of a compact level of the row-sparse backward; fill = NaN: the logits of a rows-only
// training forward, whose other rows nobody may read; trunk.py).  float4 rows.
__global__ void __launch_bounds__(kBlock) k_expand_rows(const float* __restrict__ src, const int* __restrict__ pos, int64_t n_rows, int d,
                                                        float fill, float* __restrict__ out) {
  const int q = d >> 2;
  const int64_t total = n_rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    const int p = pos[r];
    float4 v = make_float4(fill, fill, fill, fill);
    if (p >= 0) v = *reinterpret_cast<const float4*>(src + (int64_t)p * d + c);
    __builtin_nontemporal_store(v.x, out + r * d + c);
    __builtin_nontemporal_store(v.y, out + r * d + c + 1);
    __builtin_nontemporal_store(v.z, out + r * d + c + 2);
    __builtin_nontemporal_store(v.w, out + r * d + c + 3);
  }
}

// The same row pack with the rows narrowed to bf16 (round-to-nearest-even) on their way out: the bf16 halo wire of the node-sharded
// exchange leaves the pack kernel ready to send (no separate conversion pass over the packed rows).
__global__ void __launch_bounds__(kBlock) k_gather_rows_bf16(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ idx,
                                                             int64_t n_idx, int d, bf16_t* __restrict__ out, int vec_ok) {
  if (vec_ok) {
    const int q = d >> 2;
    const int64_t total = n_idx * q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / q;
      const int c = (int)(i - r * q) * 4;
      const float4 v = *reinterpret_cast<const float4*>(src + idx[r] * ld + c);
      *reinterpret_cast<uint2*>(out + r * d + c) = pack4_bf16(v.x, v.y, v.z, v.w);
    }
  } else {
    const int64_t total = n_idx * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / d;
      out[i] = f32_to_bf16(src[idx[r] * ld + (i - r * d)]);
    }
  }
}

// out[c] = sum_p partial[p][c]: one block per column, strided partial sums per thread then a fixed-order
// LDS tree — the result does not depend on scheduling.
__global__ void __launch_bounds__(kBlock) k_colsum_finish(const float* __restrict__ partial, int nparts, int d, float* __restrict__ out) {
  __shared__ float s_t[kBlock];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < nparts; p += kBlock) s += partial[(int64_t)p * d + c];
  s_t[threadIdx.x] = s;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_t[threadIdx.x] += s_t[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = s_t[0];
}

// sum of squares -> partial[block] (double accumulation across the partials in the finish kernel)
__global__ void __launch_bounds__(kBlock) k_sumsq(const float* __restrict__ x, int64_t n, float* __restrict__ partial, int vec_ok) {
  __shared__ float s_w[kBlock / kWave];
  float s = 0.f;
  const int64_t nq = (n + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    if (vec_ok && i + 4 <= n) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int k = 0; k < 4; ++k)
        if (i + k < n) s += x[i + k] * x[i + k];
    }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += s_w[w];
    partial[blockIdx.x] = t;
  }
}

// out[0] = sqrt(sum partial) (Frobenius norm, th.norm(self.le) GCN.py:232); out[1] = sum
__global__ void k_norm_finish(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
  double t = 0.0;
  for (int p = threadIdx.x; p < nparts; p += kWave) t += (double)partial[p];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  if (threadIdx.x == 0) {
    out[0] = (float)sqrt(t);
    out[1] = (float)t;
  }
}

// Fused log_softmax + nll_loss(mean over masked rows) forward AND its gradient
// (trainer_node_classification.py:390-391).  One lane per row; C is small (<= 256).
//   loss_partial[block] = sum_{r in block, mask[r]} (logsumexp(z_r) - z_r[y_r])
//   grad[r, c] = mask[r] ? (softmax(z_r)[c] - [c == y_r]) * inv_count : 0
// (A sub-wave-group-per-row variant with consecutive addresses inside a row measured slower at C = 40: 1.91 vs 1.59 ms on
// 10^7 rows — the 160-byte rows of neighbouring lanes already share cache lines.)
__global__ void __launch_bounds__(kBlock) k_nll_fused(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
                                                      const uint8_t* __restrict__ mask, int64_t rows, int C, float inv_count,
                                                      float* __restrict__ grad, float* __restrict__ loss_partial) {
  __shared__ float s_w[kBlock / kWave];
  float local = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float* zr = z + r * ld;
    float* gr = grad ? grad + r * (int64_t)C : nullptr;
    if (mask && !mask[r]) {
      if (gr)
        for (int c = 0; c < C; ++c) gr[c] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, zr[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(zr[c] - mx);
    const float lse = mx + logf(se);
    const int64_t t = y[r];
    local += lse - zr[t];
    if (gr) {
      const float inv = 1.f / se;
      for (int c = 0; c < C; ++c) gr[c] = (expf(zr[c] - mx) * inv - (c == t ? 1.f : 0.f)) * inv_count;
    }
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += s_w[w];
    loss_partial[blockIdx.x] = t;
  }
}

// The same arithmetic, expression for expression, with a row of C = 4 NV floats (NV <= 16, 16-byte aligned rows) held in registers: ten
// float4 loads and ten float4 stores per row at C = 40 instead of 120 + 40 scalar ones, every exp() evaluated once (1.64 -> 0.82 ms on 10^7 rows; same loss bits).
template <int NV>
__global__ void __launch_bounds__(kBlock) k_nll_fused_v4(const float* __restrict__ z, int64_t ld, const int64_t* __restrict__ y,
                                                         const uint8_t* __restrict__ mask, int64_t rows, float inv_count,
                                                         float* __restrict__ grad, float* __restrict__ loss_partial) {
  constexpr int C = 4 * NV;
  __shared__ float s_w[kBlock / kWave];
  float local = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    float4* gr = grad ? reinterpret_cast<float4*>(grad + r * (int64_t)C) : nullptr;
    if (mask && !mask[r]) {
      if (gr) {
#pragma unroll
        for (int q = 0; q < NV; ++q) gr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const float4* zr = reinterpret_cast<const float4*>(z + r * ld);
    float v[C];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const float4 t4 = zr[q];
      v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, v[c]);
    const int t = (int)y[r];
    float se = 0.f, zt = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      zt = c == t ? v[c] : zt;
      v[c] = expf(v[c] - mx);
      se += v[c];
    }
    const float lse = mx + logf(se);
    local += lse - zt;
    if (gr) {
      const float inv = 1.f / se;
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (v[4 * q + k] * inv - (4 * q + k == t ? 1.f : 0.f)) * inv_count;
        gr[q] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += s_w[w];
    loss_partial[blockIdx.x] = t;
  }
}

// one wavefront: lane l sums partials l, l+64, ... in double, then a fixed-order butterfly (deterministic)
__device__ __forceinline__ double wave_sum_partials(const float* __restrict__ partial, int nparts) {
  double t = 0.0;
  for (int p = threadIdx.x; p < nparts; p += kWave) t += (double)partial[p];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  return t;
}

__global__ void k_loss_finish(const float* __restrict__ partial, int nparts, float inv_count, float* __restrict__ out) {
  const double t = wave_sum_partials(partial, nparts);
  if (threadIdx.x == 0) out[0] = (float)(t * (double)inv_count);
}

// torch.optim.Adam semantics (trainer_node_classification.py:310): g += wd*p; m,v EMA; bias-corrected step
__global__ void __launch_bounds__(kBlock) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                 float wd, float bc1, float bc2_sqrt, const int64_t* __restrict__ step_dev, int vec_ok) {
  const int64_t nq = (n + 3) / 4;
  if (step_dev) {   // hipGraph mode: the step count lives in device memory, bias corrections are derived here
    const double t = (double)*step_dev;
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  const float step = lr / bc1;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = vec_ok && i + 4 <= n;
    if (full) {
      float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
      float4 c = *reinterpret_cast<const float4*>(m + i), d = *reinterpret_cast<const float4*>(v + i);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool in = i + k < n;
        pv[k] = in ? p[i + k] : 0.f; gv[k] = in ? g[i + k] : 0.f; mv[k] = in ? m[i + k] : 0.f; vv[k] = in ? v[i + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gv[k] + wd * pv[k];
      mv[k] = b1 * mv[k] + (1.f - b1) * gg;
      vv[k] = b2 * vv[k] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      pv[k] = pv[k] - step * (mv[k] / denom);
    }
    if (full) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (int k = 0; k < 4; ++k)
        if (i + k < n) { p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k]; }
    }
  }
}

// All parameter tensors of the model in ONE launch: blockIdx.y selects the tensor, blockIdx.x strides over its elements.
// The table travels by value in the kernel arguments (no device-side table to keep alive, capturable in a hipGraph).
constexpr int kAdamMax = 24;
struct AdamTable {
  float* p[kAdamMax];
  const float* g[kAdamMax];
  float* m[kAdamMax];
  float* v[kAdamMax];
  int64_t n[kAdamMax];
  const float* c[kAdamMax];   // per-tensor extra L2 coefficient read from device memory (null: none), added to weight_decay
  float* sq[kAdamMax];        // per-tensor partial sums of squares of the UPDATED parameter, one float per block of the launch (null: not wanted)
};

__global__ void __launch_bounds__(kBlock) k_adam_multi(AdamTable t, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                       float bc2_sqrt, const int64_t* __restrict__ step_dev, const int32_t* __restrict__ guard) {
  // a gradient check of this step failed (cb_rows_zero_outside_mask_f32 set the word): parameters and moments stay as they are
  if (guard && *guard != 0) return;
  const int ti = blockIdx.y;
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  const int64_t n = t.n[ti];
  if (step_dev) {
    const double s = (double)*step_dev;
    bc1 = (float)(1.0 - pow((double)b1, s));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, s));
  }
  const float step = lr / bc1;
  if (t.c[ti]) {
    const float c = *t.c[ti];
    if (isfinite(c)) wd += c;    // se_reg / ||le|| with ||le|| == 0: no regulariser gradient (torch.norm's subgradient at 0)
  }
  const bool vec_ok = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) == 0;
  const int64_t nq = (n + 3) / 4;
  float ssq = 0.f;      // sum of squares of the updated values this thread wrote: k_sumsq's thread-to-element map and summation order
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q * 4;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = vec_ok && i + 4 <= n;
    if (full) {
      const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
      const float4 c = *reinterpret_cast<const float4*>(m + i), d = *reinterpret_cast<const float4*>(v + i);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool in = i + k < n;
        pv[k] = in ? p[i + k] : 0.f; gv[k] = in ? g[i + k] : 0.f; mv[k] = in ? m[i + k] : 0.f; vv[k] = in ? v[i + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // same arithmetic as k_adam
      const float gg = gv[k] + wd * pv[k];
      mv[k] = b1 * mv[k] + (1.f - b1) * gg;
      vv[k] = b2 * vv[k] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      pv[k] = pv[k] - step * (mv[k] / denom);
    }
    if (full) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      ssq += pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2] + pv[3] * pv[3];
    } else {
      for (int k = 0; k < 4; ++k)
        if (i + k < n) { p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k]; ssq += pv[k] * pv[k]; }
    }
  }
  if (t.sq[ti]) {      // (uniform over the block) ||p||_F^2 of the updated tensor as k_sumsq would leave it: the next forward's th.norm(le) for free
    __shared__ float s_w[kBlock / kWave];
    for (int off = 32; off > 0; off >>= 1) ssq += __shfl_xor(ssq, off);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = ssq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tt = 0.f;
      for (int w = 0; w < kBlock / kWave; ++w) tt += s_w[w];
      t.sq[ti][blockIdx.x] = tt;
    }
  }
}

static inline int aligned16(const void* a) { return ((uintptr_t)a % 16) == 0; }

}  // namespace cb

using namespace cb;

extern "C" int cb_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, int64_t offset,
                              void* stream) {
  CB_CHECK_ARG(n >= 0 && offset >= 0 && (n == 0 || (x && out)) && p >= 0.f && p < 1.f, CB_E_INVALID,
               "cb_dropout_f32: bad argument (p=%f)", p);
  if (n == 0) return CB_OK;
  const uint32_t thresh = dropout_threshold(p);
  const int vec_ok = aligned16(x) && aligned16(out);
  hipLaunchKernelGGL(k_dropout, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, x, out, n, thresh,
                     1.f / (1.f - p), seed, seed_dev, offset, vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_axpby_f32(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream) {
  CB_CHECK_ARG(n >= 0 && (n == 0 || (x && y && out)), CB_E_INVALID, "cb_axpby_f32: bad argument");
  if (n == 0) return CB_OK;
  const int vec_ok = aligned16(x) && aligned16(y) && aligned16(out);
  hipLaunchKernelGGL(k_axpby, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, a, x, b, y, out, n, vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_colsum_workspace_bytes(int64_t rows, int64_t d) {
  if (rows <= 0 || d <= 0) return 0;
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  return (size_t)nb * (size_t)d * sizeof(float);
}

extern "C" int cb_act_bwd_f32(const float* g, const float* act, const float* row_scale, float* out, int64_t rows, int64_t d,
                              float* colsum, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d >= 0 && d < (1 << 20), CB_E_INVALID, "cb_act_bwd_f32: bad size");
  if (rows == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(g && (out || colsum), CB_E_INVALID, "cb_act_bwd_f32: null pointer");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(rows, d)), CB_E_WORKSPACE,
               "cb_act_bwd_f32: workspace too small (%zu < %zu)", ws_bytes, cb_colsum_workspace_bytes(rows, d));
  CB_CHECK_ARG(d % 4 != 0 || (aligned16(g) && (!act || aligned16(act)) && (!out || aligned16(out))), CB_E_INVALID,
               "cb_act_bwd_f32: 16-byte alignment required when d %% 4 == 0");
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)nb), dim3(kBlock), kBlock * 4 * sizeof(float), st, g, act, row_scale, out, rows,
                     (int)d, colsum ? (float*)ws : nullptr);
  CB_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)ws, (int)nb, (int)d, colsum);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// Rows of g outside `mask` must be exactly zero (the promise a row-sparse backward rests on: the loss_rows argument of the forward, ops.py / trunk.py).  A streaming pass
// over the matrix (contiguous rows: float4 per thread, the row of an element by one division); a violation is recorded in the device error
// word (never silent: cb_device_status reports it) and, if given, in the caller's `guard` word in device memory: an optimiser launch that
// follows on the same stream and is handed the same word leaves parameters and moments untouched (the truncated gradients never reach them).
template <bool VEC4>
__global__ void __launch_bounds__(kBlock) k_rows_zero_check(const float* __restrict__ g, int64_t ld, int64_t rows, int d, const uint8_t* __restrict__ mask,
                                                            int* __restrict__ err, int32_t* __restrict__ guard) {
  const int64_t per_row = VEC4 ? d / 4 : d;
  const int64_t n = rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / per_row, c = (i - r * per_row) * (VEC4 ? 4 : 1);
    bool bad;
    if constexpr (VEC4) {
      const float4 v = *reinterpret_cast<const float4*>(g + r * ld + c);
      bad = v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
    } else {
      bad = g[r * ld + c] != 0.f;
    }
    if (bad && !mask[r]) {
      __hip_atomic_store(err + 1, (int)(r & 0x7fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(err + 2, (int)(r >> 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(err, CB_DEVERR_GRADROWS, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (guard) *guard = 1;      // device memory: read by the Adam launch that follows on the stream (cb_adam_multi_norm_f32, `guard`)
    }
  }
}

extern "C" int cb_rows_zero_outside_mask_f32(const float* g, int64_t ld, int64_t rows, int64_t d, const uint8_t* mask, int32_t* guard, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d >= 0 && d < (1 << 20) && ld >= d, CB_E_INVALID, "cb_rows_zero_outside_mask_f32: bad size");
  if (rows == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(g && mask, CB_E_INVALID, "cb_rows_zero_outside_mask_f32: null pointer");
  int* err = device_error_word();
  CB_CHECK_ARG(err != nullptr, CB_E_HIP, "cb_rows_zero_outside_mask_f32: the device error word could not be allocated (%s)", cb_last_error());
  if (d % 4 == 0 && ld % 4 == 0 && aligned16(g))
    hipLaunchKernelGGL((k_rows_zero_check<true>), dim3((unsigned)grid_for(rows * (d / 4))), dim3(kBlock), 0, (hipStream_t)stream, g, ld, rows, (int)d, mask, err, guard);
  else
    hipLaunchKernelGGL((k_rows_zero_check<false>), dim3((unsigned)grid_for(rows * d)), dim3(kBlock), 0, (hipStream_t)stream, g, ld, rows, (int)d, mask, err, guard);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_reduce_workspace_bytes(void) { return (size_t)kMaxBlocks * sizeof(float); }

extern "C" int cb_frobenius_norm_f32(const float* x, int64_t n, float* out2, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(n >= 0 && out2 && (n == 0 || x), CB_E_INVALID, "cb_frobenius_norm_f32: bad argument");
  CB_CHECK_ARG(ws && ws_bytes >= cb_reduce_workspace_bytes(), CB_E_WORKSPACE, "cb_frobenius_norm_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = n ? grid_for((n + 3) / 4) : 0;
  if (nb) {
    hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(kBlock), 0, st, x, n, (float*)ws, aligned16(x));
    CB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(64), 0, st, (const float*)ws, nb, out2);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_nll_logsoftmax_f32(const float* logits, int64_t ld, const int64_t* y, const uint8_t* mask, int64_t rows,
                                     int64_t C, int64_t count, float* loss, float* grad, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && C > 0 && C <= 4096 && ld >= C && loss && (rows == 0 || (logits && y)), CB_E_INVALID,
               "cb_nll_logsoftmax_f32: bad argument");
  CB_CHECK_ARG(ws && ws_bytes >= cb_reduce_workspace_bytes(), CB_E_WORKSPACE, "cb_nll_logsoftmax_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const float inv = count > 0 ? 1.f / (float)count : 0.f;
  const int nb = rows ? grid_for(rows) : 0;
  if (nb) {
    const bool v4 = C % 4 == 0 && C <= 64 && ld % 4 == 0 && aligned16(logits) && (!grad || aligned16(grad));
#define CB_NLL_V4(NV_) hipLaunchKernelGGL((k_nll_fused_v4<NV_>), dim3(nb), dim3(kBlock), 0, st, logits, ld, y, mask, rows, inv, grad, (float*)ws)
    if (v4 && C == 40) CB_NLL_V4(10);
    else if (v4 && C == 48) CB_NLL_V4(12);
    else if (v4 && C == 8) CB_NLL_V4(2);
    else if (v4 && C == 4) CB_NLL_V4(1);
    else if (v4 && C == 64) CB_NLL_V4(16);
    else hipLaunchKernelGGL(k_nll_fused, dim3(nb), dim3(kBlock), 0, st, logits, ld, y, mask, rows, (int)C, inv, grad, (float*)ws);
#undef CB_NLL_V4
    CB_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, st, (const float*)ws, nb, inv, loss);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int64_t step, const int64_t* step_dev, void* stream) {
  CB_CHECK_ARG(n >= 0 && (step >= 1 || step_dev) && (n == 0 || (p && g && m && v)), CB_E_INVALID, "cb_adam_step_f32: bad argument");
  if (step < 1) step = 1;
  if (n == 0) return CB_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int vec_ok = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
  hipLaunchKernelGGL(k_adam, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, (float)bc1, (float)sqrt(bc2), step_dev, vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_adam_norm_workspace_bytes(int32_t n_norms) { return (size_t)(n_norms > 0 ? n_norms : 0) * cb_reduce_workspace_bytes(); }

extern "C" int cb_adam_multi_norm_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                      const int64_t* numel, const float* const* extra_decay, float* const* norm_out, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int64_t step, const int64_t* step_dev, const int32_t* guard,
                                      void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(n_tensors >= 0 && (step >= 1 || step_dev) && (n_tensors == 0 || (p && g && m && v && numel)), CB_E_INVALID,
               "cb_adam_multi_f32: bad argument");
  int n_norms = 0;
  if (norm_out)
    for (int i = 0; i < n_tensors; ++i) n_norms += norm_out[i] != nullptr;
  CB_CHECK_ARG(n_norms == 0 || (ws && ws_bytes >= cb_adam_norm_workspace_bytes(n_norms)), CB_E_WORKSPACE,
               "cb_adam_multi_norm_f32: workspace too small for %d norms", n_norms);
  int norm_slot = 0;
  if (step < 1) step = 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  for (int base = 0; base < n_tensors; base += kAdamMax) {
    AdamTable t{};
    const int cnt = n_tensors - base < kAdamMax ? n_tensors - base : kAdamMax;
    int64_t nmax = 0;
    for (int i = 0; i < cnt; ++i) {
      CB_CHECK_ARG(numel[base + i] >= 0 && (numel[base + i] == 0 || (p[base + i] && g[base + i] && m[base + i] && v[base + i])), CB_E_INVALID,
                   "cb_adam_multi_f32: null tensor %d", base + i);
      t.p[i] = p[base + i]; t.g[i] = g[base + i]; t.m[i] = m[base + i]; t.v[i] = v[base + i]; t.n[i] = numel[base + i];
      t.c[i] = extra_decay ? extra_decay[base + i] : nullptr;
      if (norm_out && norm_out[base + i]) t.sq[i] = (float*)ws + (size_t)(norm_slot++) * kMaxBlocks;
      if (t.n[i] > nmax) nmax = t.n[i];
    }
    const int nb = nmax ? grid_for((nmax + 3) / 4) : 0;
    if (nb) {
      hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)nb, (unsigned)cnt), dim3(kBlock), 0, (hipStream_t)stream, t, lr,
                         beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), step_dev, guard);
      CB_LAUNCH_CHECK();
    }
    for (int i = 0; i < cnt; ++i)      // out[0] = ||p||_F, out[1] = ||p||_F^2 (cb_frobenius_norm_f32's pair) of every tensor that asked
      if (t.sq[i]) {
        hipLaunchKernelGGL(k_norm_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)t.sq[i], nb, norm_out[base + i]);
        CB_LAUNCH_CHECK();
      }
  }
  return CB_OK;
}

extern "C" int cb_adam_multi_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                 const int64_t* numel, const float* const* extra_decay, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int64_t step, const int64_t* step_dev, const int32_t* guard, void* stream) {
  return cb_adam_multi_norm_f32(n_tensors, p, g, m, v, numel, extra_decay, nullptr, lr, beta1, beta2, eps, weight_decay, step, step_dev, guard, nullptr,
                                0, stream);
}

static int launch_trunk_bwd(int mode, int out_bf16, const float* g, const uint64_t* bits, const float* act, const float* row_scale,
                            void* out, float* gx0, int accumulate, int64_t rows, int64_t d, float drop_p, uint64_t seed,
                            const uint64_t* seed_dev, int64_t row0, float c_act, float c_mix, float* colsum, void* ws, size_t ws_bytes,
                            hipStream_t st, const int64_t* ridx = nullptr, const float* g2 = nullptr, uint64_t seed2 = 0, float c2 = 0.f,
                            const int32_t* g2_pos = nullptr) {
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  const uint32_t thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  const float ks = 1.f / (1.f - drop_p);
  float* partial = colsum ? (float*)ws : nullptr;
#define CB_TB_ARGS g, (const unsigned long long*)bits, act, row_scale, out, gx0, accumulate, rows, (int)d, thresh, ks, seed, seed_dev, row0, c_act, c_mix, partial, ridx, g2, seed2, c2, g2_pos
  const dim3 grid((unsigned)nb), blk(kBlock);
  const size_t sh = kBlock * 4 * sizeof(float);
  if (mode == 0 && ridx) hipLaunchKernelGGL((k_trunk_bwd<0, false, true, true>), grid, blk, sh, st, CB_TB_ARGS);
  else if (mode == 0 && !out) hipLaunchKernelGGL((k_trunk_bwd<0, false, false>), grid, blk, sh, st, CB_TB_ARGS);
  else if (mode == 0 && out_bf16) hipLaunchKernelGGL((k_trunk_bwd<0, true>), grid, blk, sh, st, CB_TB_ARGS);
  else if (mode == 0) hipLaunchKernelGGL((k_trunk_bwd<0, false>), grid, blk, sh, st, CB_TB_ARGS);
  else hipLaunchKernelGGL((k_trunk_bwd<1, false>), grid, blk, sh, st, CB_TB_ARGS);
#undef CB_TB_ARGS
  CB_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)ws, (int)nb, (int)d, colsum);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

extern "C" int cb_trunk_layer_bwd_f32(const float* g, const uint64_t* relu_bits, const float* row_scale, void* out, int out_bf16,
                                      float* gx0, int accumulate, int64_t rows, int64_t d, float drop_p, uint64_t seed,
                                      const uint64_t* seed_dev, int64_t row0, float c_act, float c_mix, const float* g2, uint64_t seed2, float c2,
                                      const int32_t* g2_pos, float* colsum, void* ws, size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_layer_bwd_f32: d must be a multiple of 256");
  if (rows == 0) return CB_OK;
  CB_CHECK_ARG(!g2 || aligned16(g2), CB_E_INVALID, "cb_trunk_layer_bwd_f32: misaligned second gradient");
  CB_CHECK_ARG(g && relu_bits && (out || colsum) && aligned16(g) && ((uintptr_t)out % (out_bf16 ? 8 : 16) == 0) && (!gx0 || aligned16(gx0)),
               CB_E_INVALID, "cb_trunk_layer_bwd_f32: null or misaligned pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_trunk_layer_bwd_f32: dropout p out of range");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(rows, d)), CB_E_WORKSPACE, "cb_trunk_layer_bwd_f32: workspace too small");
  return launch_trunk_bwd(0, out_bf16, g, relu_bits, nullptr, row_scale, out, gx0, accumulate, rows, d, drop_p, seed, seed_dev, row0, c_act, c_mix,
                          colsum, ws, ws_bytes, (hipStream_t)stream, nullptr, g2, seed2, c2, g2_pos);
}

// cb_trunk_layer_bwd_f32 for layer 0 of the 'Initial' trunk (all rows, fp32, no in-place accumulator) which also FOLDS the mix gradients (round 6):
//   out_m = c_mix * ( dropout_bwd_seed(g) + sum_q dropout_bwd_{mix_seeds[q]}(mix_g[q][mix_pos[q][r] | r]) ),  n_mix <= 2 operands (host arrays; mix_pos[q] NULL: a
// dense [rows, d] operand; else int32 [rows] positions in a compact one, < 0: absent) — what cb_gemm_tn_instage_f32 reads beside dL/d dropout(X0).  out and
// colsum exactly as cb_trunk_layer_bwd_f32 (bit-identical).  The elementwise form of cb_spmm_csr_store_bwd_mix_f32's epilogue, for the levels whose reverse
// aggregation does not carry the store backward (dense levels, mid-size graphs, row shards).
extern "C" int cb_trunk_layer_bwd_fold_f32(const float* g, const uint64_t* relu_bits, const float* row_scale, float* out, int64_t rows, int64_t d, float drop_p,
                                           uint64_t seed, const uint64_t* seed_dev, int64_t row0, float c_act, float c_mix, int32_t n_mix, const float* const* mix_g,
                                           const int32_t* const* mix_pos, const uint64_t* mix_seeds, float* out_m, float* colsum, void* ws, size_t ws_bytes,
                                           int32_t cs_src, const uint64_t* cs_bits, float cs_c, float* colsum2, void* ws2, size_t ws2_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: d must be a multiple of 256");
  if (rows == 0) return CB_OK;
  CB_CHECK_ARG(g && relu_bits && out && out_m && aligned16(g) && aligned16(out) && aligned16(out_m), CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: null or misaligned pointer");
  CB_CHECK_ARG(n_mix >= 0 && n_mix <= 2 && (n_mix == 0 || (mix_g && mix_seeds)), CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: 0..2 mix operands");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: dropout p out of range");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(rows, d)), CB_E_WORKSPACE, "cb_trunk_layer_bwd_fold_f32: workspace too small");
  CB_CHECK_ARG(!colsum2 || (cs_src >= 0 && cs_src < n_mix && cs_bits && (uintptr_t)cs_bits % 8 == 0 && ws2 && ws2_bytes >= cb_colsum_workspace_bytes(rows, d)),
               CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: the second column sum needs an operand index, its mask words and a workspace");
  FoldOps fo{};
  fo.n = n_mix; fo.out_m = out_m;
  fo.cs_src = cs_src; fo.cs_bits = (const unsigned long long*)cs_bits; fo.cs_c = cs_c; fo.cs_partial = colsum2 ? (float*)ws2 : nullptr;
  for (int q = 0; q < n_mix; ++q) {
    CB_CHECK_ARG(mix_g[q] && aligned16(mix_g[q]), CB_E_INVALID, "cb_trunk_layer_bwd_fold_f32: null or misaligned mix operand %d", q);
    fo.g[q] = mix_g[q]; fo.pos[q] = mix_pos ? mix_pos[q] : nullptr; fo.seed[q] = mix_seeds[q];
  }
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_trunk_bwd_fold, dim3((unsigned)nb), dim3(kBlock), kBlock * 4 * sizeof(float), st, g, (const unsigned long long*)relu_bits, row_scale, out, fo, rows,
                     (int)d, drop_p > 0.f ? dropout_threshold(drop_p) : 0u, 1.f / (1.f - drop_p), seed, seed_dev, row0, c_act, c_mix, colsum ? (float*)ws : nullptr);
  CB_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)ws, (int)nb, (int)d, colsum);
    CB_LAUNCH_CHECK();
  }
  if (colsum2) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)ws2, (int)nb, (int)d, colsum2);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

// cb_trunk_layer_bwd_f32 over a SUBSET of the rows: g and out are compact [n_rows, d] matrices holding rows row_index[0 .. n_rows) of the full
// ones (ascending global row ids); relu_bits / row_scale are the full arrays, the dropout mask is drawn at the global row.  colsum = the
// column sums over the subset (all other rows of a row-sparse backward are zero).
extern "C" int cb_trunk_layer_bwd_rows_f32(const float* g, const int64_t* row_index, int64_t n_rows, const uint64_t* relu_bits, const float* row_scale,
                                           float* out, int64_t d, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, float c_act,
                                           const float* g2, uint64_t seed2, float c2, const int32_t* g2_pos, float* colsum, void* ws, size_t ws_bytes,
                                           void* stream) {
  CB_CHECK_ARG(n_rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_layer_bwd_rows_f32: d must be a multiple of 256");
  if (n_rows == 0) {
    if (colsum) CB_HIP(hipMemsetAsync(colsum, 0, (size_t)d * sizeof(float), (hipStream_t)stream));
    return CB_OK;
  }
  CB_CHECK_ARG(g && row_index && relu_bits && out && aligned16(g) && aligned16(out), CB_E_INVALID, "cb_trunk_layer_bwd_rows_f32: null or misaligned pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_trunk_layer_bwd_rows_f32: dropout p out of range");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(n_rows, d)), CB_E_WORKSPACE, "cb_trunk_layer_bwd_rows_f32: workspace too small");
  CB_CHECK_ARG(!g2 || (aligned16(g2) && g2_pos), CB_E_INVALID, "cb_trunk_layer_bwd_rows_f32: the second gradient needs 16-byte aligned rows and its position map");
  return launch_trunk_bwd(0, 0, g, relu_bits, nullptr, row_scale, out, nullptr, 0, n_rows, d, drop_p, seed, seed_dev, row0, c_act, 0.f, colsum, ws, ws_bytes,
                          (hipStream_t)stream, row_index, g2, seed2, c2, g2_pos);
}

extern "C" int cb_trunk_input_bwd_f32(const float* g, const float* add, const float* act, float* out, int64_t rows, int64_t d,
                                      float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws,
                                      size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_input_bwd_f32: d must be a multiple of 256");
  if (rows == 0) return CB_OK;
  CB_CHECK_ARG(g && add && act && out && aligned16(g) && aligned16(add) && aligned16(act) && aligned16(out), CB_E_INVALID,
               "cb_trunk_input_bwd_f32: null or misaligned pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_trunk_input_bwd_f32: dropout p out of range");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(rows, d)), CB_E_WORKSPACE, "cb_trunk_input_bwd_f32: workspace too small");
  return launch_trunk_bwd(1, 0, g, nullptr, act, nullptr, out, const_cast<float*>(add), 1, rows, d, drop_p, seed, seed_dev, row0, 0.f, 0.f, colsum,
                          ws, ws_bytes, (hipStream_t)stream);
}

static int trunk_input_bwd_multi_impl(const float* g, uint64_t seed, int32_t n_mix, const float* const* g_mix, const uint64_t* seeds_mix,
                                      float c_mix, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                                      const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                                      const uint64_t* act_bits, const int32_t* const* g_mix_pos, void* stream, int32_t n_cs, const int32_t* cs_src,
                                      const uint64_t* const* cs_bits, const float* cs_c, float* const* colsum2, void* ws2, size_t ws2_bytes) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_input_bwd_multi_f32: d must be a multiple of 256");
  CB_CHECK_ARG(n_mix >= 0 && n_mix <= kMixMax && (n_mix == 0 || (g_mix && seeds_mix)), CB_E_INVALID,
               "cb_trunk_input_bwd_multi_f32: 0..%d mixed-in gradients", kMixMax);
  if (rows == 0) return CB_OK;
  CB_CHECK_ARG(g && (act || act_bits) && out && aligned16(g) && (!act || aligned16(act)) && aligned16(out) && ((uintptr_t)act_bits % 8 == 0),
               CB_E_INVALID, "cb_trunk_input_bwd_multi_f32: null or misaligned pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, CB_E_INVALID, "cb_trunk_input_bwd_multi_f32: dropout p out of range");
  CB_CHECK_ARG(!colsum || (ws && ws_bytes >= cb_colsum_workspace_bytes(rows, d)), CB_E_WORKSPACE, "cb_trunk_input_bwd_multi_f32: workspace too small");
  MixTable mt{};
  mt.n = n_mix;
  for (int i = 0; i < n_mix; ++i) mt.pos[i] = g_mix_pos ? g_mix_pos[i] : nullptr;
  CB_CHECK_ARG(n_cs >= 0 && n_cs <= 2 && (n_cs == 0 || (cs_src && cs_bits && cs_c && colsum2)), CB_E_INVALID, "cb_trunk_input_bwd_multi_cs_f32: 0..2 extra column sums");
  const size_t plane = cb_colsum_workspace_bytes(rows, d);
  CB_CHECK_ARG(n_cs == 0 || (ws2 && ws2_bytes >= (size_t)n_cs * plane), CB_E_WORKSPACE, "cb_trunk_input_bwd_multi_cs_f32: second workspace too small");
  for (int q = 0; q < n_cs; ++q) {
    CB_CHECK_ARG(cs_src[q] >= 0 && cs_src[q] < n_mix && cs_bits[q] && (uintptr_t)cs_bits[q] % 8 == 0 && colsum2[q], CB_E_INVALID,
                 "cb_trunk_input_bwd_multi_cs_f32: extra column sum %d needs an operand index, its mask words and a result vector", q);
    mt.cs_src[q] = cs_src[q]; mt.cs_bits[q] = (const unsigned long long*)cs_bits[q]; mt.cs_c[q] = cs_c[q]; mt.cs_partial[q] = (float*)((char*)ws2 + (size_t)q * plane);
  }
  for (int i = 0; i < n_mix; ++i) {
    CB_CHECK_ARG(g_mix[i] && aligned16(g_mix[i]), CB_E_INVALID, "cb_trunk_input_bwd_multi_f32: null or misaligned mixed-in gradient %d", i);
    mt.g[i] = g_mix[i];
    mt.seed[i] = seeds_mix[i];
  }
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  const uint32_t thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  hipStream_t st = (hipStream_t)stream;
#define CB_MIX_LAUNCH(N_)                                                                                                        \
  hipLaunchKernelGGL((k_trunk_input_bwd_multi<N_>), dim3((unsigned)nb), dim3(kBlock), kBlock * 4 * sizeof(float), st, g, mt, act, (const unsigned long long*)act_bits, out, rows, \
                     (int)d, thresh, 1.f / (1.f - drop_p), seed, seed_dev, row0, c_mix, colsum ? (float*)ws : nullptr)
  switch (n_mix) {
    case 0: CB_MIX_LAUNCH(0); break;
    case 1: CB_MIX_LAUNCH(1); break;
    case 2: CB_MIX_LAUNCH(2); break;
    case 3: CB_MIX_LAUNCH(3); break;
    case 4: CB_MIX_LAUNCH(4); break;
    case 5: CB_MIX_LAUNCH(5); break;
    case 6: CB_MIX_LAUNCH(6); break;
    default: CB_MIX_LAUNCH(7); break;
  }
#undef CB_MIX_LAUNCH
  CB_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)ws, (int)nb, (int)d, colsum);
    CB_LAUNCH_CHECK();
  }
  for (int q = 0; q < n_cs; ++q) {
    hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)mt.cs_partial[q], (int)nb, (int)d, colsum2[q]);
    CB_LAUNCH_CHECK();
  }
  return CB_OK;
}

extern "C" int cb_trunk_input_bwd_multi_f32(const float* g, uint64_t seed, int32_t n_mix, const float* const* g_mix, const uint64_t* seeds_mix,
                                            float c_mix, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                                            const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                                            const uint64_t* act_bits, const int32_t* const* g_mix_pos, void* stream) {
  return trunk_input_bwd_multi_impl(g, seed, n_mix, g_mix, seeds_mix, c_mix, act, out, rows, d, drop_p, seed_dev, row0, colsum, ws, ws_bytes, act_bits, g_mix_pos,
                                    stream, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}

// The same, which also returns n_cs (<= 2) extra column sums: colsum2[q] = the column sums of cs_c[q] * dropout_bwd_{seeds_mix[cs_src[q]]}(g_mix[cs_src[q]])
// through the mask words cs_bits[q] (indexed by the node row, also for a compact operand) — the bias gradients of the stores whose backward was applied by
// cb_spmm_csr_store_bwd_f32.  Dense operand: the partial-sum order of cb_trunk_layer_bwd_f32's column sums (bit-identical).  ws2: n_cs planes of
// cb_colsum_workspace_bytes(rows, d).
extern "C" int cb_trunk_input_bwd_multi_cs_f32(const float* g, uint64_t seed, int32_t n_mix, const float* const* g_mix, const uint64_t* seeds_mix,
                                               float c_mix, const float* act, float* out, int64_t rows, int64_t d, float drop_p,
                                               const uint64_t* seed_dev, int64_t row0, float* colsum, void* ws, size_t ws_bytes,
                                               const uint64_t* act_bits, const int32_t* const* g_mix_pos, int32_t n_cs, const int32_t* cs_src,
                                               const uint64_t* const* cs_bits, const float* cs_c, float* const* colsum2, void* ws2, size_t ws2_bytes,
                                               void* stream) {
  return trunk_input_bwd_multi_impl(g, seed, n_mix, g_mix, seeds_mix, c_mix, act, out, rows, d, drop_p, seed_dev, row0, colsum, ws, ws_bytes, act_bits, g_mix_pos,
                                    stream, n_cs, cs_src, cs_bits, cs_c, colsum2, ws2, ws2_bytes);
}

extern "C" int cb_gather_rows_f32(const float* src, int64_t ld, const int64_t* idx, int64_t n_idx, int64_t d, float* out,
                                  void* stream) {
  CB_CHECK_ARG(n_idx >= 0 && d >= 0 && d < (1 << 24) && ld >= d, CB_E_INVALID, "cb_gather_rows_f32: bad size");
  if (n_idx == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(src && idx && out, CB_E_INVALID, "cb_gather_rows_f32: null pointer");
  const int vec_ok = aligned16(src) && aligned16(out) && d % 4 == 0 && ld % 4 == 0;
  const int64_t work = vec_ok ? n_idx * (d / 4) : n_idx * d;
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(work)), dim3(kBlock), 0, (hipStream_t)stream, src, ld, idx, n_idx, (int)d, out, vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

namespace cb {
// The trunk's fused store (cb_spmm_core.h FusedEpi: ReLU, mask words, mix, dropout) on a SUBSET of the rows, after a dense transform instead of
// inside an aggregation — the rows-only forward of trunk.py (the last layer on the loss rows):
//   act = relu(y[r]);  out[r] = dropout_seed((c_act * act + c_mix * mix_src[mix_index[r]]));  mask words at the GLOBAL row row_index[r].
// y / out: compact [n_rows, d]; relu_bits: the full array; mix_src (may be null): its row mix_index[r] (mix_index null: row_index[r], i.e. the full
// array).  One wavefront per row, lane l = columns 4l .. 4l+3 of each tile.
__global__ void __launch_bounds__(kBlock) k_trunk_store_rows(const float* __restrict__ y, const int64_t* __restrict__ ridx, int64_t n_rows, int d,
                                                             const float* __restrict__ mix_src, int64_t ld_mix, const int64_t* __restrict__ midx, float c_act,
                                                             float c_mix, uint32_t thresh,
                                                             float keep_scale, uint64_t seed, const uint64_t* __restrict__ seed_dev, int64_t row0,
                                                             unsigned long long* __restrict__ bits, int relu_only, float* __restrict__ out,
                                                             float* __restrict__ out_act) {
  if (seed_dev) seed += *seed_dev;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tiles = d >> 8;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + w; r < n_rows; r += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t rr = ridx[r], mr = midx ? midx[r] : rr;
    for (int tile = 0; tile < tiles; ++tile) {
      const int c = tile * 256 + lane * 4;
      const float4 y4 = *reinterpret_cast<const float4*>(y + r * d + c);
      float a[4] = {fmaxf(y4.x, 0.f), fmaxf(y4.y, 0.f), fmaxf(y4.z, 0.f), fmaxf(y4.w, 0.f)}, m[4] = {1.f, 1.f, 1.f, 1.f};
      if (thresh) keep4(seed, ((row0 + rr) * d + c) >> 2, thresh, keep_scale, m);
      if (bits) {
        unsigned long long mine = 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long wq = __ballot(a[k] > 0.f && (relu_only || m[k] != 0.f));
          if (lane == k) mine = wq;
        }
        if (lane < 4) bits[(rr * tiles + tile) * 4 + lane] = mine;
      }
      if (out_act) *reinterpret_cast<float4*>(out_act + r * d + c) = make_float4(a[0], a[1], a[2], a[3]);
      float x[4] = {a[0], a[1], a[2], a[3]};
      if (mix_src) {
        const float4 q = *reinterpret_cast<const float4*>(mix_src + mr * ld_mix + c);
        x[0] = mix2(c_act, a[0], c_mix, q.x); x[1] = mix2(c_act, a[1], c_mix, q.y); x[2] = mix2(c_act, a[2], c_mix, q.z); x[3] = mix2(c_act, a[3], c_mix, q.w);
      }
      if (thresh) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] *= m[k];
      }
      *reinterpret_cast<float4*>(out + r * d + c) = make_float4(x[0], x[1], x[2], x[3]);
    }
  }
}
}  // namespace cb

extern "C" int cb_trunk_store_rows_f32(const float* y, const int64_t* row_index, int64_t n_rows, int64_t d, const float* mix_src, int64_t ld_mix,
                                       const int64_t* mix_index, float c_act, float c_mix, float drop_p, uint64_t seed, const uint64_t* seed_dev, int64_t row0, uint64_t* relu_bits,
                                       int bits_relu_only, float* out, float* out_act, void* stream) {
  CB_CHECK_ARG(n_rows >= 0 && d > 0 && d % 256 == 0 && d < (1 << 20), CB_E_INVALID, "cb_trunk_store_rows_f32: d must be a multiple of 256");
  if (n_rows == 0) return CB_OK;
  CB_CHECK_ARG(y && row_index && out && aligned16(y) && aligned16(out) && (!out_act || aligned16(out_act)) && (!mix_src || (aligned16(mix_src) && ld_mix % 4 == 0 && ld_mix >= d)) &&
                   (!relu_bits || (uintptr_t)relu_bits % 8 == 0),
               CB_E_INVALID, "cb_trunk_store_rows_f32: null or misaligned pointer");
  CB_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID, "cb_trunk_store_rows_f32: dropout p / row offset out of range");
  int64_t nb = (n_rows + kBlock / kWave - 1) / (kBlock / kWave);
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  hipLaunchKernelGGL(k_trunk_store_rows, dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, y, row_index, n_rows, (int)d, mix_src, ld_mix, mix_index, c_act, c_mix,
                     drop_p > 0.f ? dropout_threshold(drop_p) : 0u, 1.f / (1.f - drop_p), seed, seed_dev, row0, (unsigned long long*)relu_bits, bits_relu_only, out, out_act);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_expand_rows_f32(const float* src, const int32_t* pos, int64_t n_rows, int64_t d, float fill, float* out, void* stream) {
  CB_CHECK_ARG(n_rows >= 0 && d >= 0 && d < (1 << 24), CB_E_INVALID, "cb_expand_rows_f32: bad size");
  if (n_rows == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(pos && out, CB_E_INVALID, "cb_expand_rows_f32: null pointer");
  CB_CHECK_ARG(d % 4 == 0 && aligned16(out) && (!src || aligned16(src)), CB_E_INVALID, "cb_expand_rows_f32: 16-byte aligned rows with d %% 4 == 0 expected");
  hipLaunchKernelGGL(k_expand_rows, dim3(grid_for(n_rows * (d / 4))), dim3(kBlock), 0, (hipStream_t)stream, src, pos, n_rows, (int)d, fill, out);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_gather_rows_bf16_f32(const float* src, int64_t ld, const int64_t* idx, int64_t n_idx, int64_t d, uint16_t* out,
                                       void* stream) {
  CB_CHECK_ARG(n_idx >= 0 && d >= 0 && d < (1 << 24) && ld >= d, CB_E_INVALID, "cb_gather_rows_bf16_f32: bad size");
  if (n_idx == 0 || d == 0) return CB_OK;
  CB_CHECK_ARG(src && idx && out, CB_E_INVALID, "cb_gather_rows_bf16_f32: null pointer");
  const int vec_ok = aligned16(src) && ((uintptr_t)out % 8 == 0) && d % 4 == 0 && ld % 4 == 0;
  const int64_t work = vec_ok ? n_idx * (d / 4) : n_idx * d;
  hipLaunchKernelGGL(k_gather_rows_bf16, dim3(grid_for(work)), dim3(kBlock), 0, (hipStream_t)stream, src, ld, idx, n_idx, (int)d, out, vec_ok);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

// =============================================================================================
// Normalisation tricks (GNN_model/norm_tricks.py) as fused reductions.
// =============================================================================================
namespace cb {

// ---- row-wise: node_norm (norm_tricks.py:53-84) ---------------------------------------------
// y = (x - c*mu) * std^-q with mu, std = sqrt(var_biased + eps) over the features of one row:
//   'n': c=1,q=1   'v': c=0,q=1   'm': c=1,q=0   'srv'/'pr'(power_root 2): c=0,q=1/2
// One wavefront per row; the row stays in registers for d <= 64*NR (d <= 1024), otherwise it is re-read.
// stats[r] = {mu, std} is kept for the backward:
//   dx = s*(g - c*mean(g)) - q * std^(-q-2) / d * (x - mu) * sum_k g_k (x_k - c*mu)
__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ void __launch_bounds__(kBlock) k_node_norm_fwd(const float* __restrict__ x, float* __restrict__ y, float2* __restrict__ stats,
                                                          int64_t rows, int d, float c, float q, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nw) {
    const float* xr = x + r * d;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) s += xr[j];
    const float mu = wave_sum(s) / d;
    float v = 0.f;
    for (int j = lane; j < d; j += 64) { const float t = xr[j] - mu; v += t * t; }
    const float sd = sqrtf(wave_sum(v) / d + eps);
    const float sc = (q == 0.f) ? 1.f : (q == 1.f ? 1.f / sd : 1.f / sqrtf(sd));
    float* yr = y + r * d;
    for (int j = lane; j < d; j += 64) yr[j] = (xr[j] - c * mu) * sc;
    if (lane == 0 && stats) stats[r] = make_float2(mu, sd);
  }
}

__global__ void __launch_bounds__(kBlock) k_node_norm_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float2* __restrict__ stats, float* __restrict__ dx, int64_t rows,
                                                          int d, float c, float q) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nw) {
    const float* xr = x + r * d;
    const float* gr = g + r * d;
    const float2 st = stats[r];
    const float mu = st.x, sd = st.y;
    float sg = 0.f, sgx = 0.f;
    for (int j = lane; j < d; j += 64) { const float gv = gr[j]; sg += gv; sgx += gv * (xr[j] - c * mu); }
    sg = wave_sum(sg);
    sgx = wave_sum(sgx);
    const float sc = (q == 0.f) ? 1.f : (q == 1.f ? 1.f / sd : 1.f / sqrtf(sd));
    const float k2 = (q == 0.f) ? 0.f : q * sc / (sd * sd) / d * sgx;   // q * std^(-q-2) / d * sum g (x - c mu)
    const float gbar = c * sg / d;
    float* dr = dx + r * d;
    for (int j = lane; j < d; j += 64) dr[j] = sc * (gr[j] - gbar) - k2 * (xr[j] - mu);
  }
}

// ---- column statistics: colsum(x) and colsum(x^2) in one pass (two-stage, fixed order) -------------
__global__ void __launch_bounds__(kBlock) k_colstats(const float* __restrict__ x, const float* __restrict__ w, int64_t rows, int d,
                                                     float* __restrict__ p_sum, float* __restrict__ p_sq) {
  // thread owns one column per pass; block owns a row slab.  w (optional) multiplies x element-wise (for sum(g*xhat)).
  const int64_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float s = 0.f, s2 = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float v = x[r * d + c];
      const float u = w ? v * w[r * d + c] : v * v;
      s += v;
      s2 += u;
    }
    p_sum[(int64_t)blockIdx.x * d + c] = s;
    p_sq[(int64_t)blockIdx.x * d + c] = s2;
  }
}

// y[r,c] = (x[r,c] - shift[c]) * scale[c] + bias[c]   (mean_norm / pair_norm / BatchNorm1d apply; any of the vectors may be null)
__global__ void __launch_bounds__(kBlock) k_col_affine(const float* __restrict__ x, const float* __restrict__ shift,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       float gscale, float* __restrict__ y, int64_t n, int d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    float v = x[i];
    if (shift) v -= shift[c];
    if (scale) v *= scale[c];
    v *= gscale;
    if (bias) v += bias[c];
    y[i] = v;
  }
}

// dx[r,c] = a[c] * g[r,c] + b[c] * xh[r,c] + e[c]   (backward combine of the column norms; xh may be null)
__global__ void __launch_bounds__(kBlock) k_col_bwd_combine(const float* __restrict__ g, const float* __restrict__ xh,
                                                            const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ e, float ga, float gb, float* __restrict__ dx,
                                                            int64_t n, int d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    float v = (a ? a[c] : 1.f) * ga * g[i];
    if (xh) v += (b ? b[c] : 1.f) * gb * xh[i];
    if (e) v += e[c];
    dx[i] = v;
  }
}

}  // namespace cb

extern "C" int cb_node_norm_fwd_f32(const float* x, float* y, float* stats2, int64_t rows, int64_t d, float c, float q, float eps,
                                    void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d < (1 << 24) && (rows == 0 || (x && y)), CB_E_INVALID, "cb_node_norm_fwd_f32: bad argument");
  if (rows == 0) return CB_OK;
  hipLaunchKernelGGL(k_node_norm_fwd, dim3(grid_for(rows * 64)), dim3(kBlock), 0, (hipStream_t)stream, x, y, (float2*)stats2, rows,
                     (int)d, c, q, eps);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_node_norm_bwd_f32(const float* x, const float* g, const float* stats2, float* dx, int64_t rows, int64_t d, float c,
                                    float q, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d < (1 << 24) && (rows == 0 || (x && g && stats2 && dx)), CB_E_INVALID,
               "cb_node_norm_bwd_f32: bad argument");
  if (rows == 0) return CB_OK;
  hipLaunchKernelGGL(k_node_norm_bwd, dim3(grid_for(rows * 64)), dim3(kBlock), 0, (hipStream_t)stream, x, g, (const float2*)stats2, dx,
                     rows, (int)d, c, q);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" size_t cb_colstats_workspace_bytes(int64_t rows, int64_t d) { return 2 * cb_colsum_workspace_bytes(rows, d); }

extern "C" int cb_colstats_f32(const float* x, const float* w, int64_t rows, int64_t d, float* colsum, float* colsum2, void* ws,
                               size_t ws_bytes, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && d < (1 << 20) && colsum && colsum2 && (rows == 0 || x), CB_E_INVALID, "cb_colstats_f32: bad argument");
  CB_CHECK_ARG(ws && ws_bytes >= cb_colstats_workspace_bytes(rows > 0 ? rows : 1, d), CB_E_WORKSPACE, "cb_colstats_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (rows + 63) / 64;
  if (nb > kMaxBlocks) nb = kMaxBlocks;
  if (nb < 1) nb = 1;
  float* p1 = (float*)ws;
  float* p2 = p1 + (size_t)nb * d;
  hipLaunchKernelGGL(k_colstats, dim3((unsigned)nb), dim3(kBlock), 0, st, x, w, rows, (int)d, p1, p2);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)p1, (int)nb, (int)d, colsum);
  CB_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)d), dim3(kBlock), 0, st, (const float*)p2, (int)nb, (int)d, colsum2);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_col_affine_f32(const float* x, const float* shift, const float* scale, const float* bias, float gscale, float* y,
                                 int64_t rows, int64_t d, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && (rows == 0 || (x && y)), CB_E_INVALID, "cb_col_affine_f32: bad argument");
  if (rows == 0) return CB_OK;
  hipLaunchKernelGGL(k_col_affine, dim3(grid_for(rows * d)), dim3(kBlock), 0, (hipStream_t)stream, x, shift, scale, bias, gscale, y,
                     rows * d, (int)d);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_col_bwd_combine_f32(const float* g, const float* xh, const float* a, const float* b, const float* e, float ga,
                                      float gb, float* dx, int64_t rows, int64_t d, void* stream) {
  CB_CHECK_ARG(rows >= 0 && d > 0 && (rows == 0 || (g && dx)), CB_E_INVALID, "cb_col_bwd_combine_f32: bad argument");
  if (rows == 0) return CB_OK;
  hipLaunchKernelGGL(k_col_bwd_combine, dim3(grid_for(rows * d)), dim3(kBlock), 0, (hipStream_t)stream, g, xh, a, b, e, ga, gb, dx,
                     rows * d, (int)d);
  CB_LAUNCH_CHECK();
  return CB_OK;
}
