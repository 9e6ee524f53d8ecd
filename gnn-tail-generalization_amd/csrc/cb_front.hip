// The forward front of the residual trunk in ONE kernel:
//     x   = F.dropout(x)                          GNN_model/GCN.py:104
//     X0  = relu(Linear_0(x))                     GCN.py:105-107        -> stored (the 'Initial' mixes and the backward need it) + mask words of X0 > 0
//     X0d = F.dropout(X0)                         GCN.py:110            -> stays on chip (optionally also stored: out_drop)
//     Z0  = a . (X0d W_0) + E_0                   GCN.py:213,225,230-235 (first GCNConv's transform)   -> stored
// Replaces cb_gemm_nn_indrop_drop2_f32 followed by cb_gemm_nn_f32: the two kernels furthest below their roofline in round 3 (the input Linear
// wrote X0 AND its dropped copy at 0.41 of HBM, the layer-0 GEMM re-read the copy at the power-limited MFMA clock).  Here a block keeps its
// 64 rows of X0d in LDS and multiplies them by W_0 before anything else is read: 10 GB of writes and 10 GB of reads per step disappear.
//
// One persistent block of 4 wavefronts per slot (two slots per CU: 75 KB of LDS each), tiles of 64 rows, per tile
//   P0  the tile of x -> dropout -> fp32 LDS tile [64][K1 + 4]
//   P1  tile x W_in^T on the matrix cores (tile_times_image_coop: three-limb bf16 products, A limbs split once per K step into LDS planes, B fragments
//       from the L2-resident image)
//   P2  accumulators -> the SAME LDS region as a [64][260] tile (the x tile is dead)
//   P3  row pass: a wavefront owns 16 rows, a lane 4 columns: + bias, ReLU -> X0 row store (1 KiB, streaming), four ballots = the row's mask
//       words, Philox keep-mask of the dropout in front of layer 0 -> X0d back into the tile (-> out_drop row store if requested)
//   P4  tile x W_0 on the matrix cores -> rowscale . acc + addend -> Z0 through wave-private strips
// Two blocks share a CU.  Measured (profiles/r04_front_kernel.md): 14.4 ms at the headline shape against 15.3 ms for the two kernels; the
// co-resident blocks fall into lock step (the phases' times add up: 8.9 ms of K loops at 52 % of the bf16 MFMA peak + 5.4 ms of loads, row
// pass and stores), so what the kernel buys is the 20 GB of traffic, not yet an overlap of its own phases.
// Bit-identical to the two-kernel form: same operand values (dropout products rounded to fp32 before the limb split), same limb products
// in the same order, same epilogue expressions (tests/test_gpu_kernels.py::test_forward_front_*).
#include "cb_common.h"
#include "cb_limb_core.h"
#include "cb_philox.h"
#include "cb_tile_gemm.h"

namespace cb {

struct FrontArgs {
  const float* x;            // [M, ld_x]
  int64_t ld_x;
  const uint4* image_in;     // W_in^T as a [K1, 256] image
  const float* bias_in;      // [256] or null
  const uint4* image_0;      // W_0 as a [256, 256] image
  const float* rowscale;     // [M] or null  (a = D_out^-1/2)
  const float* addend;       // [M, ld_add] or null (E_0)
  int64_t ld_add;
  float* x0;                 // [M, ld_x0]
  int64_t ld_x0;
  unsigned long long* bits;  // [M][4] or null
  float* x0_drop;            // [M, ld_drop] or null
  int64_t ld_drop;
  float* z0;                 // [M, ld_z]
  int64_t ld_z;
  int64_t M;
  uint32_t thresh;           // dropout threshold (0: no dropout anywhere)
  float keep_scale;
  uint64_t seed_x, seed_x0;
  const uint64_t* seed_dev;
  int64_t row0;
  int n_tiles;
};

template <int NS1>      // K1 = 16 * NS1 input features
__global__ void __launch_bounds__(256, 2) k_front(FrontArgs fa) {
  constexpr int K1 = 16 * NS1, LDX = K1 + 4, NV = kTM * K1 / 4 / 256;      // float4 of x per thread per tile
  static_assert(NV >= 1 && kTM * LDX <= kTM * kTLD, "x tile must fit the region of the X0 tile");
  __shared__ __attribute__((aligned(16))) float tile[kTM * kTLD];
  // Round 6: the K loops split the A tile's limbs ONCE per element (tile_times_image_coop: the block's 256 threads split each K step's 64 x 16 slab into
  // bf16 planes, every wavefront reads its fragments from them) instead of once per multiplying wavefront: 14.3 -> 13.7 ms at the headline shape, bit-identical
  // (the matrix-core kernels run at the board's power limit: less VALU / LDS work per MFMA is a higher clock).  The planes (two stages of RowOperand<64>) and
  // the wave-private strips of the Z0 epilogue share one region: the strips are used after the last K step's barrier only, the planes before it.
  constexpr int kScratch = 2 * RowOperand<kTM>::BYTES > 4 * 8 * kCLD * 4 ? 2 * RowOperand<kTM>::BYTES : 4 * 8 * kCLD * 4;
  __shared__ __attribute__((aligned(16))) char scratch[kScratch];
  float (*cstrip)[8 * kCLD] = reinterpret_cast<float (*)[8 * kCLD]>(scratch);
  __shared__ float rs_tile[2][kTM];   // (two buffers by tile parity: a wavefront may start the next tile's P0 while others still run this tile's epilogue)
                                      // row scales of the tile's rows: a global load per (pass, row) in the Z0 epilogue would expose its latency 16 times
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint64_t sx = fa.seed_dev ? fa.seed_x + *fa.seed_dev : fa.seed_x, s0 = fa.seed_dev ? fa.seed_x0 + *fa.seed_dev : fa.seed_x0;
  float4 xr[NV];
  auto load_x = [&](int tile_id) {
    const int64_t m0 = (int64_t)tile_id * kTM;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + 256 * j, row = idx / (K1 / 4), c4 = idx % (K1 / 4);
      const int64_t m = m0 + row;
      xr[j] = (tile_id < fa.n_tiles && m < fa.M) ? *reinterpret_cast<const float4*>(fa.x + m * fa.ld_x + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float bvec[4] = {0.f, 0.f, 0.f, 0.f};
  if (fa.bias_in) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bvec[i] = fa.bias_in[4 * lane + i];
  }
  int par = 0;
  for (int tile_id = blockIdx.x; tile_id < fa.n_tiles; tile_id += gridDim.x, par ^= 1) {
    const int64_t m0 = (int64_t)tile_id * kTM;
    load_x(tile_id);      // (not prefetched across the GEMMs: those registers hold a third ring buffer of B fragments instead; the block that
                          //  shares the CU multiplies while this one waits)
    // ---- P0: dropout of x (the product cb_dropout_f32 forms, rounded to fp32 as a value of its own) -> LDS
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + 256 * j, row = idx / (K1 / 4), c4 = idx % (K1 / 4);
      float v[4] = {xr[j].x, xr[j].y, xr[j].z, xr[j].w};
      if (fa.thresh) {
        float mk[4];
        keep4(sx, ((fa.row0 + m0 + row) * K1 + 4 * c4) >> 2, fa.thresh, fa.keep_scale, mk);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = mul_rounded(v[i], mk[i]);
      }
      *reinterpret_cast<float4*>(tile + row * LDX + 4 * c4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (t < kTM) rs_tile[par][t] = (fa.rowscale && m0 + t < fa.M) ? fa.rowscale[m0 + t] : 1.f;      // (read in P4's epilogue: four barriers later)
    __syncthreads();
    // ---- P1: X0 pre-activation = tile(x) @ W_in^T
    f32x16 acc[2][2];
    tile_times_image_coop<NS1, LDX, 1>(tile, scratch, fa.image_in, w, lane, t, acc);      // (ends with a block barrier: the x tile is dead)
    // ---- P2: accumulators -> [64][260] tile (C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    {
      const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            tile[(32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * kTLD + 64 * w + 32 * j + l31] = acc[i][j][reg];
    }
    __syncthreads();
    // ---- P3: row pass (wavefront w: rows 16 w .. 16 w + 15; lane: columns 4 lane .. 4 lane + 3)
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      const int row = 16 * w + r;
      const int64_t m = m0 + row;
      float* tp = tile + row * kTLD + 4 * lane;
      const float4 v = *reinterpret_cast<const float4*>(tp);
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i] + bvec[i], 0.f);      // nn_epilogue: o * 1 + 0 + bias, then ReLU
      float xd[4] = {o[0], o[1], o[2], o[3]};
      if (fa.thresh) {
        float mk[4];
        keep4(s0, ((fa.row0 + m) * kND + 4 * lane) >> 2, fa.thresh, fa.keep_scale, mk);
#pragma unroll
        for (int i = 0; i < 4; ++i) xd[i] = o[i] * mk[i];
      }
      *reinterpret_cast<float4*>(tp) = make_float4(xd[0], xd[1], xd[2], xd[3]);
      if (m < fa.M) {      // (wave-uniform)
        store_stream4(fa.x0 + m * fa.ld_x0 + 4 * lane, o);
        if (fa.bits) {
          unsigned long long mine = 0ull;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned long long wq = __ballot(o[k] > 0.f);
            if (lane == k) mine = wq;
          }
          if (lane < 4) fa.bits[m * 4 + lane] = mine;
        }
        if (fa.x0_drop) store_stream4(fa.x0_drop + m * fa.ld_drop + 4 * lane, xd);
      }
    }
    __syncthreads();
    // ---- P4: Z0 = rowscale . (tile(X0d) @ W_0) + addend
    tile_times_image_coop<kNS, kTLD, 1>(tile, scratch, fa.image_0, w, lane, t, acc);      // (ends with a block barrier: tile and planes are dead; the strips are wave-private)
    acc_rows_through_strip(acc, cstrip[w], w, lane, [&](int row, int n, const float4& v) {
      const int64_t m = m0 + row;
      if (m < fa.M) {
        const float rs = rs_tile[par][row];
        float ad[4] = {0.f, 0.f, 0.f, 0.f};
        if (fa.addend) {
          const float4 a4 = *reinterpret_cast<const float4*>(fa.addend + m * fa.ld_add + n);
          ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
        }
        const float zero_bias = 0.f;      // (the epilogue expression of cb_gemm_core.h's nn_epilogue: o * rs + addend + bias)
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] * rs + ad[e] + zero_bias;
        store_stream4(fa.z0 + m * fa.ld_z + n, o);
      }
    });
  }
}

static inline bool fr_al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace cb

using namespace cb;

// bytes of the fragment image of a [K, 256] weight (cb_front_image_f32); 0 = this K has no forward-front kernel (K in {64, 128})
extern "C" size_t cb_front_image_bytes(int64_t K) {
  if (K != 64 && K != 128) return 0;      // (K = 256: the prefetched x tile no longer fits the registers next to the accumulators)
  return (size_t)(K / 16) * kNT * 3 * 64 * sizeof(uint4);
}

// image of B = W^T for an nn.Linear weight W [256, K] (transpose = 1) or of B = W [K, 256] (transpose = 0)
extern "C" int cb_front_image_f32(const float* W, int64_t ld, int64_t K, int transpose, void* image, size_t image_bytes, void* stream) {
  CB_CHECK_ARG(cb_front_image_bytes(K) > 0, CB_E_INVALID, "cb_front_image_f32: K must be 64 or 128 (got %lld)", (long long)K);
  CB_CHECK_ARG(W && image && ld >= (transpose ? K : kND), CB_E_INVALID, "cb_front_image_f32: null pointer / bad leading dimension");
  CB_CHECK_ARG(image_bytes >= cb_front_image_bytes(K) && fr_al16(image), CB_E_WORKSPACE, "cb_front_image_f32: image buffer too small or misaligned");
  const int64_t sk = transpose ? 1 : ld, sn = transpose ? ld : 1;
  const int ns = (int)(K / 16);
  hipLaunchKernelGGL(k_weight_image, dim3(ns * kNT * 64 / 256), dim3(256), 0, (hipStream_t)stream, W, sk, sn, (uint4*)image, ns);
  CB_LAUNCH_CHECK();
  return CB_OK;
}

extern "C" int cb_trunk_front_f32(const float* x, int64_t ld_x, int64_t M, int64_t K, const void* image_in, const float* bias_in, const void* image_0,
                                  const float* rowscale, const float* addend, int64_t ld_add, float* x0, int64_t ld_x0, uint64_t* relu_bits,
                                  float* x0_drop, int64_t ld_drop, float* z0, int64_t ld_z, float drop_p, uint64_t seed_x, uint64_t seed_x0,
                                  const uint64_t* seed_dev, int64_t row0, void* stream) {
  CB_CHECK_ARG(M >= 0 && cb_front_image_bytes(K) > 0 && drop_p >= 0.f && drop_p < 1.f && row0 >= 0, CB_E_INVALID,
               "cb_trunk_front_f32: bad size (K must be 64 or 128; hidden width 256) or p");
  CB_CHECK_ARG(M < ((int64_t)1 << 31) * kTM / 2, CB_E_RANGE, "cb_trunk_front_f32: size out of range");
  if (M == 0) return CB_OK;
  CB_CHECK_ARG(x && image_in && image_0 && x0 && z0, CB_E_INVALID, "cb_trunk_front_f32: null pointer");
  CB_CHECK_ARG(fr_al16(x) && ld_x % 4 == 0 && ld_x >= K && fr_al16(image_in) && fr_al16(image_0) && fr_al16(x0) && ld_x0 % 4 == 0 && ld_x0 >= kND &&
                   fr_al16(z0) && ld_z % 4 == 0 && ld_z >= kND && (!addend || (fr_al16(addend) && ld_add % 4 == 0 && ld_add >= kND)) &&
                   (!x0_drop || (fr_al16(x0_drop) && ld_drop % 4 == 0 && ld_drop >= kND)) && (!relu_bits || (uintptr_t)relu_bits % 8 == 0),
               CB_E_INVALID, "cb_trunk_front_f32: 16-byte aligned rows required");
  FrontArgs fa{x, ld_x, (const uint4*)image_in, bias_in, (const uint4*)image_0, rowscale, addend, ld_add, x0, ld_x0, (unsigned long long*)relu_bits,
               x0_drop, ld_drop, z0, ld_z, M};
  fa.thresh = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
  fa.keep_scale = 1.f / (1.f - drop_p);
  fa.seed_x = seed_x; fa.seed_x0 = seed_x0; fa.seed_dev = seed_dev; fa.row0 = row0;
  fa.n_tiles = (int)((M + kTM - 1) / kTM);
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  const int blocks = fa.n_tiles < 2 * n_cu ? fa.n_tiles : 2 * n_cu;      // two persistent blocks per CU
  const dim3 grid((unsigned)blocks), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (K == 64) hipLaunchKernelGGL((k_front<4>), grid, blk, 0, st, fa);
  else hipLaunchKernelGGL((k_front<8>), grid, blk, 0, st, fa);
  CB_LAUNCH_CHECK();
  return CB_OK;
}
