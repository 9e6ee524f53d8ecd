// Building blocks of the three-limb bf16-MFMA contractions: exact fp32 -> 3 x bf16 split, LDS operand images with their
// fragment reads, the per-wavefront software pipeline.  Design notes: cb_gemm_limb.hip.  Used by cb_gemm_limb.hip (NN / TN
// GEMMs) and cb_topk.hip (teacher-embedding scores + running top-K).
#pragma once
#include <utility>

#include "cb_common.h"
#include "cb_gemm_core.h"
#include "cb_philox.h"

namespace cb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// a == hi + mid + lo exactly, each limb a bf16 obtained by round-to-nearest-even (v_cvt_pk_bf16_f32) of what is left:
// |a - hi| <= 2^-9 |a|, |a - hi - mid| <= 2^-17 |a|, and the last residual has at most 8 significant bits, so its conversion
// is exact.  Works on pairs because the hardware converts and packs two values per instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float x0, float x1) {   // {bf16(x0), bf16(x1)} in one dword
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3x2(float a0, float a1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = cvt_pk_bf16(a0, a1);
  const float r0 = a0 - __uint_as_float(hi << 16), r1 = a1 - __uint_as_float(hi & 0xffff0000u);       // exact
  mid = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);     // exact
  lo = cvt_pk_bf16(s0, s1);                                                                            // exact
}

__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&pl)[3]) {
  uint32_t h[2], m[2], l[2];
  split3x2(v[0], v[1], h[0], m[0], l[0]);
  split3x2(v[2], v[3], h[1], m[1], l[1]);
  pl[0] = make_uint2(h[0], h[1]);
  pl[1] = make_uint2(m[0], m[1]);
  pl[2] = make_uint2(l[0], l[1]);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int KS = 16;   // K step

// x * m rounded to fp32 as a value of its own: what cb_dropout_f32 stores.  Left to the compiler, the product is contracted into the
// first residual of the limb split (fma(x, m, -hi)), i.e. the limbs then describe the UNROUNDED product — more exact, but no longer
// the bits of dropout-then-GEMM (HIP's __fmul_rn is a plain multiply and contracts just the same).
__device__ __forceinline__ float mul_rounded(float x, float m) {
  float r = x * m;
  asm volatile("" : "+v"(r));
  return r;
}

// ---- "row" operand: global [rows][k] (k contiguous) -> LDS planes [R rows][16 k] -------------------
// The K step's data moves in NV pieces (one float4 per lane each): load() only brings raw data into registers, stage()
// masks out-of-range elements, splits and writes the three planes.
template <int R, bool DROP = false>
struct RowOperand {
  static constexpr int PLANE = R * 32, BYTES = 3 * PLANE, NV = R / 64;   // float4 per thread per K step
  static_assert(NV >= 1, "tile rows");
  uint32_t voff[NV];  // element offset of this lane's j-th float4 from (tile row 0, k0): row t/4 + 64 j, k quad t%4;
                      // 0 when that row is outside the matrix, so a masked lane never addresses beyond the operand
  uint32_t woff;      // LDS byte offset of its 8-byte store in plane 0; j-th store: + j * 64 * 32
  uint32_t rmask;     // bit j: tile row t/4 + 64 j lies inside the matrix
  static constexpr bool HAS_SC = false;
  float sc[1];        // (interface shared with ColOperand: no per-k scale here)
  // DROP: F.dropout of this operand while it is staged (DropSpec)
  uint64_t dseed;
  uint32_t dthresh;
  float dscale;
  int64_t dK, dbase[NV];      // flat index of this lane's j-th float4 at K step 0
  __device__ __forceinline__ void set_drop(const DropSpec& d, int64_t m0, int64_t Ktot, int t) {
    dseed = d.seed_dev ? d.seed + *d.seed_dev : d.seed;
    dthresh = d.thresh;
    dscale = d.scale;
    dK = Ktot;
#pragma unroll
    for (int j = 0; j < NV; ++j) dbase[j] = (d.row0 + m0 + (t >> 2) + 64 * j) * d.width + (t & 3) * 4;
  }
  __device__ __forceinline__ void init(int64_t ld, int64_t rows_left, int t) {
    const int row = t >> 2, kq = t & 3;
    woff = row * 32 + (((kq >> 1) ^ ((row >> 3) & 1)) << 4) + ((kq & 1) << 3);
    rmask = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool in = row + 64 * j < rows_left;
      rmask |= in ? (1u << j) : 0u;
      voff[j] = in ? (uint32_t)((row + 64 * j) * ld + kq * 4) : 0u;
    }
  }
  // base = &A[tile row 0][k0] (uniform); k_left = K - k0 > 0; K % 4 == 0
  template <int J>
  __device__ __forceinline__ void load(float4 (&f)[NV], const float* __restrict__ base, int64_t /*ld*/, int64_t k_left, const float*,
                                       int t) const {
    f[J] = *reinterpret_cast<const float4*>(base + ((t & 3) * 4 < k_left ? voff[J] : 0u));
  }
  template <int J>
  __device__ __forceinline__ void stage(const float4 (&f)[NV], char* __restrict__ S, int64_t k_left, int t) const {
    const bool live = ((rmask >> J) & 1) && (t & 3) * 4 < k_left;
    float v[4] = {live ? f[J].x : 0.f, live ? f[J].y : 0.f, live ? f[J].z : 0.f, live ? f[J].w : 0.f};
    if constexpr (DROP) {      // the same product cb_dropout_f32 forms: x * (keep ? 1 / (1 - p) : 0)
      float mk[4];
      keep4(dseed, (dbase[J] + (dK - k_left)) >> 2, dthresh, dscale, mk);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = mul_rounded(v[i], mk[i]);
    }
    uint2 pl[3];
    split4(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + woff + J * (64 * 32)) = pl[p];
  }
  // fragment address (plane 0) of tile rows r0 + (lane & 31), r0 % 32 == 0
  static __device__ __forceinline__ uint32_t frag_addr(int r0, int lane) {
    const int row = r0 + (lane & 31);
    return row * 32 + (((lane >> 5) ^ ((row >> 3) & 1)) << 4);
  }
  static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, uint32_t addr, int plane) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(S + plane * PLANE + addr));
  }
};

// ---- "col" operand: global [k][cols] (cols contiguous) -> LDS planes [16 k][C cols] ------------------
template <int C, bool SCALED, bool DROP = false, int NT = 256>      // NT: threads of the block that stage the operand
struct ColOperand {
  static constexpr int ROWB = C * 2, PLANE = 16 * ROWB, BYTES = 3 * PLANE;
  static constexpr int TPR = C / 4, KPP = NT / TPR, NV = 16 / KPP;   // threads per k row, k rows per pass, float4 per thread
  static_assert(KPP >= 4 && KPP % 4 == 0 && NV >= 1, "staging map: whole swizzle groups of k rows per pass");
  static constexpr int NC = C / 32;                                    // 64-byte chunks per row
  static_assert(NC == 2 || NC == 4 || NC == 8, "tile widths 64 / 128 / 256");
  static __device__ __forceinline__ int swz(int k) { return NC == 2 ? ((k >> 1) & 1) : (k & 3); }
  uint32_t voff[NV];  // element offset of this lane's j-th float4 from (step row 0, tile col 0): k = t / TPR + KPP j, quad t % TPR
  uint32_t woff;      // LDS byte offset of its 8-byte store in plane 0; j-th store: + j * KPP * ROWB  (KPP % 4 == 0)
  bool cok;           // its 4 columns lie inside the matrix (N % 4 == 0)
  static constexpr bool HAS_SC = SCALED;
  float sc[SCALED ? NV : 1];   // raw per-row scales of the loaded step
  // DROP: F.dropout of this operand while it is staged (DropSpec); rows of the operand = the reduction axis
  uint64_t dseed;
  uint32_t dthresh;
  float dscale;
  int64_t drow, dtot, dwidth;
  int dcol;
  __device__ __forceinline__ void set_drop(const DropSpec& d, int64_t r_begin, int64_t total, int j0, int t) {
    dseed = d.seed_dev ? d.seed + *d.seed_dev : d.seed;
    dthresh = d.thresh;
    dscale = d.scale;
    drow = d.row0 + r_begin + t / TPR;
    dtot = total;
    dwidth = d.width;
    dcol = j0 + (t % TPR) * 4;
  }
  __device__ __forceinline__ void init(int64_t ld, int cols_left, int t) {
    const int k = t / TPR, nq = t % TPR;
    cok = nq * 4 < cols_left;
#pragma unroll
    for (int j = 0; j < NV; ++j) voff[j] = (uint32_t)((k + KPP * j) * ld + (cok ? nq * 4 : 0));
    woff = k * ROWB + ((((nq >> 3) ^ swz(k)) & (NC - 1)) << 6) + ((nq & 7) << 3);
  }
  // base = &B[step row 0][tile col 0] (uniform); k_left = operand rows from there > 0; kscale = their scales.
  // Lanes whose k row is past the end address row 0 of the step instead.
  template <int J>
  __device__ __forceinline__ void load(float4 (&f)[NV], const float* __restrict__ base, int64_t /*ld*/, int64_t k_left,
                                       const float* __restrict__ kscale, int t) {
    const int k = t / TPR + KPP * J;
    const bool kin = k < k_left;
    f[J] = *reinterpret_cast<const float4*>(base + (kin ? voff[J] : 0u));
    if constexpr (SCALED) sc[J] = kscale[kin ? k : 0];
  }
  template <int J>
  __device__ __forceinline__ void stage(const float4 (&f)[NV], char* __restrict__ S, int64_t k_left, int t) const {
    const bool live = cok && t / TPR + KPP * J < k_left;
    const float m = SCALED ? sc[SCALED ? J : 0] : 1.f;
    float v[4] = {live ? f[J].x * m : 0.f, live ? f[J].y * m : 0.f, live ? f[J].z * m : 0.f, live ? f[J].w * m : 0.f};
    if constexpr (DROP) {
      static_assert(!SCALED, "dropout of a row-scaled operand is not needed anywhere");
      float mk[4];
      keep4(dseed, ((drow + KPP * J + (dtot - k_left)) * dwidth + dcol) >> 2, dthresh, dscale, mk);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = mul_rounded(v[i], mk[i]);
    }
    uint2 pl[3];
    split4(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + woff + J * (KPP * ROWB)) = pl[p];
  }
  // address (plane 0, first transpose read) of the fragment of tile columns c0 .. c0+31 (c0 % 32 == 0):
  // lane L of a 16-lane group addresses row (L >> 2), column quad (L & 3) of the group's 4 x 16 block
  static __device__ __forceinline__ uint32_t frag_addr(int c0, int lane) {
    const int L = lane & 15, k = 8 * (lane >> 5) + (L >> 2);
    return k * ROWB + ((((c0 >> 5) ^ swz(k)) & (NC - 1)) << 6) + (((lane >> 4) & 1) << 5) + ((L & 3) << 3);
  }
  // k rows 8 (lane >> 5) + {0..3} and {4..7}
  static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, uint32_t addr, int plane) {
    const char* q = S + plane * PLANE + addr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * ROWB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
};

// ---- "col" operand that is COMPUTED while it is staged: the trunk's input stage (cb_trunk_input_bwd_multi_f32 with the mix gradients folded,
// cb_spmm_csr_store_bwd_mix_f32) as the A operand of the input Linear's weight gradient ----------------------------------------------------------
//   value[r][c] = (x0_bits[r] has bit c) ? keep(seed, r, c) * g[r][c] + mfold[r][c] : 0        (autograd of GCN.py:104-110 + res_tricks.py:23)
// g = dL/d dropout(X0) (the dX of the first GCNConv), mfold = the folded mix gradients, x0_bits = mask words of (X0 > 0): [rows][4] 64-bit words, word k
// bit l <-> column 4 l + k.  256 columns, NT = 512 staging threads: one wavefront per k row (TPR = 64), so the row's four mask words are wave-uniform
// (scalar loads) and bit `lane` of word i is this lane's column 4 lane + i.  The second stream's registers live in the operand (register prefetch depth 1).
// cs[i]: running column sums of the staged values (the input Linear's bias gradient), columns 4 (t % 64) + i over the rows this thread stages.
template <int NT = 512>
struct ColOperandInStage {
  static constexpr int C = 256;
  static constexpr int ROWB = C * 2, PLANE = 16 * ROWB, BYTES = 3 * PLANE;
  static constexpr int TPR = C / 4, KPP = NT / TPR, NV = 16 / KPP;
  static_assert(TPR == 64 && KPP >= 4 && KPP % 4 == 0 && NV >= 1, "one wavefront per k row");
  static constexpr int NC = C / 32;
  static __device__ __forceinline__ int swz(int k) { return k & 3; }
  uint32_t voff[NV];
  uint32_t woff;
  bool cok;
  static constexpr bool HAS_SC = false;
  float sc[1];
  int64_t mdelta;                       // (mfold - g) in elements: the second stream is read at the first one's offsets
  const unsigned long long* bits;       // mask words of this block's first row
  float4 fm[NV];
  unsigned long long bw[NV][4];
  uint64_t dseed;
  uint32_t dthresh;
  float dscale;
  int64_t drow, dtot, dwidth;
  int dcol, kw;
  float cs[4];
  __device__ __forceinline__ void set_stage(const float* g, const float* mfold, const unsigned long long* bits_block, const DropSpec& d, int64_t r_begin, int64_t total,
                                            int t) {
    mdelta = mfold - g;
    bits = bits_block;
    dseed = d.seed_dev ? d.seed + *d.seed_dev : d.seed;
    dthresh = d.thresh;
    dscale = d.scale;
    kw = __builtin_amdgcn_readfirstlane(t / TPR);
    drow = d.row0 + r_begin + kw;
    dtot = total;
    dwidth = d.width;
    dcol = (t % TPR) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = 0.f;
  }
  __device__ __forceinline__ void init(int64_t ld, int cols_left, int t) {
    const int k = t / TPR, nq = t % TPR;
    cok = nq * 4 < cols_left;
#pragma unroll
    for (int j = 0; j < NV; ++j) voff[j] = (uint32_t)((k + KPP * j) * ld + (cok ? nq * 4 : 0));
    woff = k * ROWB + ((((nq >> 3) ^ swz(k)) & (NC - 1)) << 6) + ((nq & 7) << 3);
  }
  template <int J>
  __device__ __forceinline__ void load(float4 (&f)[NV], const float* __restrict__ base, int64_t /*ld*/, int64_t k_left, const float* __restrict__, int /*t*/) {
    const int k = kw + KPP * J;
    const bool kin = k < k_left;
    const float* p = base + (kin ? voff[J] : 0u);
    f[J] = *reinterpret_cast<const float4*>(p);
    fm[J] = *reinterpret_cast<const float4*>(p + mdelta);
    const unsigned long long* b = bits + ((dtot - k_left) + (kin ? k : 0)) * 4;      // (wave-uniform address)
#pragma unroll
    for (int i = 0; i < 4; ++i) bw[J][i] = b[i];
  }
  template <int J>
  __device__ __forceinline__ void stage(const float4 (&f)[NV], char* __restrict__ S, int64_t k_left, int t) {
    const bool live = cok && kw + KPP * J < k_left;
    const int lane = t & 63;
    float mk[4];      // (no p == 0 branch: the K loop body stays one basic block; the entry point asks for p > 0)
    keep4(dseed, ((drow + KPP * J + (dtot - k_left)) * dwidth + dcol) >> 2, dthresh, dscale, mk);
    const float gv[4] = {f[J].x, f[J].y, f[J].z, f[J].w}, mv[4] = {fm[J].x, fm[J].y, fm[J].z, fm[J].w};
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tsum = mul_rounded(gv[i], mk[i]) + mv[i];
      v[i] = (live && ((bw[J][i] >> lane) & 1ull)) ? tsum : 0.f;
      cs[i] += v[i];
    }
    uint2 pl[3];
    split4(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + p * PLANE + woff + J * (KPP * ROWB)) = pl[p];
  }
  static __device__ __forceinline__ uint32_t frag_addr(int c0, int lane) {
    const int L = lane & 15, k = 8 * (lane >> 5) + (L >> 2);
    return k * ROWB + ((((c0 >> 5) ^ swz(k)) & (NC - 1)) << 6) + (((lane >> 4) & 1) << 5) + ((L & 3) << 3);
  }
  static __device__ __forceinline__ bf16x8 frag(const char* __restrict__ S, uint32_t addr, int plane) {
    const char* q = S + plane * PLANE + addr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * ROWB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
};

// The producer side of the software pipeline: registers fa / fb hold K step s+1 on entry to the MFMAs of step s; piece p
// stages its float4 into the LDS stage `dst` (if step s+1 exists) and re-fills it with step s+2 (if that exists).
template <class OPA, class OPB>
struct Producer {
  OPA& oa;
  OPB& ob;
  float4 (&fa)[OPA::NV];
  float4 (&fb)[OPB::NV];
  char* dstA;                  // LDS stage that receives step s+1
  char* dstB;
  const float* abase;          // operand bases of step s+2
  const float* bbase;
  const float* bscale;
  int64_t lda, ldb;
  int64_t left1, left2;        // operand rows / k left from the staged / the loaded step on (left1 <= 0: stage zeros; left2 > 0)
  int t;
  bool do_stage = true;        // compile-time constant at every use (prologue only)
  static constexpr int P = OPA::NV + OPB::NV;
  // branch-free on purpose: the K loop body must stay one basic block so that the compiler's vmcnt bookkeeping can let
  // older loads be consumed while younger ones are still in flight (a step that does not exist stages zeros into a stage
  // nobody reads / re-loads a clamped, valid address)
  template <int PIECE>
  __device__ __forceinline__ void one() {
    if constexpr (PIECE < OPA::NV) {
      if (do_stage) oa.template stage<PIECE>(fa, dstA, left1, t);
      oa.template load<PIECE>(fa, abase, lda, left2, nullptr, t);
    } else {
      constexpr int J = PIECE - OPA::NV;
      if (do_stage) ob.template stage<J>(fb, dstB, left1, t);
      ob.template load<J>(fb, bbase, ldb, left2, bscale, t);
    }
  }
  template <int G, int NG, int... Is>
  __device__ __forceinline__ void group_impl(std::integer_sequence<int, Is...>) {
    ((Is * NG / P == G ? one<Is>() : void()), ...);
  }
  // pieces scheduled after MFMA group G of NG
  template <int G, int NG>
  __device__ __forceinline__ void group() { group_impl<G, NG>(std::make_integer_sequence<int, P>{}); }
};

// ---- one K step (16) of a 64 x (32 WTN) wave tile ---------------------------------------------------
// OPA / OPB: RowOperand or ColOperand; aaddr[i] / baddr[j]: fragment addresses of the wave's 32-row / 32-column blocks.
// Limb products in increasing magnitude; planes are read just before their first use; consecutive MFMAs go to different
// accumulators; one slice of the producer's work after every group of four MFMAs.
template <int WTN, class OPA, class OPB, class PR>
__device__ __forceinline__ void limb_tile_step(const char* __restrict__ As, const char* __restrict__ Bs, const uint32_t (&aaddr)[2],
                                               const uint32_t (&baddr)[WTN], f32x16 (&acc)[2][WTN], PR& pr) {
  constexpr int NH = WTN / 2, NG = 6 * NH;   // column halves of the wave tile; MFMA groups (of four) per K step
#define CB_MFMA4(A_, B_, H_, G_)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
      acc[i][2 * H_ + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][2 * H_ + j], 0, 0, 0); \
  pr.template group<G_, NG>();
#define CB_HALF(H_)                                                                                                          \
  {                                                                                                                          \
    bf16x8 b_hi[2], b_mid[2], b_lo[2];                                                                                       \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_lo[i] = OPA::frag(As, aaddr[i], 2); }                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_hi[j] = OPB::frag(Bs, baddr[2 * H_ + j], 0);                             \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_hi[i] = OPA::frag(As, aaddr[i], 0); }                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_lo[j] = OPB::frag(Bs, baddr[2 * H_ + j], 2);                             \
    if (H_ == 0) { _Pragma("unroll") for (int i = 0; i < 2; ++i) a_mid[i] = OPA::frag(As, aaddr[i], 1); }                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b_mid[j] = OPB::frag(Bs, baddr[2 * H_ + j], 1);                            \
    CB_MFMA4(a_lo, b_hi, H_, H_ * 6 + 0)                                                                                     \
    CB_MFMA4(a_hi, b_lo, H_, H_ * 6 + 1)                                                                                     \
    CB_MFMA4(a_mid, b_mid, H_, H_ * 6 + 2)                                                                                   \
    CB_MFMA4(a_mid, b_hi, H_, H_ * 6 + 3)                                                                                    \
    CB_MFMA4(a_hi, b_mid, H_, H_ * 6 + 4)                                                                                    \
    CB_MFMA4(a_hi, b_hi, H_, H_ * 6 + 5)                                                                                     \
  }
  bf16x8 a_hi[2], a_mid[2], a_lo[2];
  static_assert(NH >= 1 && NH <= 4, "wave tiles of 64, 128 or 256 columns");
  CB_HALF(0)
  if constexpr (NH > 1) CB_HALF(1)
  if constexpr (NH > 2) CB_HALF(2)
  if constexpr (NH > 3) CB_HALF(3)
#undef CB_HALF
#undef CB_MFMA4
}

template <int WTN>
__device__ __forceinline__ void zero_acc_n(f32x16 (&acc)[2][WTN]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// The K loop shared by the NN / TN contractions and the top-K score sweep.  a0 / b0: operand bases of K step 0 (uniform); astep / bstep: their advance per K step
// in elements; total = length of the reduction axis covered by this block.  PD = register prefetch depth: while the MFMAs
// of step s run, step s+1 is staged from registers and step s+1+PD is requested from memory, i.e. PD K steps of loads are
// in flight per wavefront (the ring of register slots is indexed at compile time, hence the PD-fold unrolled loop; the
// step count is rounded up to a multiple of PD — the surplus steps multiply staged zeros).
template <int WTN, int PD, class OA, class OB>
__device__ __forceinline__ void limb_k_loop(OA& oa, OB& ob, char* __restrict__ smem, const float* __restrict__ a0, int64_t astep,
                                            int64_t lda, const float* __restrict__ b0, int64_t bstep, int64_t ldb,
                                            const float* __restrict__ bscale0, int64_t total, const uint32_t (&aaddr)[2],
                                            const uint32_t (&baddr)[WTN], f32x16 (&acc)[2][WTN], int t) {
  constexpr int STAGE = OA::BYTES + OB::BYTES;
  float4 fa[PD][OA::NV], fb[PD][OB::NV];
  float scs[PD][OB::NV];   // per-slot copies of the col operand's raw scales (ob.sc is the working copy)
  if (total <= 0) return;
  const int64_t nk = (total + KS - 1) / KS;
  auto clampi = [&](int64_t step) { return step < nk ? step : nk - 1; };   // loads of steps past the end re-read the last one
  {  // prologue: step 0 -> LDS stage 0; steps 1..PD -> register slots 1 % PD .. PD % PD
    Producer<OA, OB> p0{oa, ob, fa[0], fb[0], smem, smem + OA::BYTES, a0, b0, bscale0, lda, ldb, 0, total, t, false};
    p0.template group<0, 1>();                        // load step 0
    p0.left1 = total; p0.do_stage = true;
    p0.template group<0, 1>();                        // stage it (and load it once more: keeps one() branch-free)
#pragma unroll
    for (int d = 1; d <= PD; ++d) {
      const int64_t st = clampi(d);
      Producer<OA, OB> pd{oa, ob, fa[d % PD], fb[d % PD], smem, smem, a0 + st * astep, b0 + st * bstep,
                          bscale0 ? bscale0 + st * KS : nullptr, lda, ldb, 0, total - st * KS, t, false};
      pd.template group<0, 1>();
#pragma unroll
      for (int j = 0; j < OB::NV; ++j) scs[d % PD][j] = ob.sc[OB::HAS_SC ? j : 0];
    }
  }
  __syncthreads();
  for (int64_t kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int64_t kt = kt0 + u;
      char* cur = smem + (kt & 1) * STAGE;
      char* nxt = smem + ((kt + 1) & 1) * STAGE;
      const int slot = (u + 1) % PD;   // holds step kt + 1; re-filled with step kt + 1 + PD
#pragma unroll
      for (int j = 0; j < OB::NV; ++j) if (OB::HAS_SC) ob.sc[j] = scs[slot][j];
      const int64_t st = clampi(kt + 1 + PD);
      Producer<OA, OB> pr{oa, ob, fa[slot], fb[slot], nxt, nxt + OA::BYTES, a0 + st * astep, b0 + st * bstep,
                          bscale0 ? bscale0 + st * KS : nullptr, lda, ldb, total - (kt + 1) * KS, total - st * KS, t};
      limb_tile_step<WTN, OA, OB>(cur, cur + OA::BYTES, aaddr, baddr, acc, pr);
#pragma unroll
      for (int j = 0; j < OB::NV; ++j) if (OB::HAS_SC) scs[slot][j] = ob.sc[j];
      __syncthreads();
    }
  }
}

}  // namespace cb
