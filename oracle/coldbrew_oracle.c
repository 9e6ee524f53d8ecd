/* TEST INFRASTRUCTURE ONLY — plain C restatement of the integer/graph work and of the
 * aggregation of Cold Brew's TeacherGNN path, used (a) to pin the CSR build bit-exactly against
 * the numpy oracle and the HIP ingest, (b) as the `cpu_baseline` leg of bench.py (kind "port":
 * the reference's own aggregation lives in dgl==0.7.0, absent from the tree — PARITY UNPINNED at
 * that boundary, see oracle/coldbrew_oracle.py).  Never linked into the product library.
 *
 * Follows (paths relative to the reference root):
 *   orc_csr_from_coo   GNN_model/GCN.py:92-95   src = edge_index[0], dst = edge_index[1], multigraph
 *   orc_deg_norm       GNN_model/GCN.py:206-208,243-245   clamp(deg,1)^-1/2
 *   orc_spmm_csr       GNN_model/GCN.py:198,238,250,253   out[v] = act(scale[v] * sum h[src] + bias)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread count of the aggregation (bench.py reports it as `cores`); 0 = OpenMP default */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* rows = major key, in-row order = ascending minor key (stable two-pass counting sort) */
int orc_csr_from_coo(const int64_t* major, const int64_t* minor, int64_t E, int64_t N, int64_t* rowptr, int32_t* col) {
  int64_t* cnt = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
  int64_t* tmp_major = (int64_t*)malloc((size_t)(E > 0 ? E : 1) * sizeof(int64_t));
  int64_t* tmp_minor = (int64_t*)malloc((size_t)(E > 0 ? E : 1) * sizeof(int64_t));
  if (!cnt || !tmp_major || !tmp_minor) return -1;
  /* pass 1: stable sort by minor */
  for (int64_t e = 0; e < E; ++e) {
    if (minor[e] < 0 || minor[e] >= N || major[e] < 0 || major[e] >= N) { free(cnt); free(tmp_major); free(tmp_minor); return -2; }
    cnt[minor[e] + 1]++;
  }
  for (int64_t v = 0; v < N; ++v) cnt[v + 1] += cnt[v];
  for (int64_t e = 0; e < E; ++e) {
    int64_t p = cnt[minor[e]]++;
    tmp_major[p] = major[e];
    tmp_minor[p] = minor[e];
  }
  /* pass 2: stable sort by major */
  memset(cnt, 0, ((size_t)N + 1) * sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) cnt[tmp_major[e] + 1]++;
  for (int64_t v = 0; v < N; ++v) cnt[v + 1] += cnt[v];
  memcpy(rowptr, cnt, ((size_t)N + 1) * sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) {
    int64_t p = cnt[tmp_major[e]]++;
    col[p] = (int32_t)tmp_minor[e];
  }
  free(cnt); free(tmp_major); free(tmp_minor);
  return 0;
}

void orc_deg_norm(const int64_t* rowptr, int64_t N, float* norm) {
  for (int64_t v = 0; v < N; ++v) {
    int64_t d = rowptr[v + 1] - rowptr[v];
    float x = (float)(d < 1 ? 1 : d);
    norm[v] = 1.0f / sqrtf(x);
  }
}

void orc_spmm_csr(const int64_t* rowptr, const int32_t* col, int64_t N, const float* h, int64_t d, const float* scale,
                  const float* bias, int relu, float* out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t v = 0; v < N; ++v) {
    float* o = out + v * d;
    for (int64_t c = 0; c < d; ++c) o[c] = 0.f;
    for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) {
      const float* s = h + (int64_t)col[j] * d;
      for (int64_t c = 0; c < d; ++c) o[c] += s[c];
    }
    const float sc = scale ? scale[v] : 1.f;
    for (int64_t c = 0; c < d; ++c) {
      float t = o[c] * sc + (bias ? bias[c] : 0.f);
      o[c] = (relu && t < 0.f) ? 0.f : t;
    }
  }
}
