// cppf_shot.hip -- SHOT352 descriptors + normals on gfx950: replaces shot.compute / estimate_normal
// (src_shot/shot.cpp:12-100, a pybind11 wrapper over PCL 1.9.1 NormalEstimation + SHOTEstimation).
//
// Pipeline (all per batch of scenes):
//   shot_cells   workgroup/scene: uniform cell grid with cell edge >= the search radius, counting sort of the
//                scene's points by cell in LDS, points re-written in cell order (so a radius query reads 9
//                contiguous runs of points -- 3 x-adjacent cells for each of the 3x3 (y,z) neighbours);
//   shot_cov     wavefront/query: lanes sweep the 9 runs (coalesced), float64 partial sums of the normal
//                covariance and of the distance-weighted LRF covariance, column-summed through LDS in a fixed
//                order;
//   shot_eig     thread/query: two 3x3 Jacobi eigen-solves (every lane busy), normal orientation;
//   shot_hist    wavefront/query: radius neighbours compacted (ballot + prefix popcount) into an LDS list, LRF
//                sign disambiguation by ballots, interpolated 352-bin histogram in LDS (ds_add_f32), L2 norm.
//
// Numerics follow PCL 1.9.1's algorithm (features/impl/normal_3d.hpp, shot_lrf.hpp, shot.hpp) with the
// deviations listed in oracle/shot_oracle.c (float64 covariance about the query point, Jacobi eigen-solver,
// index-ordered neighbours).  Parity with PCL itself is UNPINNED (PCL is absent from the image).
#include "cppf_common.h"

#define SHOT_LEN 352
#define NR_BINS 10
#define MAX_SECTORS 32
#ifndef SH_LCAP
#define SH_LCAP 1024      // neighbour-list capacity per query (falls back to a full rescan above it)
#endif
// The histogram is kept in SH_COPIES lane-interleaved copies: neighbouring lanes hold spatially neighbouring points,
// which fall into the same bins, and LDS atomics on one address serialise (measured: ~50 passes per ds_add_f32 with
// a single copy).  Odd stride: the copies start in different banks.
#ifndef SH_COPIES
#define SH_COPIES 1
#endif
#define SH_STRIDE 353
#ifndef SHOT_DBG
#define SHOT_DBG 0     // timing probes (scratch/shot_sections.sh): 1 = no sign pass, 2 = no accumulation, 4 = no normalisation, 8 = the compiler's float division, 16 = the general tie pass, 32 = float64 sign pass, 512 = no general tie pass;
                       // shot_cov: 64 = no covariance sums, 128 = no reduction, 256 = no neighbour list to the workspace
#endif
// Neighbour lists handed from shot_cov to shot_hist through the workspace (entries per query; longer lists are rebuilt)
#define NBR_CAP 512

__device__ __forceinline__ int find_scene_pt(const int32_t* __restrict__ off, int B, int i) {
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}


// Cyclic Jacobi, symmetric 3x3, float64; eigenvalues ascending, eigenvectors in columns.  Same operation
// sequence as oracle/shot_oracle.c:jacobi3 (only + - * / sqrt: bit-identical on CPU and GPU).
struct Eig3 {
  double w[3];
  double v[3][3];
};

#define JROT(P, Q, R)                                                                 \
  {                                                                                   \
    const double apq = a[P][Q];                                                       \
    if (!(fabs(apq) < 1e-300)) {                                                      \
      const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);                         \
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
      const double c = 1.0 / sqrt(t * t + 1.0);                                       \
      const double s = t * c;                                                         \
      a[P][P] = a[P][P] - t * apq;                                                    \
      a[Q][Q] = a[Q][Q] + t * apq;                                                    \
      a[P][Q] = 0.0;                                                                  \
      a[Q][P] = 0.0;                                                                  \
      const double arp = a[R][P], arq = a[R][Q];                                      \
      a[R][P] = c * arp - s * arq;                                                    \
      a[P][R] = a[R][P];                                                              \
      a[R][Q] = s * arp + c * arq;                                                    \
      a[Q][R] = a[R][Q];                                                              \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                 \
        const double vip = e.v[i][P], viq = e.v[i][Q];                                \
        e.v[i][P] = c * vip - s * viq;                                                \
        e.v[i][Q] = s * vip + c * viq;                                                \
      }                                                                               \
    }                                                                                 \
  }

__device__ __forceinline__ void eig_swap_cols(Eig3& e, double d[3], int i, int j) {
  if (d[i] > d[j]) {
    const double t = d[i]; d[i] = d[j]; d[j] = t;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const double u = e.v[r][i]; e.v[r][i] = e.v[r][j]; e.v[r][j] = u; }
  }
}

__device__ Eig3 jacobi3(const double c[6]) {
  double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  Eig3 e;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) e.v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    JROT(0, 1, 2)
    JROT(0, 2, 1)
    JROT(1, 2, 0)
  }
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  // stable bubble order (0,1),(1,2),(0,1) == the oracle's index bubble sort
  eig_swap_cols(e, d, 0, 1);
  eig_swap_cols(e, d, 1, 2);
  eig_swap_cols(e, d, 0, 1);
  e.w[0] = d[0]; e.w[1] = d[1]; e.w[2] = d[2];
  return e;
}

__device__ __forceinline__ float sqdist3(float px, float py, float pz, float qx, float qy, float qz) {
  const float dx = px - qx, dy = py - qy, dz = pz - qz;
  return (dx * dx + dy * dy) + dz * dz;
}

// ---------------------------------------------------------------------------------------------
// SHOT352
// ---------------------------------------------------------------------------------------------
#define RAD_45 0.78539816339744830961566084581988
#define RAD_90 1.5707963267948966192313216916398
#define RAD_135 2.3561944901923449288469825374596
#define RAD_PI_7_8 2.7488935718910690836548129603691

// (atan2 for the two interpolation angles: atan2_poly, cppf_common.h)

// One neighbour's quadrilinear contribution (pcl::SHOTEstimation::interpolateSingleChannel, PCL 1.9.1
// features/impl/shot.hpp), float arithmetic, written without divergent arms: every interpolation axis (cosine bin,
// radial shell, elevation, azimuth) yields one weight 1 - |d| for the home bin and one predicated atomic |d| for the
// neighbouring bin; divisions by the constant bin widths are multiplications by their reciprocals and the two square
// roots are the 1-ulp hardware ones.  The decisions (sector, shell, hemisphere, bin) are the reference's.
#define NR_COLOR_BINS 30                                   // nr_color_bins_ of pcl::SHOTColorEstimation
#define COLOR_OFF (MAX_SECTORS * (NR_BINS + 1))            // the colour channel follows the 352 shape entries
#define SHOT_COLOR_LEN (COLOR_OFF + MAX_SECTORS * (NR_COLOR_BINS + 1))   // 1344

// COLOR: second channel of SHOT1344 (interpolateDoubleChannel): `bdc` is the neighbour's colour bin coordinate; it
// lands in the same sector with the same spatial interpolation terms, only the step along the bins is its own.
template <bool COLOR>
__device__ __forceinline__ void shot_accumulate(float px, float py, float pz, float qx, float qy, float qz, float d2,
                                                float nqx, float nqy, float nqz, const float* rf, float radius,
                                                uint32_t* shot, float fx_scale, float bdc = 0.0f) {
  if (!isfinite(nqx) || !isfinite(nqy) || !isfinite(nqz)) return;
  float cosd = (nqx * rf[6] + nqy * rf[7]) + nqz * rf[8];
  cosd = fminf(fmaxf(cosd, -1.0f), 1.0f);
  float bd = (1.0f + cosd) * (0.5f * NR_BINS);
  const float distance = __builtin_amdgcn_sqrtf(d2);
  if (distance < 1e-15f) return;
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  float xf = (dx * rf[0] + dy * rf[1]) + dz * rf[2];
  float yf = (dx * rf[3] + dy * rf[4]) + dz * rf[5];
  float zf = (dx * rf[6] + dy * rf[7]) + dz * rf[8];
  yf = (fabsf(yf) < 1e-30f) ? 0.0f : yf;
  xf = (fabsf(xf) < 1e-30f) ? 0.0f : xf;
  zf = (fabsf(zf) < 1e-30f) ? 0.0f : zf;
  const float r12 = radius * 0.5f, r14 = radius * 0.25f, r34 = radius * 0.75f, inv_r12 = 2.0f / radius;
  const bool outer = distance > r12;
  const int bit4 = ((yf > 0.0f) || ((yf == 0.0f) && (xf < 0.0f))) ? 1 : 0;
  const int bit3 = ((xf > 0.0f) || ((xf == 0.0f) && (yf > 0.0f))) ? (bit4 ^ 1) : bit4;
  int desc = (bit4 << 4) + (bit3 << 3);
  const float fx = fabsf(xf), fy = fabsf(yf);
  if ((xf * yf > 0.0f) || (xf == 0.0f)) desc += (fx >= fy) ? 0 : 4;
  else desc += (fx > fy) ? 4 : 0;
  desc += (zf > 0.0f) ? 1 : 0;
  desc += outer ? 2 : 0;
  const float stepf = floorf(bd + 0.5f);
  const int step = (int)stepf;
  const int home = desc * (NR_BINS + 1) + step;
  bd -= stepf;
  float w = 1.0f - fabsf(bd);
  {  // cosine bins wrap inside the volume
    int nb = step + ((bd > 0.0f) ? 1 : NR_BINS - 1);
    nb -= (nb >= NR_BINS) ? NR_BINS : 0;
    atomicAdd(&shot[desc * (NR_BINS + 1) + nb], (uint32_t)__float2uint_rn((fabsf(bd)) * fx_scale));
  }
  constexpr int CS = NR_COLOR_BINS + 1;
  int chome = 0;
  float wc = 0.0f;
  if (COLOR) {
    const float cstepf = floorf(bdc + 0.5f);
    const int cstep = (int)cstepf;
    chome = COLOR_OFF + desc * CS + cstep;
    bdc -= cstepf;
    wc = 1.0f - fabsf(bdc);
    int nb = cstep + ((bdc > 0.0f) ? 1 : NR_COLOR_BINS - 1);
    nb -= (nb >= NR_COLOR_BINS) ? NR_COLOR_BINS : 0;
    atomicAdd(&shot[COLOR_OFF + desc * CS + nb], (uint32_t)__float2uint_rn((fabsf(bdc)) * fx_scale));
  }
  {  // radial shells: the neighbour exists only towards the shell boundary at radius / 2
    const float rd = (distance - (outer ? r34 : r14)) * inv_r12;
    w += 1.0f - fabsf(rd);
    wc += 1.0f - fabsf(rd);
    const bool has = outer ? !(distance > r34) : !(distance < r14);
    if (has) atomicAdd(&shot[home + (outer ? -2 : 2) * (NR_BINS + 1)], (uint32_t)__float2uint_rn((fabsf(rd)) * fx_scale));
    if (COLOR && has) atomicAdd(&shot[chome + (outer ? -2 : 2) * CS], (uint32_t)__float2uint_rn((fabsf(rd)) * fx_scale));
  }
  {  // elevation: two hemispheres, neighbour only towards the equator
    const float inc = atan2_poly(__builtin_amdgcn_sqrtf(xf * xf + yf * yf), zf);      // [0, pi], exact at the poles
    const float R45 = (float)RAD_45, R90 = (float)RAD_90, R135 = (float)RAD_135;
    const bool lower = inc > R90 || (inc == R90 && zf <= 0.0f);
    const float id = (inc - (lower ? R135 : R45)) * (float)(1.0 / RAD_90);
    w += 1.0f - fabsf(id);
    wc += 1.0f - fabsf(id);
    const bool has = lower ? !(inc > R135) : !(inc < R45);
    if (has) atomicAdd(&shot[home + (lower ? 1 : -1) * (NR_BINS + 1)], (uint32_t)__float2uint_rn((fabsf(id)) * fx_scale));
    if (COLOR && has) atomicAdd(&shot[chome + (lower ? 1 : -1) * CS], (uint32_t)__float2uint_rn((fabsf(id)) * fx_scale));
  }
  if (yf != 0.0f || xf != 0.0f) {  // azimuth sectors wrap around
    const float az = atan2_poly(yf, xf);
    float ad = (az - (-(float)RAD_PI_7_8 + (float)RAD_45 * (float)(desc >> 2))) * (float)(1.0 / RAD_45);
    ad = fminf(fmaxf(ad, -0.5f), 0.5f);
    w += 1.0f - fabsf(ad);
    wc += 1.0f - fabsf(ad);
    const int sec = (desc + ((ad > 0.0f) ? 4 : MAX_SECTORS - 4)) & (MAX_SECTORS - 1);
    atomicAdd(&shot[sec * (NR_BINS + 1) + step], (uint32_t)__float2uint_rn((fabsf(ad)) * fx_scale));
    if (COLOR) atomicAdd(&shot[chome + (sec - desc) * CS], (uint32_t)__float2uint_rn((fabsf(ad)) * fx_scale));
  }
  atomicAdd(&shot[home], (uint32_t)__float2uint_rn((w) * fx_scale));
  if (COLOR) atomicAdd(&shot[chome], (uint32_t)__float2uint_rn((wc) * fx_scale));
}

#define CELL_CAP 16384     // cells per scene held in LDS by shot_cells (64 KiB of counters)
#define NSUM 19            // 9 normal sums + count, 6 LRF sums + weight sum + valid count (+1 pad)

struct CellHdr {
  float c0[3];
  float inv;       // 1 / cell edge
  int32_t d[3];
  int32_t ncell;
};

__device__ __forceinline__ int cell_coord(float p, float c0, float inv, int dim) {
  int c = (int)((p - c0) * inv);
  c = c < 0 ? 0 : c;
  return c >= dim ? dim - 1 : c;
}

// ---------------------------------------------------------------------------------------------
// shot_cells: per scene counting sort by cell.  cell index = (z * dy + y) * dx + x  (x fastest)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void shot_cells_kernel(const float* __restrict__ pts,
                                                          const int32_t* __restrict__ pt_off, float rmax,
                                                          CellHdr* __restrict__ hdrs,
                                                          int32_t* __restrict__ cell_start /* [B][CELL_CAP+1] */,
                                                          int32_t* __restrict__ sorted_idx,
                                                          float4* __restrict__ sorted_pts,
                                                          int32_t* __restrict__ scene_of) {
  __shared__ uint32_t s_cnt[CELL_CAP];
  __shared__ float s_red[16][6];
  __shared__ uint32_t s_wsum[16];
  __shared__ CellHdr s_h;
  const int b = blockIdx.x;
  const int p0 = pt_off[b], n = pt_off[b + 1] - p0;
  const float* sp = pts + 3 * (int64_t)p0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += 1024) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = sp[3 * i + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
    }
  if (wave_lane() == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_red[threadIdx.x >> 6][c] = mn[c]; s_red[threadIdx.x >> 6][3 + c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    CellHdr h;
    float ext[3];
    for (int c = 0; c < 3; ++c) {
      float lo = s_red[0][c], hi = s_red[0][3 + c];
      for (int w = 1; w < 16; ++w) { lo = fminf(lo, s_red[w][c]); hi = fmaxf(hi, s_red[w][3 + c]); }
      h.c0[c] = lo;
      ext[c] = (n > 0) ? hi - lo : 0.0f;
    }
    // cell edge >= rmax (with slack for float rounding of the cell coordinate); coarsen until the grid fits LDS
    float edge = rmax * 1.0005f;
    for (;;) {
      int64_t tot = 1;
      for (int c = 0; c < 3; ++c) { h.d[c] = (int)(ext[c] / edge) + 1; tot *= h.d[c]; }
      if (tot <= CELL_CAP) { h.ncell = (int)tot; break; }
      edge *= 1.26f;
    }
    h.inv = 1.0f / edge;
    s_h = h;
    hdrs[b] = h;
  }
  __syncthreads();
  const CellHdr h = s_h;
  for (int i = threadIdx.x; i < h.ncell; i += 1024) s_cnt[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const int cx = cell_coord(sp[3 * i], h.c0[0], h.inv, h.d[0]);
    const int cy = cell_coord(sp[3 * i + 1], h.c0[1], h.inv, h.d[1]);
    const int cz = cell_coord(sp[3 * i + 2], h.c0[2], h.inv, h.d[2]);
    atomicAdd(&s_cnt[(cz * h.d[1] + cy) * h.d[0] + cx], 1u);
  }
  __syncthreads();
  // exclusive scan of s_cnt[0..ncell): each thread owns a contiguous chunk
  const int per = (h.ncell + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(h.ncell, lo + per);
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += s_cnt[i];
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off);
    if (wave_lane() >= off) incl += o;
  }
  if (wave_lane() == 63) s_wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_wsum[w];
  uint32_t run = base + incl - sum;
  int32_t* cs = cell_start + (int64_t)b * (CELL_CAP + 1);
  for (int i = lo; i < hi; ++i) {
    const uint32_t c = s_cnt[i];
    cs[i] = (int32_t)run;
    s_cnt[i] = run;            // becomes the scatter cursor
    run += c;
  }
  if (threadIdx.x == 0) cs[h.ncell] = n;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float x = sp[3 * i], y = sp[3 * i + 1], z = sp[3 * i + 2];
    const int cx = cell_coord(x, h.c0[0], h.inv, h.d[0]);
    const int cy = cell_coord(y, h.c0[1], h.inv, h.d[1]);
    const int cz = cell_coord(z, h.c0[2], h.inv, h.d[2]);
    const uint32_t pos = atomicAdd(&s_cnt[(cz * h.d[1] + cy) * h.d[0] + cx], 1u);
    scene_of[p0 + i] = b;
    sorted_idx[p0 + pos] = i;
    sorted_pts[(int64_t)p0 + pos] = make_float4(x, y, z, __int_as_float(i));   // .w = original index
  }
}

// the 9 runs of cell-sorted points that can hold neighbours of a query in cell (cx, cy, cz)
struct Runs {
  int beg[9], end[9];
};

__device__ __forceinline__ void query_runs(const CellHdr& h, const int32_t* __restrict__ cs, float px, float py,
                                           float pz, Runs& r) {
  const int cx = cell_coord(px, h.c0[0], h.inv, h.d[0]);
  const int cy = cell_coord(py, h.c0[1], h.inv, h.d[1]);
  const int cz = cell_coord(pz, h.c0[2], h.inv, h.d[2]);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.d[0] - 1);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = cy + (k % 3) - 1, zz = cz + (k / 3) - 1;
    if (yy < 0 || yy >= h.d[1] || zz < 0 || zz >= h.d[2]) { r.beg[k] = 0; r.end[k] = 0; continue; }
    const int row = (zz * h.d[1] + yy) * h.d[0];
    r.beg[k] = cs[row + x0];
    r.end[k] = cs[row + x1 + 1];
  }
}

// ---------------------------------------------------------------------------------------------
// shot_cov: float64 sums for the normal covariance (radius rn) and the LRF covariance (radius rs)
// ---------------------------------------------------------------------------------------------
// One pass over a neighbour (shared by the compacted-list and the direct-scan forms of shot_cov)
template <bool NORMAL_SUMS = true>
__device__ __forceinline__ void cov_accumulate(const float4 qv, float px, float py, float pz, float rn2, float rs2,
                                               float rs, double a[NSUM]) {
  const float qx = qv.x, qy = qv.y, qz = qv.z;
  const float d2 = sqdist3(px, py, pz, qx, qy, qz);
  const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
  if (NORMAL_SUMS && d2 < rn2) {
    a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z;
    a[6] += x; a[7] += y; a[8] += z; a[9] += 1.0;
  }
  if (d2 < rs2) {
    a[18] += 1.0;                                             // all in-radius points (incl. self)
    if (!(qx == px && qy == py && qz == pz)) {
      const double w = (double)rs - (double)sqrt_rn(d2);
      a[10] += w * (x * x); a[11] += w * (x * y); a[12] += w * (x * z);
      a[13] += w * (y * y); a[14] += w * (y * z); a[15] += w * (z * z);
      a[16] += w; a[17] += 1.0;
    }
  }
}

// The candidates of the 9 runs are first compacted to the ones inside the larger radius (about a third of them), so
// the float64 sums run over dense wavefronts; the compacted list (positions in the cell-sorted order, scan order) is
// also left in the workspace for shot_hist, which needs the same neighbours.
// wave_rank: the rank of each of the wavefront's keys (EMAX per lane; bkt < 0: no entry) among all of them, ascending.  The caller
// supplies a bucket per key, monotone in the key (0..63): a counting sort over the buckets (LDS atomics + one prefix scan) leaves
// exact 64-bit comparisons only against the keys of the own bucket.  buf: 128 + 128 + 2 x 64 EMAX words of LDS, free on entry
// (the caller's barrier precedes the call) and again on return.  nent (wave-uniform): entry slots >= nent hold no key on any lane and
// are skipped as a whole.  Returns the number of keys.
template <int EMAX>
__device__ __forceinline__ int wave_rank(int lane, uint32_t* buf, const unsigned long long (&key)[EMAX], const int (&bkt)[EMAX],
                                         int (&rank)[EMAX], int nent = EMAX) {
  uint32_t* s_hist = buf;                                              // [64] bucket counts
  uint32_t* s_start = s_hist + 64;                                     // [64] bucket starts
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_start + 64);   // [<= 64 EMAX] keys by bucket
  s_hist[lane] = 0;
  __syncthreads();
  int ep[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    ep[e] = 0;
    if (e < nent && bkt[e] >= 0) ep[e] = (int)atomicAdd(&s_hist[bkt[e]], 1u);
  }
  __syncthreads();
  int total;
  {
    const uint32_t cnt = s_hist[lane];
    const uint32_t incl = wave_inclusive_scan_u32(cnt);
    s_start[lane] = incl - cnt;
    total = (int)__builtin_amdgcn_readlane((int)incl, 63);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EMAX; ++e)
    if (e < nent && bkt[e] >= 0) s_key[s_start[bkt[e]] + ep[e]] = key[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    rank[e] = -1;
    if (e < nent && bkt[e] >= 0) {
      const int s0 = (int)s_start[bkt[e]], n_b = (int)s_hist[bkt[e]];
      int r = s0;
#pragma unroll 1
      for (int j = 0; j < n_b; ++j) r += (s_key[s0 + j] < key[e]) ? 1 : 0;
      rank[e] = r;
    }
  }
  __syncthreads();
  return total;
}

// The 64-bit key of the (distance, original index) order: non-negative floats order like their bit patterns.
__device__ __forceinline__ unsigned long long dist_index_key(float d2, int index) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | (uint32_t)index;
}

// The float32 sums of pcl::NormalEstimation (see shot_cov_kernel<PCL>) over the entries the lanes hold (EMAX per lane; ed >= rn2
// marks an unused entry) in (distance, index) order: ranks from wave_rank (64 distance buckets),
// addends written to LDS in rank order (windows of PCL_WIN ranks), nine lanes add one column each sequentially.
// s_raw: the wavefront's 5016-byte LDS buffer (free on entry; a barrier precedes the call).  Returns the j-th sum in lane j < 9.
template <int EMAX>
__device__ __forceinline__ void pcl_float_sums(int lane, double* s_raw, float rn2, const float (&ex)[EMAX], const float (&ey)[EMAX],
                                               const float (&ez)[EMAX], const float (&ed)[EMAX], const int (&ei)[EMAX],
                                               float& pcl_sum, int& mn, int nent = EMAX) {
  float* s_add = reinterpret_cast<float*>(s_raw);                      // [9][PCL_WIN] addends of a rank window, column-major
  constexpr int PCL_WIN = 136;                                         // 9 x 136 x 4 = 4896 bytes of the 5016
  static_assert(512 + 64 * EMAX * 8 <= NSUM * 33 * 8, "keys fit the buffer");
  const float bscale = 64.0f * __builtin_amdgcn_rcpf(rn2);     // (any positive scale gives the same ranks: buckets only pre-sort)
  unsigned long long key[EMAX];
  int eb[EMAX], er[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    eb[e] = (ed[e] < rn2) ? min(63, (int)(ed[e] * bscale)) : -1;
    key[e] = dist_index_key(ed[e], ei[e]);
  }
  mn = wave_rank<EMAX>(lane, reinterpret_cast<uint32_t*>(s_raw), key, eb, er, nent);
  for (int w0 = 0; w0 < mn; w0 += PCL_WIN) {
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      const int r = er[e] - w0;
      if (e < nent && er[e] >= 0 && r >= 0 && r < PCL_WIN) {
        const float x = ex[e], y = ey[e], z = ez[e];
        s_add[0 * PCL_WIN + r] = x * x; s_add[1 * PCL_WIN + r] = x * y; s_add[2 * PCL_WIN + r] = x * z;
        s_add[3 * PCL_WIN + r] = y * y; s_add[4 * PCL_WIN + r] = y * z; s_add[5 * PCL_WIN + r] = z * z;
        s_add[6 * PCL_WIN + r] = x; s_add[7 * PCL_WIN + r] = y; s_add[8 * PCL_WIN + r] = z;
      }
    }
    __syncthreads();
    if (lane < 9) {
      const int n_w = min(PCL_WIN, mn - w0);
      const float* col = s_add + lane * PCL_WIN;
      int r = 0;
      for (; r + 8 <= n_w; r += 8) {             // eight addends requested at once, added one after the other
        const float4 v0 = *reinterpret_cast<const float4*>(col + r), v1 = *reinterpret_cast<const float4*>(col + r + 4);
        pcl_sum += v0.x; pcl_sum += v0.y; pcl_sum += v0.z; pcl_sum += v0.w;
        pcl_sum += v1.x; pcl_sum += v1.y; pcl_sum += v1.z; pcl_sum += v1.w;
      }
      for (; r + 4 <= n_w; r += 4) {
        const float4 v = *reinterpret_cast<const float4*>(col + r);
        pcl_sum += v.x; pcl_sum += v.y; pcl_sum += v.z; pcl_sum += v.w;
      }
      for (; r < n_w; ++r) pcl_sum += col[r];
    }
    __syncthreads();
  }
}

// PCL (template): the normal's sums in pcl::NormalEstimation's arithmetic instead (src_shot/shot.cpp:25-32, 66-72 ->
// computeMeanAndCovarianceMatrix of PCL 1.9.1): ONE pass of float32 sums of the RAW coordinates and their products over the
// neighbours in the order a sorted radius search returns them, (distance, index) ascending -- float32 sums of numbers near 0.6 m^2
// that cancel to a covariance of 1e-4 m^2 depend on that order at the 1e-3 level, so the order is part of the arithmetic
// (oracle/shot_oracle.c: pcl_normal, mode 1).  The neighbours of the list are ranked by a counting sort on 64 distance buckets
// (squared distances of surface points are uniform in [0, r^2): ~1.4 per bucket) with exact comparisons inside a bucket, their
// nine addends are written to LDS in rank order, and nine lanes add one column each, sequentially.  sums[0..8] then hold those
// float32 sums (as doubles), sums[9] the neighbour count.  This kernel ranks lists of up to 128 neighbours (two entries per lane:
// its registers keep eight wavefronts per SIMD); a longer list keeps the float64 sums and marks sums[9] with + 0.5 for
// shot_pcl_long_kernel, which redoes the marked queries from the workspace copy of the list (up to NBR_CAP = 512 neighbours: a
// 2 mm voxel cloud has at most ~314 inside 2 cm; beyond that the float64 sums stay).
#ifndef PCL_EMAX_SHORT
#define PCL_EMAX_SHORT 2
#endif
#define PCL_EMAX_LONG (NBR_CAP / 64)
#define PCL_LONG_SCAN 16
template <bool PCL>
__global__ __launch_bounds__(64, 8) void shot_cov_kernel(int B, const float* __restrict__ pts,
                                                      const int32_t* __restrict__ pt_off,
                                                      const CellHdr* __restrict__ hdrs,
                                                      const int32_t* __restrict__ cell_start,
                                                      const float4* __restrict__ sorted_pts,
                                                      const int32_t* __restrict__ scene_of, float rn, float rs,
                                                      double* __restrict__ sums /* [Ntot][NSUM] */,
                                                      int32_t* __restrict__ nbr_list /* [Ntot][NBR_CAP] */,
                                                      int32_t* __restrict__ nbr_cnt /* [Ntot] */) {
  // The reduction buffer holds 32 partials per column (the two lane halves are added through a cross-lane exchange first), rows
  // padded by one entry so that the columns start in different banks.  The same LDS first holds the compacted neighbours.
  // 5 KiB per wavefront: eight wavefronts per SIMD fit (at 9.9 KiB -- 64 partials per column -- it was four, and this kernel
  // spends 40 % of its cycles waiting for its gathers: throughput follows occupancy).
  constexpr int COV_LIST = NSUM * (32 + 1) * 8 / 16;           // neighbours the LDS list holds (313); beyond: direct sums
  __shared__ __attribute__((aligned(16))) double s_raw[NSUM * (32 + 1)];
  double (*s_part)[32 + 1] = reinterpret_cast<double (*)[32 + 1]>(s_raw);
  float4* s_nb = reinterpret_cast<float4*>(s_raw);
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const int b = scene_of[qi];
  const int p0 = pt_off[b];
  const CellHdr h = hdrs[b];
  const int32_t* cs = cell_start + (int64_t)b * (CELL_CAP + 1);
  const float4* sp = sorted_pts + (int64_t)p0;
  const float px = pts[3 * (int64_t)qi], py = pts[3 * (int64_t)qi + 1], pz = pts[3 * (int64_t)qi + 2];
  const float rn2 = rn * rn, rs2 = rs * rs, rm2 = fmaxf(rn2, rs2);
  Runs runs;
  query_runs(h, cs, px, py, pz, runs);
  double a[NSUM];
#pragma unroll
  for (int c = 0; c < NSUM; ++c) {
    a[c] = 0.0;
    asm("" : "+v"(a[c]));        // one zero per sum, here: otherwise every branch below re-creates the zeros it might need
  }
  // The candidates of the 9 runs are tested in float32 and the hits (about a fifth of them) compacted (ballot + prefix popcount)
  // into LDS -- and into the workspace for shot_hist, which needs the same neighbours -- so that the float64 covariance sums run
  // over a dense list (two wavefront passes for the usual ~90 neighbours instead of seven to nine over the candidates).
  int base[10];
  base[0] = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) base[k + 1] = base[k] + (runs.end[k] - runs.beg[k]);
  const int C = base[9];
  auto cand = [&](int f) {
    int j = runs.beg[0] + f;
#pragma unroll
    for (int k = 1; k < 9; ++k) j = (f >= base[k]) ? runs.beg[k] + (f - base[k]) : j;
    return j;
  };
  int m = 0;
  int32_t* const my_list = nbr_list + (int64_t)qi * NBR_CAP;
  {
    // run by run (the same scan order as the flat range): the first 64 candidates of every run are requested up front -- nine
    // independent gathers in flight and no per-lane search for the run a flat index falls into; a run rarely has more than 64
    float4 q0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int j = runs.beg[k] + lane;
      q0[k] = sp[(unsigned)(j < runs.end[k] ? j : runs.beg[0])];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      for (int jb = runs.beg[k]; jb < runs.end[k]; jb += 64) {
        const int j = jb + lane;
        const bool valid = j < runs.end[k];
        const float4 qv = (jb == runs.beg[k]) ? q0[k] : sp[(unsigned)(valid ? j : jb)];
        const bool in = valid && sqdist3(px, py, pz, qv.x, qv.y, qv.z) < rm2;
        const unsigned long long mask = wave_ballot(in);
        if (in) {
          const unsigned pos = (unsigned)(m + lanes_below(mask));
          if (pos < (unsigned)COV_LIST) s_nb[pos] = qv;
          if (!(SHOT_DBG & 256) && pos < (unsigned)NBR_CAP && nbr_list) my_list[pos] = j;
        }
        m += __popcll(mask);
      }
    }
  }
  __syncthreads();
  // PCL: this lane's list entries with d2 < rn2 (coordinates, squared distance, scene-local index), kept in registers
  float ex[PCL_EMAX_SHORT], ey[PCL_EMAX_SHORT], ez[PCL_EMAX_SHORT], ed[PCL_EMAX_SHORT];
  int ei[PCL_EMAX_SHORT];
  const bool pcl_list = PCL && rn > 0.0f && m <= 64 * PCL_EMAX_SHORT;
  if (SHOT_DBG & 64) {
  } else if (pcl_list) {
#pragma unroll
    for (int e = 0; e < PCL_EMAX_SHORT; ++e) {
      // (every lane reads its slot -- 128 slots are inside the buffer -- and only ed says whether the entry exists)
      const int c = lane + 64 * e;
      const float4 qv = s_nb[c];
      const float d2 = sqdist3(px, py, pz, qv.x, qv.y, qv.z);
      ex[e] = qv.x; ey[e] = qv.y; ez[e] = qv.z; ei[e] = __float_as_int(qv.w);
      ed[e] = (c < m && d2 < rn2) ? d2 : INFINITY;
      if (c < m) cov_accumulate<false>(qv, px, py, pz, rn2, rs2, rs, a);
    }
  } else if (PCL && rn > 0.0f && m <= COV_LIST) {
    // (m <= COV_LIST < NBR_CAP: shot_pcl_long_kernel redoes this query's normal sums in PCL's arithmetic -- the float64 ones about
    // the query point would be overwritten: only the LRF columns are summed here; the count is marked below as before)
    for (int c = lane; c < m; c += 64) cov_accumulate<false>(s_nb[c], px, py, pz, rn2, rs2, rs, a);
  } else if (m <= COV_LIST) {
    for (int c = lane; c < m; c += 64) cov_accumulate(s_nb[c], px, py, pz, rn2, rs2, rs, a);
  } else {
    // more neighbours than the list holds (a support radius far above the cloud's resolution): the sums straight from the runs
    for (int fb = 0; fb < C; fb += 64)
      if (fb + lane < C) cov_accumulate(sp[cand(fb + lane)], px, py, pz, rn2, rs2, rs, a);
  }
  __syncthreads();                                 // the list is dead: its LDS becomes the reduction buffer
  float pcl_sum = 0.0f;                            // lane j < 9: the j-th float32 sum; mn: neighbours inside rn
  int mn = 0;
  if (pcl_list) pcl_float_sums<PCL_EMAX_SHORT>(lane, s_raw, rn2, ex, ey, ez, ed, ei, pcl_sum, mn);
  if (nbr_cnt && lane == 0) {
    nbr_cnt[qi] = m;
    if (qi == 0) reinterpret_cast<float*>(nbr_cnt)[-1] = fmaxf(rn, rs);      // radius of the lists (slot before the counts)
  }
  if (SHOT_DBG & 128) { if (lane < NSUM) sums[(int64_t)qi * NSUM + lane] = a[lane % 3]; return; }
  if (lane >= 32 && lane < 32 + NSUM) s_part[lane - 32][32] = 0.0;
  // (a ranked list has its normal sums from pcl_float_sums: columns 0..9 of the float64 partials are not needed then)
  if (pcl_list) {
#pragma unroll
    for (int c = 10; c < NSUM; ++c) {
      a[c] += upper_half(a[c]);
      if (lane < 32) s_part[c][lane] = a[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NSUM; ++c) {
      a[c] += upper_half(a[c]);                    // lane l < 32: its own partial + lane l + 32's
      if (lane < 32) s_part[c][lane] = a[c];
    }
  }
  __syncthreads();
  // column sums in a fixed order (run-to-run reproducible): three lanes per column add a third of the 32 partials each,
  // the first of them adds the three thirds.  All partials of a lane are requested first, then added in index order (as a
  // load-add loop every addition waits out one LDS round trip); entries past the end add an exact 0.0.
  {
    // (entries 0..10, 11..21, 22..32: the 33rd entry of a row is the pad slot, zeroed below, so every lane adds 11 entries)
    // five columns per row of 16 lanes (lane 15 of a row idles): the three thirds of a column meet through DPP row shifts
    const int in_row = lane & 15, c = min(5 * (lane >> 4) + in_row / 3, NSUM - 1), part = in_row - 3 * (in_row / 3);
    const int lo = part * 11;
    double v[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) v[i] = s_part[c][lo + i];
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 11; ++i) t += v[i];
    const double t1 = row_down<0x101>(t), t2 = row_down<0x102>(t);
    // (PCL, list too long for LDS: the float64 sums about the query point stand in, the count is marked with + 0.5)
    const double mark = (PCL && !pcl_list && rn > 0.0f && c == 9) ? 0.5 : 0.0;
    if (5 * (lane >> 4) + in_row / 3 < NSUM && in_row < 15 && part == 0 && !(pcl_list && c < 10))
      sums[(int64_t)qi * NSUM + c] = ((t + t1) + t2) + mark;
  }
  if (pcl_list) {
    if (lane < 9) sums[(int64_t)qi * NSUM + lane] = (double)pcl_sum;
    if (lane == 9) sums[(int64_t)qi * NSUM + 9] = (double)mn;
  }
}

// The queries shot_cov_kernel<true> marked (more than 128 neighbours): the same float32 sums from the workspace copy of the
// neighbour list, up to eight entries per lane.  Every wavefront looks at PCL_LONG_SCAN queries' marks at a time and redoes the
// marked ones, one after the other (a launch of one workgroup per query cost 60 us just to find that most had nothing to do; 64
// queries per wavefront left a cloud at voxel-grid density -- every query marked -- with four wavefronts per SIMD).
// One marked query with NENT entry slots per lane (lists of up to 64 NENT neighbours): gathered from the workspace list, ranked,
// summed.  A template so that a list of 251 neighbours costs four slots of ranking and addend writes, not the capacity's eight.
template <int NENT>
__device__ __forceinline__ void pcl_long_query(int lane, double* s_raw, int64_t qi, int m, float rn2, const float4* __restrict__ sp,
                                               float px, float py, float pz, const int32_t* __restrict__ my_list,
                                               double* __restrict__ sums) {
  float ex[NENT], ey[NENT], ez[NENT], ed[NENT];
  int ei[NENT];
#pragma unroll
  for (int e = 0; e < NENT; ++e) {
    const int c = lane + 64 * e;
    ed[e] = INFINITY; ex[e] = ey[e] = ez[e] = 0.0f; ei[e] = 0;
    if (c < m) {
      const float4 qv = sp[(unsigned)my_list[c]];
      const float d2 = sqdist3(px, py, pz, qv.x, qv.y, qv.z);
      if (d2 < rn2) { ex[e] = qv.x; ey[e] = qv.y; ez[e] = qv.z; ed[e] = d2; ei[e] = __float_as_int(qv.w); }
    }
  }
  float pcl_sum = 0.0f;
  int mn = 0;
  __syncthreads();
  pcl_float_sums<NENT>(lane, s_raw, rn2, ex, ey, ez, ed, ei, pcl_sum, mn);
  if (lane < 9) sums[qi * NSUM + lane] = (double)pcl_sum;
  if (lane == 9) sums[qi * NSUM + 9] = (double)mn;
}

__global__ __launch_bounds__(64) void shot_pcl_long_kernel(int64_t total, const float* __restrict__ pts, const int32_t* __restrict__ pt_off,
                                                           const float4* __restrict__ sorted_pts,
                                                           const int32_t* __restrict__ scene_of, float rn,
                                                           const int32_t* __restrict__ nbr_list,
                                                           const int32_t* __restrict__ nbr_cnt, double* __restrict__ sums) {
  __shared__ __attribute__((aligned(16))) double s_raw[NSUM * (32 + 1)];
  const int lane = threadIdx.x;
  const float rn2 = rn * rn;
  for (int64_t q0 = (int64_t)blockIdx.x * PCL_LONG_SCAN; q0 < total; q0 += (int64_t)gridDim.x * PCL_LONG_SCAN) {
    bool todo = false;
    if (lane < PCL_LONG_SCAN && q0 + lane < total) {
      const double s9 = sums[(q0 + lane) * NSUM + 9];
      todo = s9 != floor(s9) && nbr_cnt[q0 + lane] <= NBR_CAP;      // (longer than this kernel's capacity: the float64 sums stay)
    }
    unsigned long long mask = wave_ballot(todo);
    while (mask) {
      const int64_t qi = q0 + __builtin_ctzll(mask);
      mask &= mask - 1;
      const int m = nbr_cnt[qi];
      const float4* sp = sorted_pts + (int64_t)pt_off[scene_of[qi]];
      const float px = pts[3 * qi], py = pts[3 * qi + 1], pz = pts[3 * qi + 2];
      const int32_t* my_list = nbr_list + qi * NBR_CAP;
      static_assert(PCL_EMAX_LONG == 8, "the dispatch below covers 3 .. 8 entry slots");
      switch ((m + 63) >> 6) {                     // wave-uniform: m comes from a scalar load
        case 0: case 1: case 2: case 3: pcl_long_query<3>(lane, s_raw, qi, m, rn2, sp, px, py, pz, my_list, sums); break;
        case 4: pcl_long_query<4>(lane, s_raw, qi, m, rn2, sp, px, py, pz, my_list, sums); break;
        case 5: pcl_long_query<5>(lane, s_raw, qi, m, rn2, sp, px, py, pz, my_list, sums); break;
        case 6: pcl_long_query<6>(lane, s_raw, qi, m, rn2, sp, px, py, pz, my_list, sums); break;
        default: pcl_long_query<8>(lane, s_raw, qi, m, rn2, sp, px, py, pz, my_list, sums); break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// shot_eig: normal (pcl::NormalEstimation + flipNormalTowardsViewpoint(origin)) and the unsigned LRF axes
// ---------------------------------------------------------------------------------------------
struct LrfPre {
  double v1[3], v3[3];
  int32_t valid, nn;
};

// pcl::computeRoots2 / computeRoots / eigen33 (PCL 1.9.1 common/impl/eigen.hpp, Scalar = float), as oracle/shot_oracle.c restates
// them: the eigenvector of the smallest eigenvalue of a symmetric float 3x3 in closed form (trigonometric roots of the
// characteristic polynomial of the matrix scaled by its largest entry; the longest cross product of two rows of M - root I).
__device__ __forceinline__ void pcl_roots2(float b, float c, float roots[3]) {
  roots[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = __builtin_sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

__device__ void pcl_roots(const float m[3][3], float roots[3]) {
  const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
                   m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
  const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] -
                   m[1][2] * m[1][2];
  const float c2 = m[0][0] + m[1][1] + m[2][2];
  if (fabsf(c0) < 1.1920929e-07f) { pcl_roots2(c2, c1, roots); return; }
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = __builtin_sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = __builtin_sqrtf(-a_over_3);
  const float theta = atan2f(__builtin_sqrtf(-q), half_b) * s_inv3;
  const float cos_theta = cosf(theta), sin_theta = sinf(theta);
  roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (roots[0] >= roots[1]) { const float t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    const float t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { const float u = roots[0]; roots[0] = roots[1]; roots[1] = u; }
  }
  if (roots[0] <= 0.0f) pcl_roots2(c2, c1, roots);
}

__device__ void pcl_eigen33(const float cov[3][3], float vec[3]) {
  float scale = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) scale = fmaxf(scale, fabsf(cov[i][j]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float m[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i][j] = cov[i][j] / scale;
  float roots[3];
  pcl_roots(m, roots);
#pragma unroll
  for (int i = 0; i < 3; ++i) m[i][i] -= roots[0];
  float v[3][3], len[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ra = (k == 2) ? 1 : 0, rb = (k == 0) ? 1 : 2;          // row pairs (0,1), (0,2), (1,2)
    const float* a = m[ra];
    const float* b = m[rb];
    v[k][0] = a[1] * b[2] - a[2] * b[1];
    v[k][1] = a[2] * b[0] - a[0] * b[2];
    v[k][2] = a[0] * b[1] - a[1] * b[0];
    len[k] = v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2];
  }
  int best = 2;
  if (len[0] >= len[1] && len[0] >= len[2]) best = 0;
  else if (len[1] >= len[0] && len[1] >= len[2]) best = 1;
  const float bx = best == 0 ? v[0][0] : best == 1 ? v[1][0] : v[2][0];
  const float by = best == 0 ? v[0][1] : best == 1 ? v[1][1] : v[2][1];
  const float bz = best == 0 ? v[0][2] : best == 1 ? v[1][2] : v[2][2];
  const float inv = __builtin_sqrtf(best == 0 ? len[0] : best == 1 ? len[1] : len[2]);
  vec[0] = bx / inv; vec[1] = by / inv; vec[2] = bz / inv;
}

// pcl: the sums of a shot_cov_kernel<true> launch -- float32 sums of the raw coordinates in slots 0..8 (count in slot 9; a count
// with a fractional part marks a query that kept the float64 sums): covariance = E[x x^T] - E[x] E[x]^T in float32, closed-form
// eigenvector, float32 viewpoint flip (pcl::NormalEstimation + flipNormalTowardsViewpoint, PCL 1.9.1).
__global__ __launch_bounds__(256) void shot_eig_kernel(int64_t total, const float* __restrict__ pts,
                                                       const double* __restrict__ sums, float* __restrict__ normals,
                                                       LrfPre* __restrict__ pre, int pcl) {
  const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= total) return;
  const double* s = sums + qi * NSUM;
  if (normals) {
    float* o = normals + 3 * qi;
    const int cnt = (int)s[9];
    if (cnt < 3) {
      o[0] = o[1] = o[2] = NAN;
    } else if (pcl && s[9] == (double)cnt) {
      float accu[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) accu[c] = (float)s[c] / (float)cnt;
      float cov[3][3];
      cov[0][0] = accu[0] - accu[6] * accu[6];
      cov[0][1] = accu[1] - accu[6] * accu[7];
      cov[0][2] = accu[2] - accu[6] * accu[8];
      cov[1][1] = accu[3] - accu[7] * accu[7];
      cov[1][2] = accu[4] - accu[7] * accu[8];
      cov[2][2] = accu[5] - accu[8] * accu[8];
      cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
      float v[3];
      pcl_eigen33(cov, v);
      const float vx = 0.0f - pts[3 * qi], vy = 0.0f - pts[3 * qi + 1], vz = 0.0f - pts[3 * qi + 2];
      const float ct = vx * v[0] + vy * v[1] + vz * v[2];
      if (ct < 0.0f) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
      const double inv = 1.0 / (double)cnt;
      const double mx = s[6] * inv, my = s[7] * inv, mz = s[8] * inv;
      const double cov[6] = {s[0] * inv - mx * mx, s[1] * inv - mx * my, s[2] * inv - mx * mz,
                             s[3] * inv - my * my, s[4] * inv - my * mz, s[5] * inv - mz * mz};
      const Eig3 e = jacobi3(cov);
      double nx = e.v[0][0], ny = e.v[1][0], nz = e.v[2][0];
      const double px = (double)pts[3 * qi], py = (double)pts[3 * qi + 1], pz = (double)pts[3 * qi + 2];
      const double ct = (0.0 - px) * nx + (0.0 - py) * ny + (0.0 - pz) * nz;
      if (ct < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
      o[0] = (float)nx; o[1] = (float)ny; o[2] = (float)nz;
    }
  }
  if (pre) {
    LrfPre p;
    p.valid = (int)s[17];
    p.nn = (int)s[18];
    bool ok = p.valid >= 5;
    if (ok) {
      double cov[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) cov[c] = s[10 + c] / s[16];
      const Eig3 e = jacobi3(cov);
      ok = isfinite(e.w[0]) && isfinite(e.w[1]) && isfinite(e.w[2]);
      p.v1[0] = e.v[0][2]; p.v1[1] = e.v[1][2]; p.v1[2] = e.v[2][2];
      p.v3[0] = e.v[0][0]; p.v3[1] = e.v[1][0]; p.v3[2] = e.v[2][0];
    }
    if (!ok) { p.valid = -1; p.v1[0] = p.v1[1] = p.v1[2] = p.v3[0] = p.v3[1] = p.v3[2] = NAN; }
    pre[qi] = p;
  }
}

// normals permuted into the cell-sorted order, so that the histogram pass reads a neighbour's position and normal
// with the same index (no dependent index load in its inner loop)
__global__ __launch_bounds__(256) void shot_sort_normals_kernel(int64_t total, const float* __restrict__ nrm,
                                                                const int32_t* __restrict__ pt_off,
                                                                const int32_t* __restrict__ scene_of,
                                                                const int32_t* __restrict__ sorted_idx,
                                                                float4* __restrict__ sorted_nrm) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const int64_t src = (int64_t)pt_off[scene_of[g]] + sorted_idx[g];
  sorted_nrm[g] = make_float4(nrm[3 * src], nrm[3 * src + 1], nrm[3 * src + 2], 0.0f);
}

// ---------------------------------------------------------------------------------------------
// shot_hist: sign disambiguation + interpolated histogram
// ---------------------------------------------------------------------------------------------
// normalised CIELab of one point's colour the way pcl::SHOTColorEstimation::RGB2CIELAB computes it from PCL's two lookup
// tables (sRGB gamma at 256 levels; cube root of XYZ quantised to 1/4000 with the exponent 0.3333f), evaluated in place
// of the tables; components stored as uint8 = float * 255.f truncated like src_shot/shot.cpp:114-116.
__device__ __forceinline__ float lab_srgb(float c) {
  const float v = c * 255.f;
  int i = (v == v) ? (int)v : 0;
  i = min(max(i, 0), 255);
  const float f = (float)i / 255.0f;
  return (f > 0.04045f) ? powf((f + 0.055f) / 1.055f, 2.4f) : f / 12.92f;
}
__device__ __forceinline__ float lab_xyz(float v) {
  const int i = min((int)(v * 4000), 3999);
  const float f = (float)i / 4000.0f;
  return (f > 0.008856f) ? powf(f, 0.3333f) : (float)((7.787 * (double)f) + (16.0 / 116.0));
}
__device__ __forceinline__ float4 lab_norm(const float* __restrict__ rgb) {
  const float fr = lab_srgb(rgb[0]), fg = lab_srgb(rgb[1]), fb = lab_srgb(rgb[2]);
  const float x = (fr * 0.412453f + fg * 0.357580f) + fb * 0.180423f;
  const float y = (fr * 0.212671f + fg * 0.715160f) + fb * 0.072169f;
  const float z = (fr * 0.019334f + fg * 0.119193f) + fb * 0.950227f;
  const float vx = lab_xyz(x / 0.95047f), vy = lab_xyz(y), vz = lab_xyz(z / 1.08883f);
  const float L = fminf(116.0f * vy - 16.0f, 100.0f);
  const float A = fminf(fmaxf(500.0f * (vx - vy), -120.0f), 120.0f);
  const float Bv = fminf(fmaxf(200.0f * (vy - vz), -120.0f), 120.0f);
  return make_float4(L / 100.0f, A / 120.0f, Bv / 120.0f, 0.0f);
}

// Lab of every point, in the caller's order and in the cell-sorted order the histogram kernel walks
__global__ __launch_bounds__(256) void shot_lab_kernel(int64_t total, const float* __restrict__ colors,
                                                       const int32_t* __restrict__ pt_off,
                                                       const int32_t* __restrict__ scene_of,
                                                       const int32_t* __restrict__ sorted_idx,
                                                       float4* __restrict__ lab, float4* __restrict__ sorted_lab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  lab[i] = lab_norm(colors + 3 * i);
  // sorted position i of its scene holds original point sorted_idx[i] (scene-local)
  const int b = scene_of[i];
  sorted_lab[i] = lab_norm(colors + 3 * ((int64_t)pt_off[b] + sorted_idx[i]));
}

template <bool COLOR>
__global__ __launch_bounds__(64) void shot_hist_kernel(int B, const float* __restrict__ pts,
                                                       const int32_t* __restrict__ pt_off,
                                                       const CellHdr* __restrict__ hdrs,
                                                       const int32_t* __restrict__ cell_start,
                                                       const int32_t* __restrict__ sorted_idx,
                                                       const float4* __restrict__ sorted_pts,
                                                       const int32_t* __restrict__ scene_of,
                                                       const float4* __restrict__ sorted_nrm,
                                                       const LrfPre* __restrict__ pre, float radius, int nan_to_zero,
                                                       const int32_t* __restrict__ nbr_list,
                                                       const int32_t* __restrict__ nbr_cnt,
                                                       float* __restrict__ out_shot,
                                                       float* __restrict__ out_rf,
                                                       const float4* __restrict__ lab = nullptr,
                                                       const float4* __restrict__ sorted_lab = nullptr) {
  constexpr int LEN = COLOR ? SHOT_COLOR_LEN : SHOT_LEN;
  constexpr int HWORDS = COLOR ? SHOT_COLOR_LEN + 1 : SH_COPIES * SH_STRIDE;
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[HWORDS];
  uint32_t* hist = COLOR ? s_hist : s_hist + (threadIdx.x & (SH_COPIES - 1)) * SH_STRIDE;
  __shared__ __attribute__((aligned(16))) int s_list[SH_LCAP];       // positions in the cell-sorted order (its upper part doubles as 64-bit key storage)
  __shared__ float s_rf[9];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const int b = scene_of[qi];
  const int p0 = pt_off[b];
  const CellHdr h = hdrs[b];
  const int32_t* cs = cell_start + (int64_t)b * (CELL_CAP + 1);
  const float4* sp = sorted_pts + (int64_t)p0;       // (x, y, z, original index)
  const float4* sn = sorted_nrm + (int64_t)p0;       // cell-sorted order, like sp
  const float px = pts[3 * (int64_t)qi], py = pts[3 * (int64_t)qi + 1], pz = pts[3 * (int64_t)qi + 2];
  const float r2 = radius * radius;
  float* o = out_shot + (int64_t)LEN * qi;
  const LrfPre lp = pre[qi];
  const int valid = lp.valid;
  bool ok = valid >= 5;
  if (!ok || lp.nn < 5) {
    for (int c = lane; c < LEN; c += 64) o[c] = nan_to_zero ? 0.0f : NAN;
    if (out_rf && lane < 9) out_rf[9 * (int64_t)qi + lane] = NAN;
    if (!ok) return;
    // LRF exists but too few points for the descriptor: frame is still reported
  }
  // neighbour list (positions in cell order), all points with d2 < r2 incl. self: taken from shot_cov's compacted list
  // when that was built with this radius (copied) or a larger one (filtered), otherwise rebuilt from the 9 runs
  Runs runs;
#pragma unroll
  for (int k = 0; k < 9; ++k) { runs.beg[k] = 0; runs.end[k] = 0; }
  int m = 0;
  const int shared_cnt = nbr_cnt ? nbr_cnt[qi] : -1;
  const float list_r = nbr_cnt ? reinterpret_cast<const float*>(nbr_cnt)[-1] : -1.0f;
  if (shared_cnt >= 0 && shared_cnt <= NBR_CAP && list_r * list_r >= r2) {
    const int32_t* nl = nbr_list + (int64_t)qi * NBR_CAP;
    if (list_r * list_r == r2) {
      for (int c = lane; c < shared_cnt; c += 64) s_list[c] = nl[c];
      m = shared_cnt;
    } else {
      for (int base = 0; base < shared_cnt; base += 64) {
        const int c = base + lane;
        int j = 0;
        bool in = false;
        if (c < shared_cnt) { j = nl[c]; const float4 qv = sp[j]; in = sqdist3(px, py, pz, qv.x, qv.y, qv.z) < r2; }
        const unsigned long long mask = wave_ballot(in);
        if (in) s_list[m + lanes_below(mask)] = j;
        m += __popcll(mask);
      }
    }
  } else {
    query_runs(h, cs, px, py, pz, runs);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      for (int jb = runs.beg[k]; jb < runs.end[k]; jb += 64) {
        const int j = jb + lane;
        bool in = false;
        if (j < runs.end[k]) { const float4 qv = sp[j]; in = sqdist3(px, py, pz, qv.x, qv.y, qv.z) < r2; }
        const unsigned long long mask = wave_ballot(in);
        if (in) {
          const int pos = m + lanes_below(mask);
          if (pos < SH_LCAP) s_list[pos] = j;
        }
        m += __popcll(mask);
      }
    }
  }
  __syncthreads();
  const bool listed = (m <= SH_LCAP);
  // the usual case (m <= 128): each lane keeps its two neighbours' positions and normals in registers for all passes, so the
  // sign pass, the histogram pass and (shot_accumulate) the normals are one round of gathers instead of three dependent ones
  const bool small = !COLOR && m <= 128;
  float4 cp0 = make_float4(px, py, pz, 0.0f), cp1 = cp0, cn0 = cp0, cn1 = cp0;
  if (small) {
    if (lane < m) { const int j = s_list[lane]; cp0 = sp[j]; cn0 = sn[j]; }
    if (lane + 64 < m) { const int j = s_list[lane + 64]; cp1 = sp[j]; cn1 = sn[j]; }
  }
  // candidates of the later sweeps: the list, or (overflow) every point of the 9 runs again
  double v1[3] = {lp.v1[0], lp.v1[1], lp.v1[2]}, v3[3] = {lp.v3[0], lp.v3[1], lp.v3[2]};
  int plus1 = 0, plus3 = 0, c1 = 0, c3 = 0;
  const int med = valid / 2;
  // pass over neighbours: MODE 0 = sign counts, 1 = tie-break (median rule), 2 = histogram
#define FOR_EACH_NEIGHBOUR(...)                                                             \
  if (listed) {                                                                              \
    for (int base = 0; base < m; base += 64) {                                               \
      const int c = base + lane;                                                             \
      const bool act = c < m;                                                                \
      const int j = act ? s_list[c] : 0;                                                     \
      __VA_ARGS__                                                                                 \
    }                                                                                        \
  } else {                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 9; ++k)                                            \
      for (int jb = runs.beg[k]; jb < runs.end[k]; jb += 64) {                               \
        const int j = min(jb + lane, runs.end[k] - 1);                                       \
        const float4 qv_ = sp[j];                                                            \
        const bool act = (jb + lane < runs.end[k]) && (sqdist3(px, py, pz, qv_.x, qv_.y, qv_.z) < r2); \
        __VA_ARGS__                                                                               \
      }                                                                                      \
  }

  // Sign of the projections on v1 / v3 (the reference's float64 dot product): a float32 dot decides it whenever it is further from
  // zero than 2e-6 (|x| + |y| + |z|) -- eight times the worst float32 error for a unit vector -- and only the lanes it leaves
  // open evaluate the float64 expression (rare: the wavefront skips that arm otherwise).
  const float v1f[3] = {(float)v1[0], (float)v1[1], (float)v1[2]}, v3f[3] = {(float)v3[0], (float)v3[1], (float)v3[2]};
  auto signs_of = [&](const float4 qv, bool act, bool& p1, bool& p3) {
    p1 = false; p3 = false;
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    if (act && !(qx == px && qy == py && qz == pz)) {
      const float xf = qx - px, yf = qy - py, zf = qz - pz;
      const float d1 = fmaf(zf, v1f[2], fmaf(yf, v1f[1], xf * v1f[0]));
      const float d3 = fmaf(zf, v3f[2], fmaf(yf, v3f[1], xf * v3f[0]));
      const float bnd = 2e-6f * ((fabsf(xf) + fabsf(yf)) + fabsf(zf));
      p1 = d1 > 0.0f;
      p3 = d3 > 0.0f;
      if ((SHOT_DBG & 32) || !(fabsf(d1) > bnd) || !(fabsf(d3) > bnd)) {
        const double x = (double)xf, y = (double)yf, z = (double)zf;
        p1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) >= 0.0;
        p3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) >= 0.0;
      }
    }
  };
  if (SHOT_DBG & 1) {
  } else if (small) {            // the two register-held neighbours, one after the other (no selection between them inside a loop)
    bool p1, p3;
    signs_of(cp0, lane < m, p1, p3);
    plus1 += __popcll(wave_ballot(p1));
    plus3 += __popcll(wave_ballot(p3));
    if (m > 64) {
      signs_of(cp1, lane + 64 < m, p1, p3);
      plus1 += __popcll(wave_ballot(p1));
      plus3 += __popcll(wave_ballot(p3));
    }
  } else {
  FOR_EACH_NEIGHBOUR({
    bool p1, p3;
    signs_of(sp[j], act, p1, p3);
    plus1 += __popcll(wave_ballot(p1));
    plus3 += __popcll(wave_ballot(p3));
  })
  }
  plus1 = 2 * plus1 - valid;
  plus3 = 2 * plus3 - valid;
  if ((plus1 == 0 || plus3 == 0) && small && !(SHOT_DBG & 16)) {
    // tie (one query in six: an even count splits evenly about as often as a fair coin does) with the neighbours in registers:
    // a neighbour's rank in the (distance, original index) order is the number of smaller 64-bit keys (distance bits : index),
    // counted by wave_rank (64 distance buckets; its tables live in the unused upper part of the list buffer)
    static_assert(SH_LCAP >= 1024, "wave_rank's 768 words live in list entries 256..1023");
    const float bscale = 64.0f * __builtin_amdgcn_rcpf(r2);
    auto key_of = [&](const float4 q, bool in, int& bucket) {
      const bool self = (q.x == px && q.y == py && q.z == pz);
      const float d2 = sqdist3(px, py, pz, q.x, q.y, q.z);
      bucket = (!in || self) ? -1 : min(63, (int)(d2 * bscale));
      return (!in || self) ? ~0ull : dist_index_key(d2, __float_as_int(q.w));
    };
    unsigned long long key[2];
    int bkt[2], rk[2];
    key[0] = key_of(cp0, lane < m, bkt[0]);
    key[1] = key_of(cp1, lane + 64 < m, bkt[1]);
    wave_rank<2>(lane, reinterpret_cast<uint32_t*>(s_list + 256), key, bkt, rk);
    const unsigned long long k0 = key[0], k1 = key[1];
    const int rk0 = rk[0], rk1 = rk[1];
    auto decide = [&](const float4 q, unsigned long long key, int rank, bool& h1, bool& h3) {
      h1 = false; h3 = false;
      if (key != ~0ull && rank >= med - 2 && rank <= med + 2) {
        const double x = (double)(q.x - px), y = (double)(q.y - py), z = (double)(q.z - pz);
        h1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) > 0.0;
        h3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) > 0.0;
      }
    };
    bool h1a, h3a, h1b, h3b;
    decide(cp0, k0, rk0, h1a, h3a);
    decide(cp1, k1, rk1, h1b, h3b);
    c1 = __popcll(wave_ballot(h1a)) + __popcll(wave_ballot(h1b));
    c3 = __popcll(wave_ballot(h3a)) + __popcll(wave_ballot(h3b));
    if (plus1 == 0) plus1 = (c1 < 3) ? -1 : 1;
    if (plus3 == 0) plus3 = (c3 < 3) ? -1 : 1;
  } else if ((plus1 == 0 || plus3 == 0) && !(SHOT_DBG & 512)) {
    // tie: the 5 neighbours around the median of the (distance, original index)-sorted valid list decide.  Only those five ranks
    // matter: a first sweep counts the neighbours per distance bucket (64 buckets; a prefix sum gives every bucket's rank range),
    // a second sweep collects the keys of the buckets that overlap ranks med - 2 .. med + 2 -- a handful -- and these are ranked
    // among themselves.  (Ranking every neighbour against every other was 0.46 of shot_hist's 0.94 ms on clouds at the density
    // of a 2 mm voxel grid, ~250 neighbours.)  More than 64 keys in those buckets (many equal distances): the exhaustive count below.
    static_assert(HWORDS >= 64 + 128 + 64, "the tie-break tables live in the histogram buffer, which is still unused here");
    bool ranked = false;
    {
      uint32_t* s_cnt = s_hist;                                                             // [64] neighbours per bucket
      unsigned long long* s_ck = reinterpret_cast<unsigned long long*>(s_hist + 64);        // [64] collected keys
      int* s_cj = reinterpret_cast<int*>(s_hist + 64 + 128);                                // [64] their list positions
      const float bscale = 64.0f * __builtin_amdgcn_rcpf(r2);
      s_cnt[lane] = 0u;
      __syncthreads();
      FOR_EACH_NEIGHBOUR({
        if (act) {
          const float4 qv = sp[j];
          if (!(qv.x == px && qv.y == py && qv.z == pz))
            atomicAdd(&s_cnt[min(63, (int)(sqdist3(px, py, pz, qv.x, qv.y, qv.z) * bscale))], 1u);
        }
      })
      __syncthreads();
      const int cnt = (int)s_cnt[lane];
      const int incl = (int)wave_inclusive_scan_u32((uint32_t)cnt), excl = incl - cnt;
      const unsigned long long overlap = wave_ballot(cnt > 0 && excl <= med + 2 && incl > med - 2);
      int nc = 0, base = 0, b_lo = 64, b_hi = -1;
      if (overlap) {
        b_lo = __builtin_ctzll(overlap);
        b_hi = 63 - __builtin_clzll(overlap);
        base = __shfl(excl, b_lo);
        FOR_EACH_NEIGHBOUR({
          bool cand = false;
          unsigned long long key = 0ull;
          if (act) {
            const float4 qv = sp[j];
            if (!(qv.x == px && qv.y == py && qv.z == pz)) {
              const float d2 = sqdist3(px, py, pz, qv.x, qv.y, qv.z);
              const int bk = min(63, (int)(d2 * bscale));
              cand = bk >= b_lo && bk <= b_hi;
              key = dist_index_key(d2, __float_as_int(qv.w));
            }
          }
          const unsigned long long cmask = wave_ballot(cand);
          if (cand) {
            const int pos = nc + lanes_below(cmask);
            if (pos < 64) { s_ck[pos] = key; s_cj[pos] = j; }
          }
          nc += __popcll(cmask);
        })
      }
      __syncthreads();
      if (nc <= 64) {
        ranked = true;
        bool h1 = false, h3 = false;
        if (lane < nc) {
          const unsigned long long key = s_ck[lane];
          int rank = base;
          for (int c2 = 0; c2 < nc; ++c2) rank += (s_ck[c2] < key) ? 1 : 0;
          if (rank >= med - 2 && rank <= med + 2) {
            const float4 qv = sp[s_cj[lane]];
            const double x = (double)(qv.x - px), y = (double)(qv.y - py), z = (double)(qv.z - pz);
            h1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) > 0.0;
            h3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) > 0.0;
          }
        }
        c1 = __popcll(wave_ballot(h1));
        c3 = __popcll(wave_ballot(h3));
      }
      __syncthreads();
    }
    if (!ranked) {
    FOR_EACH_NEIGHBOUR({
      bool h1 = false, h3 = false;
      if (act) {
        const float4 qv = sp[j];
        const float qx = qv.x, qy = qv.y, qz = qv.z;
        if (!(qx == px && qy == py && qz == pz)) {
          const float d2 = sqdist3(px, py, pz, qx, qy, qz);
          const int oj = __float_as_int(qv.w);
          int rank = 0;
          if (listed) {
            for (int c2 = 0; c2 < m; ++c2) {
              const int j2 = s_list[c2];
              const float4 uv = sp[j2];
              const float ux = uv.x, uy = uv.y, uz = uv.z;
              const float e2 = sqdist3(px, py, pz, ux, uy, uz);
              const bool nb2 = !(ux == px && uy == py && uz == pz);
              rank += (nb2 && (e2 < d2 || (e2 == d2 && __float_as_int(uv.w) < oj))) ? 1 : 0;
            }
          } else {
            for (int k2 = 0; k2 < 9; ++k2)
              for (int j2 = runs.beg[k2]; j2 < runs.end[k2]; ++j2) {
                const float4 uv = sp[j2];
                const float ux = uv.x, uy = uv.y, uz = uv.z;
                const float e2 = sqdist3(px, py, pz, ux, uy, uz);
                const bool nb2 = (e2 < r2) && !(ux == px && uy == py && uz == pz);
                rank += (nb2 && (e2 < d2 || (e2 == d2 && __float_as_int(uv.w) < oj))) ? 1 : 0;
              }
          }
          if (rank >= med - 2 && rank <= med + 2) {
            const double x = (double)(qx - px), y = (double)(qy - py), z = (double)(qz - pz);
            h1 = ((x * v1[0] + y * v1[1]) + z * v1[2]) > 0.0;
            h3 = ((x * v3[0] + y * v3[1]) + z * v3[2]) > 0.0;
          }
        }
      }
      c1 += __popcll(wave_ballot(h1));
      c3 += __popcll(wave_ballot(h3));
    })
    }
    if (plus1 == 0) plus1 = (c1 < 3) ? -1 : 1;
    if (plus3 == 0) plus3 = (c3 < 3) ? -1 : 1;
  }
  if (plus1 < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
  if (plus3 < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
  float rf[9];
  rf[0] = (float)v1[0]; rf[1] = (float)v1[1]; rf[2] = (float)v1[2];
  rf[6] = (float)v3[0]; rf[7] = (float)v3[1]; rf[8] = (float)v3[2];
  rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
  rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
  rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  if (out_rf && lane < 9) {
    float v = rf[0];
#pragma unroll
    for (int c = 1; c < 9; ++c) v = (lane == c) ? rf[c] : v;
    out_rf[9 * (int64_t)qi + lane] = v;
  }
  if (lp.nn < 5) return;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 9; ++c) s_rf[c] = rf[c];
  }
  for (int c = lane; c < HWORDS; c += 64) s_hist[c] = 0u;
  float4 labq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (COLOR) labq = lab[qi];
  // colour bin coordinate of a neighbour (SHOTColorEstimation::computePointSHOT): L1-type Lab distance in [0, 1] x 30 bins
  auto color_bd = [&](const float4 lq) {
    double cd = ((double)fabsf(labq.x - lq.x) + (((double)fabsf(labq.y - lq.y) + (double)fabsf(labq.z - lq.z)) / 2)) / 3;
    cd = fmin(fmax(cd, 0.0), 1.0);
    return (float)(cd * NR_COLOR_BINS);
  };
  // Bins are summed in fixed point: a bin receives at most 4 per neighbour, so 2^(29 - ceil(log2(m + 1))) units per 1.0
  // cannot overflow 32 bits.  Integer LDS atomics run at full rate (float ones were measured ~50 cycles per
  // wavefront instruction), the sum does not depend on the order of the neighbours, and one unit (<= 2^-22 for the
  // usual m < 128) is below the float rounding of the bins it replaces.
  const float fx_scale = (float)(1u << (29 - (32 - __clz(m))));
  __syncthreads();
  if (SHOT_DBG & 2) {
  } else if (small) {
    if (lane < m)
      shot_accumulate<COLOR>(px, py, pz, cp0.x, cp0.y, cp0.z, sqdist3(px, py, pz, cp0.x, cp0.y, cp0.z), cn0.x, cn0.y, cn0.z, s_rf,
                             radius, hist, fx_scale);
    if (m > 64 && lane + 64 < m)
      shot_accumulate<COLOR>(px, py, pz, cp1.x, cp1.y, cp1.z, sqdist3(px, py, pz, cp1.x, cp1.y, cp1.z), cn1.x, cn1.y, cn1.z, s_rf,
                             radius, hist, fx_scale);
  } else if (listed) {
    // the next 64 neighbours' positions and normals are requested before the current 64 are accumulated
    bool act = lane < m;
    float4 qp, qn, ql = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    {
      const int j = act ? s_list[lane] : 0;
      qp = sp[j]; qn = sn[j];
      if (COLOR) ql = sorted_lab[(int64_t)p0 + j];
    }
    for (int base = 0; base < m; base += 64) {
      const float qx = qp.x, qy = qp.y, qz = qp.z, nx = qn.x, ny = qn.y, nz = qn.z;
      const float4 lq = ql;
      const bool cur = act;
      const int cn = base + 64 + lane;
      act = cn < m;
      if (base + 64 < m) {
        const int j = act ? s_list[cn] : 0;
        qp = sp[j]; qn = sn[j];
        if (COLOR) ql = sorted_lab[(int64_t)p0 + j];
      }
      if (cur)
        shot_accumulate<COLOR>(px, py, pz, qx, qy, qz, sqdist3(px, py, pz, qx, qy, qz), nx, ny, nz, s_rf, radius, hist,
                               fx_scale, COLOR ? color_bd(lq) : 0.0f);
    }
  } else {
    FOR_EACH_NEIGHBOUR({
      if (act) {
        const float4 qv = sp[j], nv = sn[j];
        shot_accumulate<COLOR>(px, py, pz, qv.x, qv.y, qv.z, sqdist3(px, py, pz, qv.x, qv.y, qv.z), nv.x, nv.y, nv.z, s_rf,
                               radius, hist, fx_scale, COLOR ? color_bd(sorted_lab[(int64_t)p0 + j]) : 0.0f);
      }
    })
  }
#undef FOR_EACH_NEIGHBOUR
  __syncthreads();
  if (SHOT_DBG & 4) { for (int c = lane; c < LEN; c += 64) __builtin_nontemporal_store(__uint_as_float(s_hist[c]), &o[c]); return; }
  double acc = 0.0;
  const float fx_inv = 1.0f / fx_scale;                // a power of two, like fx_scale: the product below is exact
  for (int c = lane; c < LEN; c += 64) {
    uint32_t u = s_hist[c];
#pragma unroll
    for (int k = 1; k < (COLOR ? 1 : SH_COPIES); ++k) u += s_hist[k * SH_STRIDE + c];
    const float v = (float)u * fx_inv;
    s_hist[c] = __float_as_uint(v);                    // each bin is folded by the lane that normalises it below
    acc += (double)v * (double)v;
  }
  acc = sqrt(wave_sum(acc));
  const float facc = (float)acc;
  // v / facc, correctly rounded, as the float division's own expansion computes it (reciprocal refined once, quotient refined
  // twice with exact remainders) without its range scaling: bins are 0 or >= 2^-29 and facc is their norm, far from the ranges
  // that need it; 0 / 0 still gives NaN.  The reciprocal is shared by the 352 quotients.
  const float r0 = __builtin_amdgcn_rcpf(facc);
  const float r1 = fmaf(fmaf(-facc, r0, 1.0f), r0, r0);
  for (int c = lane; c < LEN; c += 64) {
    const float x = __uint_as_float(s_hist[c]);
    const float q0 = x * r1;
    const float q1 = fmaf(fmaf(-facc, q0, x), r1, q0);
    float v = fmaf(fmaf(-facc, q1, x), r1, q1);
    if (SHOT_DBG & 8) v = x / facc;
    if (nan_to_zero && !(v == v)) v = 0.0f;            // an all-zero histogram normalises to 0/0
    __builtin_nontemporal_store(v, &o[c]);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }

struct ShotWs {
  CellHdr* hdr; int32_t* cell_start; int32_t* sorted_idx; float4* sorted_pts; double* sums; LrfPre* pre;
  int32_t* scene_of; float4* sorted_nrm; int32_t* nbr_cnt; int32_t* nbr_list;
};

static ShotWs carve(void* ws, int B, int64_t n) {
  char* p = (char*)ws;
  ShotWs w;
  w.hdr = (CellHdr*)p; p += up256((int64_t)B * sizeof(CellHdr));
  w.cell_start = (int32_t*)p; p += up256((int64_t)B * (CELL_CAP + 1) * 4);
  w.sorted_idx = (int32_t*)p; p += up256(n * 4);
  w.sorted_pts = (float4*)p; p += up256(n * 16);
  w.sums = (double*)p; p += up256(n * NSUM * 8);
  w.pre = (LrfPre*)p; p += up256(n * (int64_t)sizeof(LrfPre));
  w.scene_of = (int32_t*)p; p += up256(n * 4);
  w.sorted_nrm = (float4*)p; p += up256(n * 16);
  w.nbr_cnt = (int32_t*)(p + 256); p += up256(n * 4 + 256);        // one slot before the counts: the lists' radius
  w.nbr_list = (int32_t*)p;
  return w;
}

extern "C" int64_t cppf_shot352_workspace_bytes(int B, int64_t total_points) {
  if (B <= 0 || total_points <= 0) return 0;
  return up256((int64_t)B * sizeof(CellHdr)) + up256((int64_t)B * (CELL_CAP + 1) * 4) + up256(total_points * 4) +
         up256(total_points * 16) + up256(total_points * NSUM * 8) + up256(total_points * (int64_t)sizeof(LrfPre)) +
         up256(total_points * 4) + up256(total_points * 16) + up256(total_points * 4 + 256) +
         up256(total_points * (int64_t)NBR_CAP * 4);
}

// CPPF_SHOT_F64_NORMALS (flags bit 0 of the entry points): the round-1..3 arithmetic for the normals -- float64 covariance about
// the query point, Jacobi -- instead of pcl::NormalEstimation's (single-pass float32 sums of the raw coordinates in
// (distance, index) order, closed-form eigen33, float32 viewpoint flip), which is the default since round 4.
template <class... Args>
static void launch_cov(bool pcl, dim3 grid, hipStream_t st, Args... args) {
  if (pcl) hipLaunchKernelGGL(shot_cov_kernel<true>, grid, dim3(64), 0, st, args...);
  else hipLaunchKernelGGL(shot_cov_kernel<false>, grid, dim3(64), 0, st, args...);
}

static int shot_run(int B, const float* pts, const int32_t* pt_off, int64_t n, float normal_r, float shot_r,
                    const float* normals_in, float* out_normal, float* out_shot, float* out_rf, void* workspace,
                    int64_t workspace_bytes, hipStream_t st, int flags) {
  const bool pcl = !(flags & CPPF_SHOT_F64_NORMALS);
  CPPF_CHECK_ARG(workspace && workspace_bytes >= cppf_shot352_workspace_bytes(B, n));
  const ShotWs w = carve(workspace, B, n);
  const bool want_n = out_normal != nullptr, want_s = out_shot != nullptr;
  const float rn = want_n ? normal_r : 0.0f, rs = want_s ? shot_r : 0.0f;
  hipLaunchKernelGGL(shot_cells_kernel, dim3(B), dim3(1024), 0, st, pts, pt_off, fmaxf(rn, rs), w.hdr, w.cell_start,
                     w.sorted_idx, w.sorted_pts, w.scene_of);
  CPPF_LAUNCH_CHECK();
  launch_cov(pcl, dim3((unsigned)n), st, B, pts, pt_off, (const CellHdr*)w.hdr, (const int32_t*)w.cell_start,
             (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, rn, rs, w.sums,
             (want_s || pcl) ? w.nbr_list : (int32_t*)nullptr,      // (PCL normals read long lists back from it)
             (want_s || pcl) ? w.nbr_cnt : (int32_t*)nullptr);
  CPPF_LAUNCH_CHECK();
  if (pcl && want_n) {
    hipLaunchKernelGGL(shot_pcl_long_kernel, dim3((unsigned)((n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN < 16384 ? (n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN : 16384)), dim3(64), 0, st, n, pts, pt_off,
                       (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, rn, (const int32_t*)w.nbr_list,
                       (const int32_t*)w.nbr_cnt, w.sums);
    CPPF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(shot_eig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pts, w.sums,
                     out_normal, want_s ? w.pre : (LrfPre*)nullptr, (int)pcl);
  CPPF_LAUNCH_CHECK();
  if (want_s) {
    hipLaunchKernelGGL(shot_sort_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                       normals_in ? normals_in : out_normal, pt_off, w.scene_of, w.sorted_idx, w.sorted_nrm);
    CPPF_LAUNCH_CHECK();
    hipLaunchKernelGGL(shot_hist_kernel<false>, dim3((unsigned)n), dim3(64), 0, st, B, pts, pt_off, w.hdr, w.cell_start,
                       w.sorted_idx, w.sorted_pts, w.scene_of, w.sorted_nrm, w.pre, shot_r, 0, w.nbr_list, w.nbr_cnt,
                       out_shot, out_rf);
    CPPF_LAUNCH_CHECK();
  }
  return CPPF_OK;
}

extern "C" int cppf_estimate_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                                     float normal_r, float* out_normal, void* workspace, int64_t workspace_bytes,
                                     int flags, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out_normal && normal_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  return shot_run(B, pts, pt_off, total_points, normal_r, 0.0f, nullptr, out_normal, nullptr, nullptr, workspace,
                  workspace_bytes, (hipStream_t)stream, flags);
}

extern "C" int cppf_shot352(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r,
                            float shot_r, float* out_shot, float* out_normal, float* out_rf, void* workspace,
                            int64_t workspace_bytes, int flags, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out_shot && out_normal && normal_r > 0.0f && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  return shot_run(B, pts, pt_off, total_points, normal_r, shot_r, nullptr, out_normal, out_shot, out_rf, workspace,
                  workspace_bytes, (hipStream_t)stream, flags);
}

extern "C" int cppf_shot352_from_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                                         const float* normals, float shot_r, float* out_shot, float* out_rf,
                                         void* workspace, int64_t workspace_bytes, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && normals && out_shot && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  return shot_run(B, pts, pt_off, total_points, 0.0f, shot_r, normals, nullptr, out_shot, out_rf, workspace,
                  workspace_bytes, (hipStream_t)stream, 0);
}

// Two-call form of cppf_shot352 sharing one workspace: prepare = cells + covariances + eigen-solves (normals out,
// local-frame axes kept in the workspace), describe = histogram kernel only.  describe must follow a prepare on
// the same inputs, same stream, same workspace.
extern "C" int cppf_shot_prepare(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r,
                                 float shot_r, float* out_normal, void* workspace, int64_t workspace_bytes,
                                 int flags, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && out_normal && normal_r > 0.0f && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  CPPF_CHECK_ARG(workspace && workspace_bytes >= cppf_shot352_workspace_bytes(B, total_points));
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = total_points;
  const ShotWs w = carve(workspace, B, n);
  hipLaunchKernelGGL(shot_cells_kernel, dim3(B), dim3(1024), 0, st, pts, pt_off, fmaxf(normal_r, shot_r), w.hdr,
                     w.cell_start, w.sorted_idx, w.sorted_pts, w.scene_of);
  CPPF_LAUNCH_CHECK();
  const bool pcl = !(flags & CPPF_SHOT_F64_NORMALS);
  launch_cov(pcl, dim3((unsigned)n), st, B, pts, pt_off, (const CellHdr*)w.hdr, (const int32_t*)w.cell_start,
             (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, normal_r, shot_r, w.sums, w.nbr_list, w.nbr_cnt);
  CPPF_LAUNCH_CHECK();
  if (pcl) {
    hipLaunchKernelGGL(shot_pcl_long_kernel, dim3((unsigned)((n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN < 16384 ? (n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN : 16384)), dim3(64), 0, st, n, pts, pt_off,
                       (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, normal_r, (const int32_t*)w.nbr_list,
                       (const int32_t*)w.nbr_cnt, w.sums);
    CPPF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(shot_eig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pts, w.sums, out_normal,
                     w.pre, (int)pcl);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

extern "C" int cppf_shot_describe(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                                  const float* normals, float shot_r, int nan_to_zero, float* out_shot, float* out_rf,
                                  void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && pt_off && normals && out_shot && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  CPPF_CHECK_ARG(workspace && workspace_bytes >= cppf_shot352_workspace_bytes(B, total_points));
  const ShotWs w = carve(workspace, B, total_points);
  hipLaunchKernelGGL(shot_sort_normals_kernel, dim3((unsigned)((total_points + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, total_points, normals, pt_off, w.scene_of, w.sorted_idx, w.sorted_nrm);
  CPPF_LAUNCH_CHECK();
  hipLaunchKernelGGL(shot_hist_kernel<false>, dim3((unsigned)total_points), dim3(64), 0, (hipStream_t)stream, B, pts, pt_off,
                     w.hdr, w.cell_start, w.sorted_idx, w.sorted_pts, w.scene_of, w.sorted_nrm, w.pre, shot_r, nan_to_zero,
                     w.nbr_list, w.nbr_cnt, out_shot, out_rf);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// ---------------------------------------------------------------------------------------------
// shot.compute_color (src_shot/shot.cpp:102-161): SHOT1344 = 352 shape + 992 colour entries per point.
// Same pipeline as cppf_shot352 with the two-channel histogram kernel; the colour of a point enters through its
// normalised CIELab triple (shot_lab_kernel).  Workspace = cppf_shot352's + two float4 tables of Lab values.
// ---------------------------------------------------------------------------------------------
extern "C" int64_t cppf_shot1344_workspace_bytes(int B, int64_t total_points) {
  if (B <= 0 || total_points <= 0) return 0;
  return cppf_shot352_workspace_bytes(B, total_points) + 2 * up256(total_points * 16);
}

extern "C" int cppf_shot1344(int B, const float* pts, const float* colors, const int32_t* pt_off, int64_t total_points,
                             float normal_r, float shot_r, float* out_shot, float* out_normal, void* workspace,
                             int64_t workspace_bytes, int flags, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && colors && pt_off && out_shot && out_normal && normal_r > 0.0f && shot_r > 0.0f);
  CPPF_CHECK_ARG(total_points >= 0 && total_points <= 0x7fffffffLL);
  if (total_points <= 0) return CPPF_OK;
  CPPF_CHECK_ARG(workspace && workspace_bytes >= cppf_shot1344_workspace_bytes(B, total_points));
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = total_points;
  const ShotWs w = carve(workspace, B, n);
  float4* lab = (float4*)((char*)workspace + cppf_shot352_workspace_bytes(B, n));
  float4* sorted_lab = (float4*)((char*)lab + up256(n * 16));
  hipLaunchKernelGGL(shot_cells_kernel, dim3(B), dim3(1024), 0, st, pts, pt_off, fmaxf(normal_r, shot_r), w.hdr,
                     w.cell_start, w.sorted_idx, w.sorted_pts, w.scene_of);
  CPPF_LAUNCH_CHECK();
  const bool pcl = !(flags & CPPF_SHOT_F64_NORMALS);
  launch_cov(pcl, dim3((unsigned)n), st, B, pts, pt_off, (const CellHdr*)w.hdr, (const int32_t*)w.cell_start,
             (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, normal_r, shot_r, w.sums, w.nbr_list, w.nbr_cnt);
  CPPF_LAUNCH_CHECK();
  if (pcl) {
    hipLaunchKernelGGL(shot_pcl_long_kernel, dim3((unsigned)((n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN < 16384 ? (n + PCL_LONG_SCAN - 1) / PCL_LONG_SCAN : 16384)), dim3(64), 0, st, n, pts, pt_off,
                       (const float4*)w.sorted_pts, (const int32_t*)w.scene_of, normal_r, (const int32_t*)w.nbr_list,
                       (const int32_t*)w.nbr_cnt, w.sums);
    CPPF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(shot_eig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, pts, w.sums, out_normal,
                     w.pre, (int)pcl);
  CPPF_LAUNCH_CHECK();
  hipLaunchKernelGGL(shot_sort_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, out_normal, pt_off,
                     w.scene_of, w.sorted_idx, w.sorted_nrm);
  hipLaunchKernelGGL(shot_lab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, colors, pt_off, w.scene_of,
                     w.sorted_idx, lab, sorted_lab);
  CPPF_LAUNCH_CHECK();
  hipLaunchKernelGGL(shot_hist_kernel<true>, dim3((unsigned)n), dim3(64), 0, st, B, pts, pt_off, w.hdr, w.cell_start,
                     w.sorted_idx, w.sorted_pts, w.scene_of, w.sorted_nrm, w.pre, shot_r, 0, w.nbr_list, w.nbr_cnt,
                     out_shot, (float*)nullptr, (const float4*)lab, (const float4*)sorted_lab);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}
