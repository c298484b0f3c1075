/*
 * shot_oracle.c -- CPU oracle for shot.compute / estimate_normal.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * PARITY UNPINNED.  The reference's src_shot/shot.cpp:12-100 is a thin pybind11 wrapper over
 * PCL 1.9.1 (environment.yml:39: pcl=1.9.1, flann=1.9.1): pcl::NormalEstimation<PointXYZ,Normal>,
 * pcl::SHOTEstimation<PointXYZ,Normal,SHOT352> (+ SHOTLocalReferenceFrameEstimation) over a
 * pcl::search::KdTree (FLANN).  PCL is a third-party dependency that is NOT under /root/reference
 * and is not installed in this image; the reference has no tests or golden vectors for it.  This file
 * therefore restates PCL 1.9.1's published algorithm (features/impl/normal_3d.hpp, shot_lrf.hpp,
 * shot.hpp) from its documentation and is validated by invariants and hand-built known-answer
 * clouds (tests/test_shot.py), not by PCL outputs.
 *
 * Stated deviations from PCL 1.9.1 (all below PCL's own float rounding noise):
 *  - neighbour order is ascending point index (PCL: FLANN distance order).  Only float summation
 *    order depends on it, except the 5-point median rule of the LRF sign tie-break, which is
 *    evaluated here on the (distance, index)-sorted list like PCL does.
 *  - normal covariance is accumulated in double about the query point (PCL 1.9.1: single-pass float
 *    on raw coordinates, whose cancellation error at z ~ 0.8 m is ~1e-3 relative).
 *  - 3x3 symmetric eigenproblems use cyclic Jacobi in double (PCL: closed-form float eigen33 for
 *    normals, Eigen::SelfAdjointEigenSolver<Matrix3d> for the LRF).
 *
 * "PCL-arithmetic" mode (shot_oracle_compute_ex, mode = 1) removes the three deviations as far as a restatement can:
 * neighbours in (distance, index) order for every sum, the normal's covariance single-pass in float on raw coordinates
 * (computeMeanAndCovarianceMatrix of 1.9.1) with the closed-form float eigen33 / computeRoots, float viewpoint flip.
 * (Eigen's iterative SelfAdjointEigenSolver<Matrix3d> of the LRF stays a double Jacobi: both are accurate to ~1e-15.)
 * tests/test_shot.py reports the difference between the two modes on the bench clouds -- the measured size of the
 * stated deviations, i.e. the bound to quote for "deviation from PCL's arithmetic".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SHOT_LEN 352
#define NR_BINS 10          /* nr_shape_bins_ */
#define MAX_SECTORS 32

/* Cyclic Jacobi for a symmetric 3x3 matrix; eigenvalues ascending in w, eigenvectors in columns of v. */
static void jacobi3(const double a_in[6] /* xx xy xz yy yz zz */, double w[3], double v[3][3]) {
  double a[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p][q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0);
        const double s = t * c;
        a[p][p] = a[p][p] - t * apq;
        a[q][q] = a[q][q] + t * apq;
        a[p][q] = 0.0;
        a[q][p] = 0.0;
        const int r = 3 - p - q;
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = c * arp - s * arq;
        a[p][r] = a[r][p];
        a[r][q] = s * arp + c * arq;
        a[q][r] = a[r][q];
        for (int i = 0; i < 3; ++i) {
          const double vip = v[i][p], viq = v[i][q];
          v[i][p] = c * vip - s * viq;
          v[i][q] = s * vip + c * viq;
        }
      }
    }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[order[j]] > d[order[j + 1]]) { int tmp = order[j]; order[j] = order[j + 1]; order[j + 1] = tmp; }
  double vv[3][3];
  memcpy(vv, v, sizeof(vv));
  for (int k = 0; k < 3; ++k) {
    w[k] = d[order[k]];
    for (int i = 0; i < 3; ++i) v[i][k] = vv[i][order[k]];
  }
}

static float sqdist(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

/* pcl::NormalEstimation with radius search + flipNormalTowardsViewpoint(vp = origin). */
void shot_oracle_normals(const float* pts, int n, float radius, float* normals) {
  const float r2 = radius * radius;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + 3 * i;
    double s[9] = {0};
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
      const float* q = pts + 3 * j;
      if (!(sqdist(p, q) < r2)) continue;
      const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
      s[0] += x * x; s[1] += x * y; s[2] += x * z; s[3] += y * y; s[4] += y * z; s[5] += z * z;
      s[6] += x; s[7] += y; s[8] += z;
      ++cnt;
    }
    float* o = normals + 3 * i;
    if (cnt < 3) { o[0] = o[1] = o[2] = NAN; continue; }
    const double inv = 1.0 / (double)cnt;
    const double mx = s[6] * inv, my = s[7] * inv, mz = s[8] * inv;
    const double cov[6] = {s[0] * inv - mx * mx, s[1] * inv - mx * my, s[2] * inv - mx * mz,
                           s[3] * inv - my * my, s[4] * inv - my * mz, s[5] * inv - mz * mz};
    double w[3], v[3][3];
    jacobi3(cov, w, v);
    double nx = v[0][0], ny = v[1][0], nz = v[2][0];
    /* flip towards the viewpoint (0,0,0): (vp - p) . n < 0  -> negate */
    const double ct = (0.0 - (double)p[0]) * nx + (0.0 - (double)p[1]) * ny + (0.0 - (double)p[2]) * nz;
    if (ct < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
    o[0] = (float)nx; o[1] = (float)ny; o[2] = (float)nz;
  }
}

typedef struct { float d2; int idx; } Nb;
static int nb_cmp(const void* a, const void* b) {
  const Nb* x = (const Nb*)a; const Nb* y = (const Nb*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* SHOTLocalReferenceFrameEstimation::getLocalRF.  rf = rows x, y, z (float); returns 0 if NaN. */
static int shot_lrf(const float* pts, int n, int i, float radius, float rf[9]) {
  const float* p = pts + 3 * i;
  const float r2 = radius * radius;
  double cov[6] = {0}, sum = 0.0;
  int valid = 0;
  for (int j = 0; j < n; ++j) {
    const float* q = pts + 3 * j;
    const float d2 = sqdist(p, q);
    if (!(d2 < r2)) continue;
    if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) continue;
    const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
    const double w = (double)radius - (double)sqrtf(d2);
    cov[0] += w * (x * x); cov[1] += w * (x * y); cov[2] += w * (x * z);
    cov[3] += w * (y * y); cov[4] += w * (y * z); cov[5] += w * (z * z);
    sum += w;
    ++valid;
  }
  if (valid < 5) { for (int c = 0; c < 9; ++c) rf[c] = NAN; return 0; }
  for (int c = 0; c < 6; ++c) cov[c] /= sum;
  double w[3], v[3][3];
  jacobi3(cov, w, v);
  if (!isfinite(w[0]) || !isfinite(w[1]) || !isfinite(w[2])) { for (int c = 0; c < 9; ++c) rf[c] = NAN; return 0; }
  double v1[3] = {v[0][2], v[1][2], v[2][2]};   /* largest eigenvalue  -> x */
  double v3[3] = {v[0][0], v[1][0], v[2][0]};   /* smallest eigenvalue -> z */
  int plus1 = 0, plus3 = 0;
  for (int j = 0; j < n; ++j) {
    const float* q = pts + 3 * j;
    const float d2 = sqdist(p, q);
    if (!(d2 < r2)) continue;
    if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) continue;
    const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
    if ((x * v1[0] + y * v1[1]) + z * v1[2] >= 0.0) ++plus1;
    if ((x * v3[0] + y * v3[1]) + z * v3[2] >= 0.0) ++plus3;
  }
  plus1 = 2 * plus1 - valid;
  plus3 = 2 * plus3 - valid;
  if (plus1 == 0 || plus3 == 0) {
    /* tie: look at the 5 neighbours around the median of the distance-sorted valid list */
    Nb* nb = (Nb*)malloc(sizeof(Nb) * (size_t)valid);
    int m = 0;
    for (int j = 0; j < n; ++j) {
      const float* q = pts + 3 * j;
      const float d2 = sqdist(p, q);
      if (!(d2 < r2)) continue;
      if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) continue;
      nb[m].d2 = d2; nb[m].idx = j; ++m;
    }
    qsort(nb, (size_t)m, sizeof(Nb), nb_cmp);
    const int med = valid / 2;
    int c1 = 0, c3 = 0;
    for (int t = -2; t <= 2; ++t) {
      const float* q = pts + 3 * nb[med - t].idx;
      const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
      if ((x * v1[0] + y * v1[1]) + z * v1[2] > 0.0) ++c1;
      if ((x * v3[0] + y * v3[1]) + z * v3[2] > 0.0) ++c3;
    }
    free(nb);
    if (plus1 == 0) plus1 = (c1 < 3) ? -1 : 1;
    if (plus3 == 0) plus3 = (c3 < 3) ? -1 : 1;
  }
  if (plus1 < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
  if (plus3 < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
  rf[0] = (float)v1[0]; rf[1] = (float)v1[1]; rf[2] = (float)v1[2];
  rf[6] = (float)v3[0]; rf[7] = (float)v3[1]; rf[8] = (float)v3[2];
  /* y = z x x (float) */
  rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
  rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
  rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  return 1;
}

#define RAD_45 0.78539816339744830961566084581988
#define RAD_90 1.5707963267948966192313216916398
#define RAD_135 2.3561944901923449288469825374596
#define RAD_PI_7_8 2.7488935718910690836548129603691

/* one neighbour's contribution (SHOTEstimation::interpolateSingleChannel body); returns nothing, adds into shot[] */
#define NR_COLOR_BINS 30   /* nr_color_bins_ of SHOTColorEstimation */
#define COLOR_OFF (MAX_SECTORS * (NR_BINS + 1))

/* bdc < 0: shape channel only (SHOT352, interpolateSingleChannel); bdc >= 0: colour bin coordinate of this neighbour,
 * SHOT1344's second channel (interpolateDoubleChannel): same sector, same spatial interpolation terms, its own step. */
static void shot_accumulate2(const float* p, const float* q, float d2, const float* nq, const float rf[9], double radius,
                             float* shot, double* margin, double bdc) {
  /* createBinDistanceShape */
  if (!isfinite(nq[0]) || !isfinite(nq[1]) || !isfinite(nq[2])) return;
  double cosd = (double)((nq[0] * rf[6] + nq[1] * rf[7]) + nq[2] * rf[8]);
  if (cosd > 1.0) cosd = 1.0;
  if (cosd < -1.0) cosd = -1.0;
  double bin_distance = ((1.0 + cosd) * NR_BINS) / 2;
  /* PCL's "quadrilinear interpolation" adds the SUM of the four 1-D weights to the neighbour's own bin and only the
   * individual complements to the adjacent bins, so the descriptor is a discontinuous function of the neighbour: it
   * jumps by (w - 0.5) / |h| wherever the neighbour crosses a decision boundary -- a cosine step (bin_distance = k + 0.5),
   * the radial shell (distance = r/2), the equator of the frame (zf = 0), an azimuth octant (xf = 0, yf = 0,
   * |xf| = |yf|) -- and where it enters the support (distance = r).  `margin` records how close this neighbour comes to
   * the nearest of them (dimensionless), so a test can tell a rounding-induced jump from an error. */
  const double cos_margin = fabs((bin_distance - floor(bin_distance)) - 0.5);
  const double distance = sqrt((double)d2);
  if (fabs(distance) < 1e-15) return;
  const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
  double xf = (double)((dx * rf[0] + dy * rf[1]) + dz * rf[2]);
  double yf = (double)((dx * rf[3] + dy * rf[4]) + dz * rf[5]);
  double zf = (double)((dx * rf[6] + dy * rf[7]) + dz * rf[8]);
  if (fabs(yf) < 1e-30) yf = 0;
  if (fabs(xf) < 1e-30) xf = 0;
  if (fabs(zf) < 1e-30) zf = 0;
  const double r12 = radius / 2.0, r14 = radius / 4.0, r34 = radius * 3.0 / 4.0;
  if (margin) {
    double mg = cos_margin;
    const double m_rad = fabs(distance - r12) / radius;
    const double m_el = fabs(zf) / distance;
    const double ax = fabs(xf), ay = fabs(yf);
    double m_az = ax < ay ? ax : ay;
    if (fabs(ax - ay) < m_az) m_az = fabs(ax - ay);
    m_az /= distance;
    if (m_rad < mg) mg = m_rad;
    if (m_el < mg) mg = m_el;
    if (m_az < mg) mg = m_az;
    if (mg < *margin) *margin = mg;
  }

  const int bit4 = ((yf > 0) || ((yf == 0.0) && (xf < 0))) ? 1 : 0;
  const int bit3 = ((xf > 0) || ((xf == 0.0) && (yf > 0))) ? !bit4 : bit4;
  int desc_index = (bit4 << 3) + (bit3 << 2);
  desc_index = desc_index << 1;
  if ((xf * yf > 0) || (xf == 0.0))
    desc_index += (fabs(xf) >= fabs(yf)) ? 0 : 4;
  else
    desc_index += (fabs(xf) > fabs(yf)) ? 4 : 0;
  desc_index += zf > 0 ? 1 : 0;
  desc_index += (distance > r12) ? 2 : 0;

  const int step_index = (int)floor(bin_distance + 0.5);
  const int volume_index = desc_index * (NR_BINS + 1);
  bin_distance -= step_index;
  double w = 1.0 - fabs(bin_distance);
  if (bin_distance > 0)
    shot[volume_index + ((step_index + 1) % NR_BINS)] += (float)bin_distance;
  else
    shot[volume_index + ((step_index - 1 + NR_BINS) % NR_BINS)] += -(float)bin_distance;
  const int has_c = bdc >= 0.0;
  const int cstride = NR_COLOR_BINS + 1;
  int step_c = 0;
  double wc = 0.0;
  if (has_c) {
    if (margin) {
      const double cm = fabs((bdc - floor(bdc)) - 0.5);
      if (cm < *margin) *margin = cm;
    }
    step_c = (int)floor(bdc + 0.5);
    const int vol_c = COLOR_OFF + desc_index * cstride;
    bdc -= step_c;
    wc = 1.0 - fabs(bdc);
    if (bdc > 0)
      shot[vol_c + ((step_c + 1) % NR_COLOR_BINS)] += (float)bdc;
    else
      shot[vol_c + ((step_c - 1 + NR_COLOR_BINS) % NR_COLOR_BINS)] -= (float)bdc;
  }

  if (distance > r12) {
    const double rd = (distance - r34) / r12;
    if (distance > r34) { w += 1 - rd; wc += 1 - rd; }
    else {
      w += 1 + rd; wc += 1 + rd;
      shot[(desc_index - 2) * (NR_BINS + 1) + step_index] -= (float)rd;
      if (has_c) shot[COLOR_OFF + (desc_index - 2) * cstride + step_c] -= (float)rd;
    }
  } else {
    const double rd = (distance - r14) / r12;
    if (distance < r14) { w += 1 + rd; wc += 1 + rd; }
    else {
      w += 1 - rd; wc += 1 - rd;
      shot[(desc_index + 2) * (NR_BINS + 1) + step_index] += (float)rd;
      if (has_c) shot[COLOR_OFF + (desc_index + 2) * cstride + step_c] += (float)rd;
    }
  }

  double inc_cos = zf / distance;
  if (inc_cos < -1.0) inc_cos = -1.0;
  if (inc_cos > 1.0) inc_cos = 1.0;
  const double inc = acos(inc_cos);
  if (inc > RAD_90 || (fabs(inc - RAD_90) < 1e-30 && zf <= 0)) {
    const double id = (inc - RAD_135) / RAD_90;
    if (inc > RAD_135) { w += 1 - id; wc += 1 - id; }
    else {
      w += 1 + id; wc += 1 + id;
      shot[(desc_index + 1) * (NR_BINS + 1) + step_index] -= (float)id;
      if (has_c) shot[COLOR_OFF + (desc_index + 1) * cstride + step_c] -= (float)id;
    }
  } else {
    const double id = (inc - RAD_45) / RAD_90;
    if (inc < RAD_45) { w += 1 + id; wc += 1 + id; }
    else {
      w += 1 - id; wc += 1 - id;
      shot[(desc_index - 1) * (NR_BINS + 1) + step_index] += (float)id;
      if (has_c) shot[COLOR_OFF + (desc_index - 1) * cstride + step_c] += (float)id;
    }
  }

  if (yf != 0.0 || xf != 0.0) {
    const double az = atan2(yf, xf);
    const int sel = desc_index >> 2;
    double ad = (az - (-RAD_PI_7_8 + RAD_45 * sel)) / RAD_45;
    ad = fmax(-0.5, fmin(ad, 0.5));
    if (ad > 0) {
      w += 1 - ad; wc += 1 - ad;
      shot[((desc_index + 4) % MAX_SECTORS) * (NR_BINS + 1) + step_index] += (float)ad;
      if (has_c) shot[COLOR_OFF + ((desc_index + 4) % MAX_SECTORS) * cstride + step_c] += (float)ad;
    } else {
      w += 1 + ad; wc += 1 + ad;
      shot[((desc_index - 4 + MAX_SECTORS) % MAX_SECTORS) * (NR_BINS + 1) + step_index] -= (float)ad;
      if (has_c) shot[COLOR_OFF + ((desc_index - 4 + MAX_SECTORS) % MAX_SECTORS) * cstride + step_c] -= (float)ad;
    }
  }
  shot[volume_index + step_index] += (float)w;
  if (has_c) shot[COLOR_OFF + desc_index * cstride + step_c] += (float)wc;
}

static void shot_accumulate(const float* p, const float* q, float d2, const float* nq, const float rf[9], double radius,
                            float* shot, double* margin) {
  shot_accumulate2(p, q, d2, nq, rf, radius, shot, margin, -1.0);
}

/* shot.compute(pc, normal_r, shot_r): out_shot [n,352], out_normal [n,3]; optional out_rf [n,9]. */
void shot_oracle_compute(const float* pts, int n, float normal_r, float shot_r, float* out_shot, float* out_normal,
                         float* out_rf) {
  shot_oracle_normals(pts, n, normal_r, out_normal);
  const float r2 = shot_r * shot_r;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + 3 * i;
    float* shot = out_shot + (size_t)SHOT_LEN * i;
    float rf[9];
    const int ok = shot_lrf(pts, n, i, shot_r, rf);
    if (out_rf) memcpy(out_rf + 9 * i, rf, sizeof(rf));
    int nn = 0;
    for (int j = 0; j < n; ++j) nn += (sqdist(p, pts + 3 * j) < r2) ? 1 : 0;
    if (!ok || nn < 5) { for (int c = 0; c < SHOT_LEN; ++c) shot[c] = NAN; continue; }
    memset(shot, 0, sizeof(float) * SHOT_LEN);
    for (int j = 0; j < n; ++j) {
      const float* q = pts + 3 * j;
      const float d2 = sqdist(p, q);
      if (!(d2 < r2)) continue;
      shot_accumulate(p, q, d2, out_normal + 3 * j, rf, (double)shot_r, shot, NULL);
    }
    double acc = 0.0;
    for (int c = 0; c < SHOT_LEN; ++c) acc += (double)shot[c] * (double)shot[c];
    acc = sqrt(acc);
    for (int c = 0; c < SHOT_LEN; ++c) shot[c] /= (float)acc;
  }
}


/* ------------------------------------------------------------------------------------------------------------------
 * PCL-arithmetic mode
 * ------------------------------------------------------------------------------------------------------------------ */
/* pcl::computeRoots2 / computeRoots (common/impl/eigen.hpp), Scalar = float: eigenvalues ascending */
static void pcl_roots2(float b, float c, float roots[3]) {
  roots[0] = 0.0f;
  float d = b * b - 4.0f * c;
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

static void pcl_roots(const float m[3][3], float roots[3]) {
  const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
                   m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
  const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] -
                   m[1][2] * m[1][2];
  const float c2 = m[0][0] + m[1][1] + m[2][2];
  if (fabsf(c0) < 1.1920929e-07f) { pcl_roots2(c2, c1, roots); return; }
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.0f) a_over_3 = 0.0f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.0f) q = 0.0f;
  const float rho = sqrtf(-a_over_3);
  const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
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

/* pcl::eigen33 (smallest eigenvalue's eigenvector of a symmetric float 3x3) */
static void pcl_eigen33(const float cov[3][3], float vec[3]) {
  float scale = 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) scale = fmaxf(scale, fabsf(cov[i][j]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float m[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = cov[i][j] / scale;
  float roots[3];
  pcl_roots(m, roots);
  for (int i = 0; i < 3; ++i) m[i][i] -= roots[0];
  float v[3][3];
  const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  float len[3];
  for (int k = 0; k < 3; ++k) {
    const float* a = m[pr[k][0]];
    const float* b = m[pr[k][1]];
    v[k][0] = a[1] * b[2] - a[2] * b[1];
    v[k][1] = a[2] * b[0] - a[0] * b[2];
    v[k][2] = a[0] * b[1] - a[1] * b[0];
    len[k] = v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2];
  }
  int best = 2;
  if (len[0] >= len[1] && len[0] >= len[2]) best = 0;
  else if (len[1] >= len[0] && len[1] >= len[2]) best = 1;
  const float inv = sqrtf(len[best]);
  for (int c = 0; c < 3; ++c) vec[c] = v[best][c] / inv;
}

/* (distance, index)-sorted neighbours of point i within `radius` (strict <), like a sorted FLANN radius search */
static int sorted_neighbours(const float* pts, int n, int i, float radius, Nb* nb) {
  const float r2 = radius * radius;
  int m = 0;
  for (int j = 0; j < n; ++j) {
    const float d2 = sqdist(pts + 3 * i, pts + 3 * j);
    if (d2 < r2) { nb[m].d2 = d2; nb[m].idx = j; ++m; }
  }
  qsort(nb, (size_t)m, sizeof(Nb), nb_cmp);
  return m;
}

static void pcl_normal(const float* pts, int i, const Nb* nb, int m, float* o) {
  if (m < 3) { o[0] = o[1] = o[2] = NAN; return; }
  float accu[9] = {0};
  for (int t = 0; t < m; ++t) {                      /* computeMeanAndCovarianceMatrix: raw coordinates, float */
    const float* q = pts + 3 * nb[t].idx;
    accu[0] += q[0] * q[0]; accu[1] += q[0] * q[1]; accu[2] += q[0] * q[2];
    accu[3] += q[1] * q[1]; accu[4] += q[1] * q[2]; accu[5] += q[2] * q[2];
    accu[6] += q[0]; accu[7] += q[1]; accu[8] += q[2];
  }
  for (int c = 0; c < 9; ++c) accu[c] /= (float)m;
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
  const float* p = pts + 3 * i;                      /* flipNormalTowardsViewpoint, viewpoint (0,0,0), float */
  const float vx = 0.0f - p[0], vy = 0.0f - p[1], vz = 0.0f - p[2];
  const float ct = vx * v[0] + vy * v[1] + vz * v[2];
  if (ct < 0.0f) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
}

/* getLocalRF over a distance-sorted list; diag (optional): eigenvalues descending, the two sign tallies */
static int shot_lrf_sorted(const float* pts, int i, const Nb* nb, int m, float radius, float rf[9], double* diag) {
  const float* p = pts + 3 * i;
  double cov[6] = {0}, sum = 0.0;
  int valid = 0;
  for (int t = 0; t < m; ++t) {
    const float* q = pts + 3 * nb[t].idx;
    if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) continue;
    const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
    const double w = (double)radius - (double)sqrtf(nb[t].d2);
    cov[0] += w * (x * x); cov[1] += w * (x * y); cov[2] += w * (x * z);
    cov[3] += w * (y * y); cov[4] += w * (y * z); cov[5] += w * (z * z);
    sum += w;
    ++valid;
  }
  if (diag) { diag[0] = diag[1] = diag[2] = NAN; diag[3] = diag[4] = 0; }
  if (valid < 5) { for (int c = 0; c < 9; ++c) rf[c] = NAN; return 0; }
  for (int c = 0; c < 6; ++c) cov[c] /= sum;
  double w[3], v[3][3];
  jacobi3(cov, w, v);
  if (!isfinite(w[0]) || !isfinite(w[1]) || !isfinite(w[2])) { for (int c = 0; c < 9; ++c) rf[c] = NAN; return 0; }
  double v1[3] = {v[0][2], v[1][2], v[2][2]};
  double v3[3] = {v[0][0], v[1][0], v[2][0]};
  int plus1 = 0, plus3 = 0;
  int* vidx = (int*)malloc(sizeof(int) * (size_t)valid);
  int k = 0;
  for (int t = 0; t < m; ++t) {
    const float* q = pts + 3 * nb[t].idx;
    if (q[0] == p[0] && q[1] == p[1] && q[2] == p[2]) continue;
    vidx[k++] = nb[t].idx;
    const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
    if ((x * v1[0] + y * v1[1]) + z * v1[2] >= 0.0) ++plus1;
    if ((x * v3[0] + y * v3[1]) + z * v3[2] >= 0.0) ++plus3;
  }
  plus1 = 2 * plus1 - valid;
  plus3 = 2 * plus3 - valid;
  if (diag) { diag[0] = w[2]; diag[1] = w[1]; diag[2] = w[0]; diag[3] = plus1; diag[4] = plus3; }
  if (plus1 == 0 || plus3 == 0) {
    const int med = valid / 2;
    int c1 = 0, c3 = 0;
    for (int t = -2; t <= 2; ++t) {
      const float* q = pts + 3 * vidx[med - t];
      const double x = (double)(q[0] - p[0]), y = (double)(q[1] - p[1]), z = (double)(q[2] - p[2]);
      if ((x * v1[0] + y * v1[1]) + z * v1[2] > 0.0) ++c1;
      if ((x * v3[0] + y * v3[1]) + z * v3[2] > 0.0) ++c3;
    }
    if (plus1 == 0) plus1 = (c1 < 3) ? -1 : 1;
    if (plus3 == 0) plus3 = (c3 < 3) ? -1 : 1;
  }
  free(vidx);
  if (plus1 < 0) { v1[0] = -v1[0]; v1[1] = -v1[1]; v1[2] = -v1[2]; }
  if (plus3 < 0) { v3[0] = -v3[0]; v3[1] = -v3[1]; v3[2] = -v3[2]; }
  rf[0] = (float)v1[0]; rf[1] = (float)v1[1]; rf[2] = (float)v1[2];
  rf[6] = (float)v3[0]; rf[7] = (float)v3[1]; rf[8] = (float)v3[2];
  rf[3] = rf[7] * rf[2] - rf[8] * rf[1];
  rf[4] = rf[8] * rf[0] - rf[6] * rf[2];
  rf[5] = rf[6] * rf[1] - rf[7] * rf[0];
  return 1;
}

/* mode 0: the oracle's own arithmetic (shot_oracle_compute) + diagnostics; mode 1: PCL arithmetic (see header).
 * out_diag (optional) [n,DIAG_LEN] doubles: LRF eigenvalues (descending), sign tally of x (2*plus - n), sign tally of z,
 * smallest margin of a neighbour to a decision boundary of the interpolation (see shot_accumulate), L2 norm of the
 * histogram before normalisation, neighbours within shot_r, smallest |d^2 - r^2| / r^2 over ALL points (how close a
 * point comes to entering / leaving the support: its contribution does not vanish at the rim, so that is a jump too). */
#define DIAG_LEN 9
/* Threads of shot_oracle_compute_ex (bench.py's cpu_baseline: 1 = like PCL, 0 = all cores). */
static int g_threads = 1;
int shot_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void shot_oracle_set_threads(int t) { g_threads = t; }

void shot_oracle_compute_ex(const float* pts, int n, float normal_r, float shot_r, int mode, float* out_shot,
                            float* out_normal, float* out_rf, double* out_diag) {
  /* Every query point is independent (own neighbour list, own output rows): with -fopenmp and g_threads != 1 the two loops run
   * over the points in parallel, each thread with its own neighbour buffer; the outputs do not depend on the thread count.
   * g_threads = 1 (the default) is the reference's own mode: PCL's estimators at src_shot/shot.cpp:66-89 are single-threaded. */
  const int nthreads = g_threads == 1 ? 1 : (g_threads > 1 ? g_threads : shot_oracle_max_threads());
  (void)nthreads;
  if (mode == 1) {
#pragma omp parallel num_threads(nthreads)
    {
      Nb* nb = (Nb*)malloc(sizeof(Nb) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 32)
      for (int i = 0; i < n; ++i) {
        const int m = sorted_neighbours(pts, n, i, normal_r, nb);
        pcl_normal(pts, i, nb, m, out_normal + 3 * i);
      }
      free(nb);
    }
  } else {
    shot_oracle_normals(pts, n, normal_r, out_normal);
  }
#pragma omp parallel num_threads(nthreads)
  {
  Nb* nb = (Nb*)malloc(sizeof(Nb) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 32)
  for (int i = 0; i < n; ++i) {
    const float* p = pts + 3 * i;
    float* shot = out_shot + (size_t)SHOT_LEN * i;
    float rf[9];
    const int m = sorted_neighbours(pts, n, i, shot_r, nb);
    int ok;
    if (mode == 1) {
      ok = shot_lrf_sorted(pts, i, nb, m, shot_r, rf, out_diag ? out_diag + DIAG_LEN * i : NULL);
    } else {
      float tmp[9];
      shot_lrf_sorted(pts, i, nb, m, shot_r, tmp, out_diag ? out_diag + DIAG_LEN * i : NULL);   /* diagnostics only */
      ok = shot_lrf(pts, n, i, shot_r, rf);
    }
    if (out_rf) memcpy(out_rf + 9 * i, rf, sizeof(rf));
    double wrap = 1e300;
    double* wd = out_diag ? &wrap : NULL;
    if (out_diag) {
      const double r2d = (double)(shot_r * shot_r);
      double edge = 1e300;
      for (int j = 0; j < n; ++j) {
        const double e = fabs((double)sqdist(p, pts + 3 * j) - r2d) / r2d;
        if (e < edge) edge = e;
      }
      out_diag[DIAG_LEN * i + 5] = NAN; out_diag[DIAG_LEN * i + 6] = NAN; out_diag[DIAG_LEN * i + 7] = m;
      out_diag[DIAG_LEN * i + 8] = edge;
    }
    if (!ok || m < 5) { for (int c = 0; c < SHOT_LEN; ++c) shot[c] = NAN; continue; }
    memset(shot, 0, sizeof(float) * SHOT_LEN);
    if (mode == 1) {
      for (int t = 0; t < m; ++t)
        shot_accumulate(p, pts + 3 * nb[t].idx, nb[t].d2, out_normal + 3 * nb[t].idx, rf, (double)shot_r, shot, wd);
    } else {
      const float r2 = shot_r * shot_r;
      for (int j = 0; j < n; ++j) {
        const float d2 = sqdist(p, pts + 3 * j);
        if (!(d2 < r2)) continue;
        shot_accumulate(p, pts + 3 * j, d2, out_normal + 3 * j, rf, (double)shot_r, shot, wd);
      }
    }
    double acc = 0.0;
    for (int c = 0; c < SHOT_LEN; ++c) acc += (double)shot[c] * (double)shot[c];
    acc = sqrt(acc);
    if (out_diag) { out_diag[DIAG_LEN * i + 5] = wrap; out_diag[DIAG_LEN * i + 6] = acc; }
    for (int c = 0; c < SHOT_LEN; ++c) shot[c] /= (float)acc;
  }
  free(nb);
  }
}


/* ------------------------------------------------------------------------------------------------------------------
 * SHOT1344 = shape + colour (shot.compute_color, src_shot/shot.cpp:102-161 -> pcl::SHOTColorEstimation<PointXYZRGB,
 * Normal, SHOT1344>).  PARITY UNPINNED like SHOT352.  Restated from PCL 1.9.1's features/impl/shot.hpp:
 * RGB2CIELAB through the two lookup tables (sRGB gamma at 256 levels; the cube root of XYZ quantised to 1/4000, with
 * PCL's exponent 0.3333f), L / 100, a / 120, b / 120, colour distance (|dL| + (|da| + |db|) / 2) / 3 clamped to [0, 1],
 * times 30 colour bins; interpolateDoubleChannel; L2 norm over all 1344 entries.
 * ------------------------------------------------------------------------------------------------------------------ */
#define SHOT_COLOR_LEN 1344
static float g_srgb_lut[256], g_xyz_lut[4000];
static int g_lut_ready = 0;

static void lab_tables(void) {
  if (g_lut_ready) return;
  for (int i = 0; i < 256; ++i) {
    const float f = (float)i / 255.0f;
    g_srgb_lut[i] = (f > 0.04045) ? powf((f + 0.055f) / 1.055f, 2.4f) : f / 12.92f;
  }
  for (int i = 0; i < 4000; ++i) {
    const float f = (float)i / 4000.0f;
    g_xyz_lut[i] = (f > 0.008856) ? (float)powf(f, 0.3333f) : (float)((7.787 * f) + (16.0 / 116.0));
  }
  g_lut_ready = 1;
}

/* colour components as the wrapper stores them: uint8 = (float in [0,1]) * 255.f, truncated (shot.cpp:114-116) */
static int color_u8(float c) {
  const float v = c * 255.f;
  int i = (int)v;
  if (!(v == v)) i = 0;
  if (i < 0) i = 0;
  if (i > 255) i = 255;
  return i;
}

static void rgb2lab_norm(const float* rgb, float lab[3]) {
  lab_tables();
  const float fr = g_srgb_lut[color_u8(rgb[0])], fg = g_srgb_lut[color_u8(rgb[1])], fb = g_srgb_lut[color_u8(rgb[2])];
  const float x = fr * 0.412453f + fg * 0.357580f + fb * 0.180423f;
  const float y = fr * 0.212671f + fg * 0.715160f + fb * 0.072169f;
  const float z = fr * 0.019334f + fg * 0.119193f + fb * 0.950227f;
  float vx = x / 0.95047f, vy = y, vz = z / 1.08883f;
  int ix = (int)(vx * 4000), iy = (int)(vy * 4000), iz = (int)(vz * 4000);
  if (ix > 3999) ix = 3999;
  if (iy > 3999) iy = 3999;
  if (iz > 3999) iz = 3999;
  vx = g_xyz_lut[ix]; vy = g_xyz_lut[iy]; vz = g_xyz_lut[iz];
  float L = 116.0f * vy - 16.0f;
  if (L > 100) L = 100.0f;
  float A = 500.0f * (vx - vy);
  if (A > 120) A = 120.0f; else if (A < -120) A = -120.0f;
  float B2 = 200.0f * (vy - vz);
  if (B2 > 120) B2 = 120.0f; else if (B2 < -120) B2 = -120.0f;
  lab[0] = L / 100.0f; lab[1] = A / 120.0f; lab[2] = B2 / 120.0f;
}

/* out_shot [n,1344], out_normal [n,3]; out_diag optional [n,DIAG_LEN] as in shot_oracle_compute_ex (mode 0 arithmetic) */
void shot_oracle_compute_color(const float* pts, const float* colors, int n, float normal_r, float shot_r,
                               float* out_shot, float* out_normal, double* out_diag) {
  shot_oracle_normals(pts, n, normal_r, out_normal);
  float* lab = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) rgb2lab_norm(colors + 3 * i, lab + 3 * i);
  Nb* nb = (Nb*)malloc(sizeof(Nb) * (size_t)(n > 0 ? n : 1));
  const float r2 = shot_r * shot_r;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + 3 * i;
    float* shot = out_shot + (size_t)SHOT_COLOR_LEN * i;
    float rf[9], tmp[9];
    const int m = sorted_neighbours(pts, n, i, shot_r, nb);
    shot_lrf_sorted(pts, i, nb, m, shot_r, tmp, out_diag ? out_diag + DIAG_LEN * i : NULL);      /* diagnostics only */
    const int ok = shot_lrf(pts, n, i, shot_r, rf);
    double mg = 1e300;
    if (out_diag) {
      const double r2d = (double)r2;
      double edge = 1e300;
      for (int j = 0; j < n; ++j) {
        const double e = fabs((double)sqdist(p, pts + 3 * j) - r2d) / r2d;
        if (e < edge) edge = e;
      }
      out_diag[DIAG_LEN * i + 5] = NAN; out_diag[DIAG_LEN * i + 6] = NAN; out_diag[DIAG_LEN * i + 7] = m;
      out_diag[DIAG_LEN * i + 8] = edge;
    }
    if (!ok || m < 5) { for (int c = 0; c < SHOT_COLOR_LEN; ++c) shot[c] = NAN; continue; }
    memset(shot, 0, sizeof(float) * SHOT_COLOR_LEN);
    const float* lr = lab + 3 * i;
    for (int j = 0; j < n; ++j) {
      const float d2 = sqdist(p, pts + 3 * j);
      if (!(d2 < r2)) continue;
      const float* lq = lab + 3 * j;
      double cd = (fabs(lr[0] - lq[0]) + ((fabs(lr[1] - lq[1]) + fabs(lr[2] - lq[2])) / 2)) / 3;
      if (cd > 1.0) cd = 1.0;
      if (cd < 0.0) cd = 0.0;
      shot_accumulate2(p, pts + 3 * j, d2, out_normal + 3 * j, rf, (double)shot_r, shot, out_diag ? &mg : NULL,
                       cd * NR_COLOR_BINS);
    }
    double acc = 0.0;
    for (int c = 0; c < SHOT_COLOR_LEN; ++c) acc += (double)shot[c] * (double)shot[c];
    acc = sqrt(acc);
    if (out_diag) { out_diag[DIAG_LEN * i + 5] = mg; out_diag[DIAG_LEN * i + 6] = acc; }
    for (int c = 0; c < SHOT_COLOR_LEN; ++c) shot[c] /= (float)acc;
  }
  free(nb);
  free(lab);
}
