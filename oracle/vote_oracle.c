/* C restatement of the sphere-bin count of get_topk_dir (eval.py:37-51) -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * The same arithmetic as oracle/cppf_oracle.py:get_topk_dir (which stays the definition; tests/test_oracle_golden.py holds this
 * file to it bit for bit and both to the reference's own outputs in tests/golden/small.npz / full_summary.json):
 *   eval.py:41-43  chunks of bmm_size candidate rows;
 *   eval.py:43     sim = mm(pred_chunk, sphere_pts.T) in float32, K = 3 accumulated the way an FMA sgemm micro-kernel does:
 *                  fma(a2, b2, fma(a1, b1, a0 * b0));
 *   eval.py:45     (sim > cos(2 angle_tol)) compared in float32, divided by the float64 pair weights, summed over the chunk's rows
 *                  in float64 (row order, in sub-chunks of 8192 rows like the NumPy restatement), added to the float32 counts:
 *                  ONE float32 rounding per chunk.
 * Used by bench.py's cpu_baseline leg (the reference's CPU path at its best: all cores) and by the whole-batch agreement check;
 * OpenMP over the sphere bins -- every (row, bin) pair is evaluated by exactly one thread and every bin's sum runs in row order,
 * so the counts do not depend on the thread count. */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SUB 8192

int vote_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* pred f32[m,3], sphere f32[s,3], wt f64[m] (divisors), thr f32, counts f32[s] (in/out: accumulated chunk by chunk). */
void vote_oracle_sphere_counts(const float* pred, long m, const float* sphere, int s, const double* wt, float thr, long bmm_size,
                               int threads, float* counts) {
  if (bmm_size <= 0) bmm_size = m > 0 ? m : 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  for (long c0 = 0; c0 < m || c0 == 0; c0 += bmm_size) {
    const long c1 = c0 + bmm_size < m ? c0 + bmm_size : m;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < s; ++b) {
      const float bx = sphere[3 * b], by = sphere[3 * b + 1], bz = sphere[3 * b + 2];
      double part = 0.0;
      for (long j0 = c0; j0 < c1; j0 += SUB) {
        const long j1 = j0 + SUB < c1 ? j0 + SUB : c1;
        double sub = 0.0;
        for (long j = j0; j < j1; ++j) {
          const float* a = pred + 3 * j;
          const float cs = fmaf(a[2], bz, fmaf(a[1], by, a[0] * bx));
          sub += (cs > thr ? 1.0 : 0.0) / wt[j];
        }
        part = part + sub;
      }
      counts[b] = (float)((double)counts[b] + part);
    }
    if (m == 0) break;
  }
}
