// Standalone timing probe for reslayer_split_kernel variants (measurement tooling, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I cppf2_amd/csrc [-DRS_DBG_...] scratch/rs/rs_probe.hip -o /tmp/rs_probe
// Runs the tuple MLP's three wide launches + the narrow chain at bench size with random operands and prints ms per launch.
#include "../../cppf2_amd/csrc/cppf_mlp_split.hip"
#include <vector>
#include <random>
#include <cstring>
thread_local char g_cppf_err[256];

static uint16_t bf16_of(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 1280000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  float *x, *y, *prior, *uni, *b;
  int32_t* bins;
  CK(hipMalloc(&x, rows * 368 * 4));
  CK(hipMalloc(&y, rows * 256 * 4));
  CK(hipMalloc(&prior, rows * 192 * 4));
  CK(hipMalloc(&uni, rows * 6 * 4));
  CK(hipMalloc(&bins, rows * 6 * 4));
  CK(hipMalloc(&b, 16 * 256 * 4));
  {
    std::vector<float> h((size_t)rows * 368);
    for (size_t i = 0; i < h.size(); ++i) h[i] = nd(rng);
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(prior, h.data(), (size_t)rows * 192 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)rows * 6; ++i) h[i] = (float)(rng() & 0xffffff) / 16777216.0f;
    CK(hipMemcpy(uni, h.data(), (size_t)rows * 6 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 16 * 256; ++i) h[i] = 0.1f * nd(rng);
    CK(hipMemcpy(b, h.data(), 16 * 256 * 4, hipMemcpyHostToDevice));
  }
  struct Shape { const char* name; int k, n, proj, chain, decode; };
  const Shape shapes[] = {{"256->256 id, chain 1", 256, 256, 0, 1, 0}, {"128->256 proj", 128, 256, 1, 0, 0},
                          {"256->192 proj + draw", 256, 192, 1, 0, 1}, {"256->192 proj", 256, 192, 1, 0, 0},
                          {"360->128 proj, chain 4", 360, 128, 1, 4, 0}, {"128->128 id", 128, 128, 0, 0, 0}};
  for (const Shape& s : shapes) {
    const int64_t bytes = cppf_reslayer_split_stream_bytes(s.k, s.n, s.proj, s.chain);
    std::vector<uint16_t> w(bytes / 2);
    // fragments are (hi, mid, lo) triples of 1 KiB: magnitudes like a real split
    const float sc = 1.0f / sqrtf((float)s.k);
    for (size_t i = 0; i < w.size(); ++i) {
      const int part = (int)((i / 512) % 3);
      w[i] = bf16_of(nd(rng) * sc * (part == 0 ? 1.f : part == 1 ? 1.f / 256 : 1.f / 65536));
    }
    void* wq;
    CK(hipMalloc(&wq, bytes));
    CK(hipMemcpy(wq, w.data(), bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipEventRecord(e0, 0));
      int rc;
      if (s.decode)
        rc = cppf_reslayer_split_decode(y, 256, s.k, rows, wq, bytes, b, b + 2048, prior, uni, bins, nullptr);
      else
        rc = cppf_reslayer_split(s.k == 360 ? x : y, s.k == 360 ? 368 : 256, s.k, s.proj ? x : y, s.proj ? 368 : 256, s.n, rows, wq, bytes, b,
                                 s.proj ? b + 2048 : nullptr, s.chain, nullptr);
      if (rc) { printf("rc %d %s\n", rc, g_cppf_err); return 1; }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double kp = (s.k + 15) / 16 * 16;
    const double mf = 6.0 * 2.0 * rows * (kp * s.n * (s.proj ? 2 : 1) + (double)s.n * s.n * (1 + 2 * s.chain));
    printf("%-26s avg %.3f ms  best %.3f ms  %.0f TF/s bf16-executed (%.3f of 2500)\n", s.name, sum / reps, best, mf / (sum / reps) / 1e9,
           mf / (sum / reps) / 1e9 / 2500);
    fflush(stdout);
    CK(hipFree(wq));
    // re-randomise y between shapes (outputs may have blown up through repeated in-place identity layers)
    CK(hipMemcpy(y, x, (size_t)rows * 256 * 4, hipMemcpyDeviceToDevice));
  }
  return 0;
}
