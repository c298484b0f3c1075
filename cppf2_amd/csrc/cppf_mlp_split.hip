// cppf_mlp_split.hip -- a whole ResLayer of the tuple / point MLPs (train_shot.py:19-45) as ONE kernel on the bf16 matrix
// cores, computing in float32-equivalent arithmetic by operand splitting.
//
//   out = skip(x) + relu(x W1^T + b1) W2^T,     skip(x) = x  (dim_in == dim_out)   or   x W0^T + b0
//
// gfx950 has no xf32 / tf32 path; its f32-input MFMA runs at 1/16 of the bf16 rate.  A float32 value is EXACTLY the sum of
// three bf16 values (8 + 8 + 8 significand bits: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), round to
// nearest even each time), every bf16 x bf16 product is exact in float32, and the matrix core accumulates in float32.
// So  a * b = (ah + am + al)(bh + bm + bl)  is evaluated as the six products  ah bh + ah bm + am bh + am bm + ah bl + al bh;
// the three dropped ones (am bl, al bm, al bl) are below 2^-24 |a b| together, i.e. under half a unit in the last place of
// the product a float32 FMA chain starts from.  The result is not bit-equal to the f32-input MFMA chain (nor is any
// re-tiled float32 GEMM); its error against a float64 evaluation is measured equal to the library float32 GEMM's
// (tests/test_mlp_split_gpu.py).  Six bf16 MFMAs replace sixteen f32-input MFMA cycles' worth of work: 2.7x the rate.
//
// Structure (one wavefront = 32 rows of x, one workgroup = 8 wavefronts = 256 rows, persistent over row blocks):
//   * both products are computed TRANSPOSED (h^T = W1 x^T, out^T = skip^T + W2 h^T) with v_mfma_f32_32x32x16_bf16, so that
//     the rows of x are the COLUMNS of the accumulator tiles (lane l owns row l & 31) and the accumulators of the first
//     product are, after bias + ReLU and a split in registers, directly the B operands of the second -- h never leaves the
//     registers; the K order of a sum is free, so lane half g = l >> 5 contracts over exactly the features its accumulator
//     registers hold (row (reg & 3) + 8 (reg >> 2) + 4 g of a 32-row tile);
//   * x is read as the B operand straight from global memory (a lane reads 32 contiguous bytes of its own row per K step)
//     and split in registers ONCE per element -- a wavefront covers all output features of its rows;
//   * the A operands (the weights, pre-split on the host into the per-lane fragment order: cppf2_amd.models.pack_split)
//     stream through a two-stage LDS ring by LDS-DMA (global_load_lds_dwordx4), one K step of all output tiles per stage,
//     shared by the eight wavefronts: the layer's weights are read from L2 once per 256 rows;
//   * layers wider than 128 produce their output in two halves of the feature range (the h accumulators stay, the skip and
//     the second product run per half), which keeps a wavefront under 256 registers at two wavefronts per SIMD.
// HBM traffic per row: dim_in + dim_out floats (+ the residual re-read of an identity layer, which hits L2 / MALL).
#include "cppf_common.h"
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef RS_DMA
#define RS_DMA 1
#endif
#define RS_THREADS 512
#define RS_WAVES 8
#define RS_BLOCK_ROWS 256
#define RS_FRAG_BYTES 1024                      // one operand fragment: 64 lanes x 8 bf16
#define RS_TILE_BYTES (3 * RS_FRAG_BYTES)       // hi, mid, lo
#define RS_STAGE_BYTES (8 * RS_TILE_BYTES)      // one K step of up to 8 output tiles


struct RsFrag {
  bf16x8 h, m, l;
};

// exact three-way split of 8 floats (each step rounds to nearest even; the remainders are exact in float32)
__device__ __forceinline__ RsFrag rs_split(const float (&v)[8]) {
  RsFrag f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    const float r1 = v[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    f.h[i] = h;
    f.m[i] = m;
    f.l[i] = (__bf16)r2;
  }
  return f;
}

// acc += A B over one K step of 16 with both operands split: six exact-product MFMAs, smallest terms first
__device__ __forceinline__ void rs_mma6(f32x16& acc, const RsFrag& a, const RsFrag& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

// the weight stream of one layer, consumed in order by every wavefront of the workgroup (see pack_split):
//   KS1 chunks of NT tiles (W1), then per output half: [KS1 chunks of NTH tiles (W0)] + 2 NT chunks of NTH tiles (W2)
template <int NT, int NTH>
struct RsStream {
  const char* base;            // packed stream in global memory
  char* ring;                  // LDS, two stages
  int ks1, chunks;             // chunks per row block
  int next;                    // chunk (within the layer) the next DMA fetches
  int64_t left;                // chunks still to fetch over the remaining row blocks of this workgroup
  int stage;                   // stage the next acquire() returns
  int wave, lane;

  __device__ __forceinline__ void issue(int st) {
    if (left <= 0) return;
    const int tiles = next < ks1 ? NT : NTH;
    const int64_t off = next < ks1 ? (int64_t)next * (NT * RS_TILE_BYTES)
                                   : (int64_t)ks1 * (NT * RS_TILE_BYTES) + (int64_t)(next - ks1) * (NTH * RS_TILE_BYTES);
    const char* src = base + off + lane * 16;
    char* dst = ring + st * RS_STAGE_BYTES;
    for (int j = wave; j < tiles * 3; j += RS_WAVES) {
#if RS_DMA
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * RS_FRAG_BYTES),
                                       (__attribute__((address_space(3))) void*)(dst + j * RS_FRAG_BYTES), 16, 0, 0);
#else
      *reinterpret_cast<uint4*>(dst + j * RS_FRAG_BYTES + lane * 16) = *reinterpret_cast<const uint4*>(src + j * RS_FRAG_BYTES);
#endif
    }
    next = (next + 1 == chunks) ? 0 : next + 1;
    --left;
  }
  // the stage holding the next chunk in stream order; its DMA was issued one chunk earlier.  Every wavefront waits for
  // its own pieces (vmcnt) before the barrier, so after it the whole chunk is in LDS and the other stage is free.
  __device__ __forceinline__ const bf16x8* acquire() {
    __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    const int cur = stage;
    stage ^= 1;
    issue(stage);
    return reinterpret_cast<const bf16x8*>(ring + cur * RS_STAGE_BYTES) + lane;
  }
};

__device__ __forceinline__ RsFrag rs_read(const bf16x8* w, int tile) {
  RsFrag a;
  a.h = w[(tile * 3 + 0) * 64];
  a.m = w[(tile * 3 + 1) * 64];
  a.l = w[(tile * 3 + 2) * 64];
  return a;
}

// one K step: acc[u] += W[tile u] b for the NTILES tiles of the staged chunk.  The fragments of tile u + 1 are requested from
// LDS before the six MFMAs of tile u issue (their 192 cycles cover the read); the scheduling barrier per tile keeps the
// compiler from hoisting all the reads of a step to its front (12 registers per tile in flight).
template <int NTILES>
__device__ __forceinline__ void rs_step(f32x16 (&acc)[NTILES], const bf16x8* w, const RsFrag& b) {
  RsFrag a = rs_read(w, 0);
#pragma unroll
  for (int u = 0; u < NTILES; ++u) {
    RsFrag an = a;
    if (u + 1 < NTILES) an = rs_read(w, u + 1);
    rs_mma6(acc[u], a, b);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
  }
}

// acc[u] (u < NTILES) += W[tile u] x^T over all K steps: x is this lane's row, features 16 s + 8 g + (0..7) at step s
template <int NTILES, int NT, int NTH>
__device__ __forceinline__ void rs_product_x(f32x16 (&acc)[NTILES], const float* __restrict__ xrow, int k_in, int ks1, int g,
                                             RsStream<NT, NTH>& ws) {
  float xn[8];
  auto fetch = [&](int s) {
    const int f = 16 * s + 8 * g;
    f32x4 v0 = {0.0f, 0.0f, 0.0f, 0.0f}, v1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f + 8 <= k_in) {
      v0 = *reinterpret_cast<const f32x4*>(xrow + f);
      v1 = *reinterpret_cast<const f32x4*>(xrow + f + 4);
    }
    xn[0] = v0.x; xn[1] = v0.y; xn[2] = v0.z; xn[3] = v0.w; xn[4] = v1.x; xn[5] = v1.y; xn[6] = v1.z; xn[7] = v1.w;
  };
  fetch(0);
  for (int s = 0; s < ks1; ++s) {
    const RsFrag b = rs_split(xn);
    if (s + 1 < ks1) fetch(s + 1);
    const bf16x8* w = ws.acquire();
    rs_step<NTILES>(acc, w, b);
  }
}

// accumulator tile registers <- per-feature values v[32 tile + 8 q + 4 g + c] (bias vectors, rows of x)
__device__ __forceinline__ void rs_load_tile(f32x16& acc, const float* v, int g) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * g);
    acc[4 * q + 0] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
  }
}

template <int NT, bool PROJ>
__global__ __launch_bounds__(RS_THREADS, 1) void reslayer_split_kernel(const float* x, int64_t ldx, int k_in, float* out,
                                                                       int64_t ldo, int64_t rows, const char* __restrict__ wq,
                                                                       const float* __restrict__ b1,
                                                                       const float* __restrict__ b0) {
  constexpr int NH = NT > 4 ? 2 : 1;            // output halves
  constexpr int NTH = NT / NH;
  extern __shared__ __attribute__((aligned(16))) char s_ring[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int ks1 = (k_in + 15) >> 4;
  const int64_t nblocks = (rows + RS_BLOCK_ROWS - 1) / RS_BLOCK_ROWS;
  const int64_t mine = (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x;       // row blocks of this workgroup

  RsStream<NT, NTH> ws;
  ws.base = wq;
  ws.ring = s_ring;
  ws.ks1 = ks1;
  ws.chunks = ks1 + NH * ((PROJ ? ks1 : 0) + 2 * NT);
  ws.next = 0;
  ws.left = mine * ws.chunks;
  ws.stage = 0;
  ws.wave = wave;
  ws.lane = lane;
  ws.issue(0);
  // the biases live in LDS: as kernel-lifetime registers (where the compiler would hoist them) they cost 16 per tile
  float* s_b1 = reinterpret_cast<float*>(s_ring + 2 * RS_STAGE_BYTES);
  float* s_b0 = s_b1 + 32 * NT;
  for (int i = threadIdx.x; i < 32 * NT; i += RS_THREADS) {
    s_b1[i] = b1[i];
    if (PROJ) s_b0[i] = b0[i];
  }
  __syncthreads();

  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int64_t row = blk * RS_BLOCK_ROWS + wave * 32 + r;
    const bool in = row < rows;
    const int64_t row_c = in ? row : rows - 1;
    const float* xrow = x + row_c * ldx;
    // ---- h^T = relu(W1 x^T + b1) ----------------------------------------------------------------------
    f32x16 h[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) rs_load_tile(h[u], s_b1 + 32 * u, g);
    rs_product_x<NT, NT, NTH>(h, xrow, k_in, ks1, g, ws);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
#pragma unroll
      for (int e = 0; e < 16; ++e) h[u][e] = (h[u][e] < 0.0f) ? 0.0f : h[u][e];        // NaN stays NaN like torch.relu
    }
    // ---- out^T = skip^T + W2 h^T, one half of the output features at a time ------------------------------
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      f32x16 o[NTH];
      if (PROJ) {
#pragma unroll
        for (int u = 0; u < NTH; ++u) rs_load_tile(o[u], s_b0 + 32 * (hf * NTH + u), g);
        rs_product_x<NTH, NT, NTH>(o, xrow, k_in, ks1, g, ws);
      } else {
#pragma unroll
        for (int u = 0; u < NTH; ++u) rs_load_tile(o[u], xrow + 32 * (hf * NTH + u), g);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float hv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            hv[j] = h[t][8 * sp + j];
            // the split is recomputed per output half: kept across halves (what CSE would do) it costs 24 registers per h tile
            if (NH > 1) asm volatile("" : "+v"(hv[j]));
          }
          const RsFrag b = rs_split(hv);
          const bf16x8* w = ws.acquire();
          rs_step<NTH>(o, w, b);
        }
      }
      if (in) {
        float* orow = out + row * ldo + 32 * hf * NTH + 4 * g;
#pragma unroll
        for (int u = 0; u < NTH; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
            v.x = o[u][4 * q + 0]; v.y = o[u][4 * q + 1]; v.z = o[u][4 * q + 2]; v.w = o[u][4 * q + 3];
            *reinterpret_cast<f32x4*>(orow + 32 * u + 8 * q) = v;
          }
        }
      }
    }
  }
}

extern "C" int64_t cppf_reslayer_split_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj) {
  if (k_in <= 0 || n_out <= 0 || (n_out & 31) || n_out > 256) return -1;
  const int nt = n_out / 32, nh = nt > 4 ? 2 : 1, nth = nt / nh;
  if (nt != nth * nh) return -1;
  const int64_t ks1 = (k_in + 15) / 16;
  return ks1 * nt * RS_TILE_BYTES + (int64_t)nh * ((proj ? ks1 : 0) + 2 * nt) * nth * RS_TILE_BYTES;
}

template <int NT, bool PROJ>
static int rs_launch(const float* x, int64_t ldx, int k_in, float* out, int64_t ldo, int64_t rows, const char* wq,
                     const float* b1, const float* b0, int cus, hipStream_t stream) {
  const int lds_bytes = 2 * RS_STAGE_BYTES + 2 * 32 * NT * 4;
  const int64_t nblocks = (rows + RS_BLOCK_ROWS - 1) / RS_BLOCK_ROWS;
  const unsigned grid = (unsigned)(nblocks < cus ? nblocks : cus);
  hipLaunchKernelGGL((reslayer_split_kernel<NT, PROJ>), dim3(grid), dim3(RS_THREADS), lds_bytes, stream, x, ldx, k_in, out, ldo,
                     rows, wq, b1, b0);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

// out[rows, n_out] = skip(x) + relu(x[:, :k_in] W1^T + b1) W2^T with skip = x (b0 == NULL; then k_in == n_out and out may
// be x itself) or x W0^T + b0.  x float32 [rows, >= k_in] with row stride ldx, out float32 with row stride ldo (device;
// 16-byte aligned rows: ldx, ldo multiples of 4); k_in a multiple of 8; n_out in {64, 128, 192, 256}.  wq = the layer's
// weights as the packed split stream (cppf_reslayer_split_stream_bytes bytes; cppf2_amd.models.pack_split documents the
// order).  The second layer's bias is the caller's (carried as a pending offset by cppf2_amd.models.fused_stack).
extern "C" int cppf_reslayer_split(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out,
                                   int64_t rows, const void* wq, int64_t wq_bytes, const float* b1, const float* b0,
                                   void* stream) {
  CPPF_CHECK_ARG(x && out && wq && b1 && rows >= 0);
  CPPF_CHECK_ARG(k_in > 0 && (k_in & 7) == 0 && ldx >= k_in && (ldx & 3) == 0 && (ldo & 3) == 0 && ldo >= n_out);
  CPPF_CHECK_ARG(n_out == 64 || n_out == 128 || n_out == 192 || n_out == 256);
  CPPF_CHECK_ARG(b0 != nullptr || k_in == n_out);
  CPPF_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(k_in, n_out, b0 != nullptr));
  if (rows == 0) return CPPF_OK;
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  CPPF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev & 63] == 0) {
      hipDeviceProp_t prop;
      CPPF_HIP(hipGetDeviceProperties(&prop, dev));
      cus[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
  }
  const int n_cu = cus[dev & 63];
  const char* w = static_cast<const char*>(wq);
  hipStream_t st = (hipStream_t)stream;
  const bool proj = b0 != nullptr;
  switch (n_out / 32) {
    case 2: return proj ? rs_launch<2, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st)
                        : rs_launch<2, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st);
    case 4: return proj ? rs_launch<4, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st)
                        : rs_launch<4, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st);
    case 6: return proj ? rs_launch<6, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st)
                        : rs_launch<6, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st);
    default: return proj ? rs_launch<8, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st)
                         : rs_launch<8, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, n_cu, st);
  }
}
