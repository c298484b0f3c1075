// cppf_mlp_split.hip -- the ResLayers of the tuple / point MLPs (train_shot.py:19-45), one or several per kernel, on the
// bf16 matrix cores, computing on float32 operands by splitting them (exact operands, the six largest of the nine products).
//
//   y = skip(x) + relu(x W1^T + b1) W2^T,     skip(x) = x  (dim_in == dim_out)   or   x W0^T + b0
//   [then identity layers of the same width chained behind it:  y <- y + relu(y W1_l^T + b1_l) W2_l^T]
//
// gfx950 has no xf32 / tf32 path; its f32-input MFMA runs at 1/16 of the bf16 rate.  A float32 value is EXACTLY the sum of
// three bf16 values (8 + 8 + 8 significand bits: hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), round to
// nearest even each time), every bf16 x bf16 product is exact in float32, and the matrix core accumulates in float32.
// So  a * b = (ah + am + al)(bh + bm + bl)  is evaluated as the six products  ah bh + ah bm + am bh + am bm + ah bl + al bh;
// the three dropped ones (am bl, al bm, al bl) are below 2^-24 |a b| together, i.e. under half a unit in the last place of
// the product a float32 FMA chain starts from.  The result is not bit-equal to the f32-input MFMA chain (nor is any
// re-tiled float32 GEMM); its error against a float64 evaluation is of the order of the library float32 GEMMs' -- measured
// side by side: 2.4 x theirs on the tuple MLP's logits (max 1.96e-6 against 8.0e-7 of the largest logit, bench.py
// mlp_error_vs_f64), held to e_split < 3 e_library + 2e-7 by tests/test_mlp_split.py: the dropped terms and the six
// accumulations per product instead of one.  Six bf16 MFMAs replace sixteen f32-input MFMA cycles' worth of work: 2.7x the rate.
// Non-finite operands: a NaN stays a NaN; an infinite x or weight splits into Inf + NaN, so the rows it touches come out NaN
// (a float32 GEMM would give +-Inf where no 0 * Inf occurs); magnitudes above the largest bf16 (3.39e38) behave like Inf,
// and contributions below the float32 normal range may be flushed -- none of which the path's inputs reach (the SHOT
// descriptors are NaN-cleaned before the point encoder, eval.py:215).
//
// Structure (one wavefront = 32 rows of x; a workgroup = 8 wavefronts, two per SIMD with 256 registers each, for widths up
// to 128, and 4 wavefronts, one per SIMD with 512 registers, for 192 / 256; persistent over row blocks):
//   * every product is computed TRANSPOSED (h^T = W1 x^T, y^T = skip^T + W2 h^T) with v_mfma_f32_32x32x16_bf16, so that the
//     rows of x are the COLUMNS of the accumulator tiles (lane l owns row l & 31) and the accumulators of one product are,
//     after bias + ReLU and a split in registers, directly the B operands of the next -- h never leaves the registers, and
//     neither does y between chained layers; the K order of a sum is free, so lane half g = l >> 5 contracts over exactly
//     the features its accumulator registers hold (row (reg & 3) + 8 (reg >> 2) + 4 g of a 32-row tile) and the weights
//     are packed in that order;
//   * a projection layer's x W0^T rides along with x W1^T (same B operand, W0's tiles behind W1's): one pass over x;
//   * x is the B operand of the first product: a wavefront's tile of one K step (32 rows x 16 features) travels
//     global -> LDS by LDS-DMA three K steps ahead of its use and is split in registers ONCE per element (a wavefront
//     covers all output features of its rows);
//   * the A operands (the weights, pre-split on the host into the per-lane fragment order: cppf2_amd.models.pack_split)
//     stream through a two-stage LDS ring by LDS-DMA, shared by the workgroup's wavefronts: the layer's weights are read
//     from L2 once per row block; the pieces of the next chunk are issued between the MFMAs of the current one;
//   * the split of the next K step's B operand and the LDS reads of the next tile's fragments are placed in the shadow of
//     the current tile's six MFMAs; all memory waits of the main loops are counted (vmcnt(N), raw s_barrier): nothing drains
//     the queue while tiles are in flight.
// HBM traffic per row: dim_in + dim_out floats per kernel (+ the residual re-read of an identity first layer: L2 / MALL).
#include "cppf_common.h"
#include <atomic>
#include <mutex>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// wavefronts per workgroup: 8 (two per SIMD, 256 registers each) for layers up to 128 wide, 4 (one per SIMD, 512 registers:
// both accumulator sets of a 192- / 256-wide layer at once) above
__host__ __device__ constexpr int rs_waves(int nt) { return nt >= 6 ? 4 : 8; }
// launch forms of the kernel beyond the ResLayer itself (template argument MODE):
//   RS_LINEAR     y = x W^T + b, a plain nn.Linear (no ReLU, no second product, no skip) over `groups` column blocks of 32 NT
//                 outputs each -- the per-point transforms of the DINO model (train_dino.py:86-87) and the per-point slot tables
//                 below; only the first product of the ResLayer kernel, so 128 accumulator registers: always 8 wavefronts;
//   RS_SUMGATHER  the ResLayer whose input row is [heads | sum-able per-point parts]: the part of x W1^T (and x W0^T) that
//                 belongs to the tuple's points has been evaluated once per POINT and slot (a table [points, slots, 2 x 32 NT],
//                 RS_LINEAR writes it), so the accumulators start from bias + sum_i table[gidx[t, i]][i] and only the head
//                 columns go through the matrix cores.
//   RS_ENCODE     the GATHER launch that also builds the tuple's pair features itself (train_shot.py:75-83): the 30 coordinate
//                 differences and the 10 |n_i . n_j| of the tuple's points -- the first 40 input columns, which the GATHER form
//                 reads from a [rows, 40] array a separate kernel wrote (cppf_encode_tuples_shot_heads: 262 MB per 64 scenes out
//                 and in again) -- are computed by the lanes of a row from pts / normals (L2-resident) with the encode kernel's
//                 arithmetic, bit for bit, and stored straight into the x-tile slots of K steps 0, 1, 2 (with the first 8
//                 descriptor columns, which share K step 2).  Input: the sampler's scene-local indices.
#define RS_RESLAYER 0
#define RS_LINEAR 1
#define RS_SUMGATHER 2
#define RS_ENCODE 3
//   RS_SUMENCODE  RS_SUMGATHER with the head columns built in the kernel like RS_ENCODE: the DINO model's 30 coordinate differences
//                 (train_dino.py:92; zero-padded to 32 columns = K steps 0 and 1) from pts and the sampler's scene-local indices.
#define RS_SUMENCODE 5
// (RS_SUMGATHER keeps the 8-wavefront workgroup of the other 128-wide launches.  Its table loads -- 160 sixteen-byte loads per lane
// and row block, 32 distinct rows per wavefront instruction -- occupy the CU's texture-address unit for ~25 us per 256 rows while
// no wavefront of the CU multiplies: 2.5 ms per launch against 1.85 ms with the loads removed and 2.4 ms with every load an L2 hit,
// i.e. address processing, not memory latency.  Measured alternatives, none faster: two half-size workgroups per CU with and
// without a half-block start-up stagger (2.5 ms: the other workgroup's weight stream queues behind the loads), 4 wavefronts of 512
// registers building the next block's sums in a second accumulator set during the products (2.2 ms without the loads -- one
// wavefront per SIMD costs 20 % -- and 3.2 ms with them: spills).)
__host__ __device__ constexpr int rs_waves_mode(int nt, int mode) { return mode == RS_LINEAR ? 8 : rs_waves(nt); }
__host__ __device__ constexpr int rs_stage_tiles(int mode) { return 16; }     // tile-steps per ring stage
__host__ __device__ constexpr int rs_wgs_per_cu(int mode) { return 1; }
#define RS_FRAG_BYTES 1024                      // one operand fragment: 64 lanes x 8 bf16
// one staged chunk: up to STG tile-steps of three fragments (16: 48 KiB; 8: 24 KiB)
__host__ __device__ constexpr int rs_stage_bytes(int stg) { return stg * 3 * RS_FRAG_BYTES; }
// PC = operand pieces: 3 = bf16 (hi, mid, lo: exact, six products per K step), 2 = fp16 (hi, lo: 22-23 significant bits per
// operand, three products per K step; see the f16x2 section below)
__host__ __device__ constexpr int rs_tile_bytes(int pc) { return pc * RS_FRAG_BYTES; }

// measured variant kept as a switch: LDS fragments requested two tiles ahead instead of one (no faster, 12 registers more).
// (A hand-placed MFMA / filler interleave -- one MFMA, then a few independent instructions; or everything in front of each tile's
// back-to-back MFMA chain -- was measured in round 3 and removed from the source in round 6: docs/experiments/r6_removed_mlp_variants.patch.)
// Round 3 measurements (scratch/rs/: rs_probe.hip with the RS_DBG switches, filler_price.hip, mfma_chain.hip + PMC): the kernel is
// bound by the chip's POWER limit, not by issue slots or latency -- dependent MFMA chains issue every 32.2 cycles, up to four plain
// VALU instructions fit between two MFMAs for free in cycles, and removing the chunk barrier raises the matrix pipe's busy
// fraction by 5 points while the clock drops by the same factor (same wall time); what moves the time is the work executed
// beside the MFMAs (each VALU instruction per MFMA costs ~3 % through the clock, a ds_read_b128 ~5 %): without the operand
// split the 256-wide kernel runs 14 % faster, without the weight stream 9 %, without its global loads / stores 8 %.  Hence
// the compiler's own order within a tile region (fewer register moves, 5 % faster on the 256-wide identity kernel, equal elsewhere).
// Also measured, not kept: two tiles per region with their MFMA chains alternating between the two accumulators (3-8 % slower),
// output halves / quarters for the 256-wide layers at two wavefronts per SIMD (equal), register-staged instead of LDS-DMA weight
// chunks, the residual of the 256-wide identity kernel loaded before the first product (RS_EARLY_RESIDUAL: 128 registers too many).
#ifndef RS_DEEP_PREFETCH
#define RS_DEEP_PREFETCH 0
#endif
// timing probes (scratch/rs/rs_probe.hip; results are wrong with any bit set): 1 no weight pieces, 2 no operand split,
// 4 no LDS fragment reads, 8 no chunk barrier, 16 no x tiles, 32 no residual load / output store
#ifndef RS_DBG
#define RS_DBG 0
#endif
#ifndef RS_EXACT_PMAX
#define RS_EXACT_PMAX 1
#endif
#ifndef RS_XCD_BLOCKS
#define RS_XCD_BLOCKS 1          // 0: row blocks dealt round-robin over all workgroups (rounds 2-3)
#endif
#ifndef RS_EARLY_RESIDUAL
#define RS_EARLY_RESIDUAL 0      // 1: the wide identity kernel then needs 128 registers more than there are (spills)
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// K steps per staged chunk (one barrier per chunk): as many as fit a stage
__host__ __device__ constexpr int rs_spc(int tiles, int stg = 16) { return stg / tiles >= 4 ? 4 : stg / tiles >= 2 ? 2 : 1; }

struct RsFrag {                                  // one operand fragment per slice, as raw dwords (2 bf16 each)
  unsigned h[4], m[4], l[4];
};

// exact three-way split of two floats into packed bf16 pairs (each step rounds to nearest even; the remainders are
// exact in float32): dword p of a fragment's (hi, mid, lo)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int PC, bool PIN = false>
__device__ __forceinline__ void rs_split_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  // PIN: an opaque copy of the inputs keeps the arithmetic where it is written (between two tiles' MFMAs); without it the
  // compiler gathers the splits of a whole chunk in front of the chunk's first MFMA (12 registers per K step, no overlap)
  if (PIN) asm volatile("" : "+v"(v0), "+v"(v1));
  if (PIN && (RS_DBG & 2)) {
    h = __builtin_bit_cast(unsigned, v0);
    m = __builtin_bit_cast(unsigned, v1);
    l = h ^ m;
    return;
  }
  if (PC == 2) {
    // fp16 pair: hi = RNE(v) (11 significant bits), lo = RNE(v - hi) (the next 11-12): v_cvt_pk_f16_f32, two v_fma_mix_f32 /
    // v_cvt_f32_f16 + v_sub_f32, v_cvt_pk_f16_f32; subnormal results are kept (the matrix core honours them)
    const f32x2 v = {v0, v1};
    const f16x2 hb = __builtin_convertvector(v, f16x2);
    const float r0 = v0 - (float)hb[0], r1 = v1 - (float)hb[1];
    const f32x2 rv = {r0, r1};
    h = __builtin_bit_cast(unsigned, hb);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, f16x2));
    m = 0;
    if (PIN) asm volatile("" : "+v"(h), "+v"(l));
    return;
  }
  // scalar float subtractions on purpose (and -fno-slp-vectorize for this file, cppf2_amd/build.py): a v_pk_add_f32 between two
  // MFMAs costs 16 issue cycles of the matrix pipe (scratch/rs/filler_price.hip), a v_sub_f32 none
  const f32x2 v = {v0, v1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
  const f32x2 rv = {r0, r1};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, bf16x2));
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  const f32x2 qv = {q0, q1};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(qv, bf16x2));
  if (PIN) asm volatile("" : "+v"(h), "+v"(m), "+v"(l));      // ... and from sinking the arithmetic down to its first use
}

template <int PC>
__device__ __forceinline__ RsFrag rs_split(const float (&v)[8]) {
  RsFrag f;
#pragma unroll
  for (int p = 0; p < 4; ++p) rs_split_pair<PC>(v[2 * p], v[2 * p + 1], f.h[p], f.m[p], f.l[p]);
  return f;
}

template <int PC>
__device__ __forceinline__ f32x16 rs_mfma(const unsigned (&a)[4], const unsigned (&b)[4], const f32x16& c) {
  const u32x4 av = {a[0], a[1], a[2], a[3]}, bv = {b[0], b[1], b[2], b[3]};
  if (PC == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
}

// acc += A B over one K step of 16 with both operands split, smallest terms first: six exact-product MFMAs (bf16 triples), or
// the three products hi lo + lo hi + hi hi of the fp16 pairs (lo lo is below 2^-22 of the product and dropped)
template <int PC>
__device__ __forceinline__ void rs_mma(f32x16& acc, const RsFrag& a, const RsFrag& b) {
  acc = rs_mfma<PC>(a.l, b.h, acc);
  acc = rs_mfma<PC>(a.h, b.l, acc);
  if (PC == 3) {
    acc = rs_mfma<PC>(a.m, b.m, acc);
    acc = rs_mfma<PC>(a.m, b.h, acc);
    acc = rs_mfma<PC>(a.h, b.m, acc);
  }
  acc = rs_mfma<PC>(a.h, b.h, acc);
}

// s_waitcnt immediate (gfx9 encoding): vmcnt (6 bits, split 4 + 2), lgkmcnt (4 bits); expcnt not waited on
__host__ __device__ constexpr int rs_waitcnt(int vm, int lgkm) {
  return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8);
}
#define RS_WAIT(vm, lgkm) __builtin_amdgcn_s_waitcnt(rs_waitcnt(vm, lgkm))

// The weight stream of one kernel, consumed in order by every wavefront of the workgroup (cppf2_amd.models.pack_split):
// segment 0 = ks1 K steps of T0 tiles (W1; for a projection layer W1 and W0 side by side), then W2 (2 NT steps of NT
// tiles), then W1, W2 of each chained layer (2 NT steps of NT tiles each).  It moves through a two-stage LDS ring by LDS-DMA in chunks of up to rs_spc(tiles) K steps of one segment: the
// pieces of chunk c + 1 (1 KiB per wavefront instruction) are issued between the MFMAs of the first K step of chunk c.
// All memory waits of the kernel's main loops are COUNTED (the vector-memory queue completes in order): see acquire().
template <int NT, int T0, int WAVES, int PC, bool LIN = false, int STG = 16>
struct RsStream {
  static constexpr int STG_ = STG, SPC0 = rs_spc(T0, STG), SPC1 = rs_spc(NT, STG), STAGE_BYTES = rs_stage_bytes(STG);
  // pieces of a chunk per wavefront, at most (a chunk = rs_spc(tiles) K steps of `tiles` tiles, three fragments each)
  static constexpr int P0 = (SPC0 * T0 * PC + WAVES - 1) / WAVES, P1 = (SPC1 * NT * PC + WAVES - 1) / WAVES;
  static constexpr int PMAX = RS_EXACT_PMAX ? (P0 > P1 ? P0 : P1) : 48 / WAVES;
  int nseg;                    // 2 + 2 per chained identity layer
  int chunks;                  // chunks per row block
  const char* base;            // packed stream in global memory
  char* ring;                  // LDS, two stages
  int ks1;
  int seg, pos;                // segment / K step within it the next chunk starts at
  int64_t off;                 // its byte offset
  int64_t left;                // chunks still to fetch over the remaining row blocks of this workgroup
  int stage;                   // stage the next acquire() returns
  int wave, lane;
  const char* p_src;           // the chunk being fetched (this lane's 16 bytes of its piece 0), its stage, its last piece
  char* p_dst;
  int p_last;

  __device__ __forceinline__ void shape(int chain) {
    if (LIN) {                                   // `chain` column groups, each one first-product segment (ks1 steps of T0 tiles)
      nseg = chain;
      chunks = chain * ((ks1 + SPC0 - 1) / SPC0);
      return;
    }
    nseg = 2 + 2 * chain;
    chunks = (ks1 + SPC0 - 1) / SPC0 + (1 + 2 * chain) * (2 * NT / SPC1);
  }
  // selects the next chunk of the stream (destination: stage st); its pieces are then issued one by one with piece()
  __device__ __forceinline__ void plan(int st) {
    p_dst = ring + st * STAGE_BYTES;
    if (left <= 0) {                             // past the end: piece() re-reads the stream's first KiB into the idle stage
      p_src = base + lane * 16;
      p_last = 0;
      return;
    }
    const bool first = LIN || seg == 0;
    const int tiles = first ? T0 : NT;
    const int spc = first ? SPC0 : SPC1;
    const int steps = first ? ks1 : 2 * NT;
    const int ns = steps - pos < spc ? steps - pos : spc;
    const int pieces = ns * tiles * PC;
    p_src = base + off + lane * 16;
    p_last = pieces - 1;
    off += (int64_t)pieces * RS_FRAG_BYTES;
    pos += ns;
    if (pos == steps) {
      pos = 0;
      if (++seg == nseg) {
        seg = 0;
        off = 0;
      }
    }
    --left;
  }
  // the q-th piece of this wavefront (pieces wave, wave + WAVES, ...; at most PMAX per chunk).  Branch-free: past the
  // chunk's last piece it re-issues that one (same bytes to the same place), which also keeps the number of
  // vector-memory operations per chunk the same for every wavefront.
  __device__ __forceinline__ void piece(int q) {
    if (RS_DBG & 1) return;
    int j = wave + q * WAVES;
    j = j < p_last ? j : p_last;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_src + j * RS_FRAG_BYTES),
                                     (__attribute__((address_space(3))) void*)(p_dst + j * RS_FRAG_BYTES), 16, 0, 0);
  }
  __device__ __forceinline__ void prime() {                // the very first chunk: all pieces at once
    plan(0);
#pragma unroll
    for (int q = 0; q < PMAX; ++q) piece(q);
  }
  // The stage holding the next chunk in stream order.  Its pieces were issued during the first K step of the previous
  // chunk; `younger` = the vector-memory operations this wavefront has issued since (the x tiles of the K steps after
  // that one): the queue completes in order, so vmcnt(younger) means "my pieces have landed".  After the barrier every
  // wavefront's pieces have, and every wavefront is done reading the other stage, which plan() then hands to the next
  // chunk.  (A raw barrier: __syncthreads() would drain the whole queue, the x tiles in flight included.)
  template <int YOUNGER>
  __device__ __forceinline__ const u32x4* acquire() {
    RS_WAIT(YOUNGER, 0);
    if (!(RS_DBG & 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int cur = stage;
    stage ^= 1;
    plan(stage);
    return reinterpret_cast<const u32x4*>(ring + cur * STAGE_BYTES) + lane;
  }
};

template <int PC>
__device__ __forceinline__ RsFrag rs_read(const u32x4* w, int tile) {
  RsFrag a;
  if (RS_DBG & 4) tile = 0;
  const u32x4 h = w[(tile * PC + 0) * 64], l = w[(tile * PC + PC - 1) * 64];
  const u32x4 m = (PC == 3) ? w[(tile * PC + 1) * 64] : h;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    a.h[p] = h[p];
    a.m[p] = m[p];
    a.l[p] = l[p];
  }
  return a;
}

// one K step: acc[u] += W[tile u] b for the NTILES tiles at w.  The fragments of tile u + 1 are requested from LDS before
// the six MFMAs of tile u issue (their 192 cycles cover the read); `filler(p)`, p < 4, is independent vector work (the split
// of the NEXT step's B operand, a quarter per call) and `dma(q)`, q < PMAX, the LDS-DMA issue of one piece of the next
// weight chunk: both are placed in front of a tile's MFMAs so that they issue in their shadow; the scheduling barrier
// per tile keeps the compiler from hoisting all the reads / all the filler to the step's front.
template <int NTILES, bool PREFETCH, int PMAX, int PC, class F, class D>
__device__ __forceinline__ void rs_step(f32x16 (&acc)[NTILES], const u32x4* w, const RsFrag& b, F&& filler, D&& dma) {
  // PREFETCH: fragments are requested two tiles ahead (one wavefront per SIMD: nobody else covers the LDS latency);
  // otherwise one tile ahead
  RsFrag a0 = rs_read<PC>(w, 0), a1 = a0;
  if (PREFETCH && NTILES > 1) a1 = rs_read<PC>(w, 1);
#pragma unroll
  for (int u = 0; u < NTILES; ++u) {
    RsFrag a2 = a1;
    if (PREFETCH) {
      if (u + 2 < NTILES) a2 = rs_read<PC>(w, u + 2);
    } else {
      if (u + 1 < NTILES) a1 = rs_read<PC>(w, u + 1);
    }
#pragma unroll
    for (int q = (u * PMAX + NTILES - 1) / NTILES; q < ((u + 1) * PMAX + NTILES - 1) / NTILES; ++q) dma(q);
#pragma unroll
    for (int p = (u * 4) / NTILES; p < ((u + 1) * 4) / NTILES; ++p) filler(p);
    rs_mma<PC>(acc[u], a0, b);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a1;
    if (PREFETCH) a1 = a2;
  }
}

// The x tile of one K step of a wavefront's 32 rows -- lane (r, g): features 16 s + 8 g + (0..7) of row r, 32 bytes --
// travels global -> LDS by two LDS-DMA instructions (16 bytes per lane each) into one of three 2 KiB slots of the
// wavefront, three K steps ahead of its use; nothing of it is tracked by the compiler, the waits are counted by hand.
// GATHER (the tuple inputs of the SHOT model, train_shot.py:75-83, never materialised): features [0, head) of a row come
// from its block of pair features (xrow, `head` floats per tuple), feature head + fdim j + c from table[gi[j]][c] -- the
// per-point descriptor of the tuple's j-th point; 8 consecutive features never straddle two sources (head, fdim % 8 == 0).
// what varies per row block: this lane's row (GATHER: its head block) and, for GATHER, its tuple's global point indices --
// passed BY VALUE everywhere (scalars, not an array, and never behind a reference: selected by a lane-dependent j, anything
// addressable would be kept in scratch memory and the selection turned into an indexed scratch load)
struct RsRow {
  const float* xrow;
  int g0, g1, g2, g3, g4, g5, g6, g7;
};

template <bool GATHER>
struct RsX {
  char* slots;                 // the wavefront's three slots in LDS
  int k_in, g, lane;
  const float* table;          // GATHER: [points, fdim] descriptors, head width, log2(fdim)
  int head, fshift;

  __device__ __forceinline__ const float* source(int f, const RsRow rw) const {
    if (!GATHER || f < head) return rw.xrow + f;
    const int j = (f - head) >> fshift, c = (f - head) & ((1 << fshift) - 1);
    int p = rw.g0;
    p = j == 1 ? rw.g1 : p; p = j == 2 ? rw.g2 : p; p = j == 3 ? rw.g3 : p; p = j == 4 ? rw.g4 : p;
    p = j == 5 ? rw.g5 : p; p = j == 6 ? rw.g6 : p; p = j == 7 ? rw.g7 : p;
    return table + ((int64_t)p << fshift) + c;
  }
  __device__ __forceinline__ void issue(int s, int slot, const RsRow rw) const {   // always exactly two vector-memory operations
    if (RS_DBG & 16) return;
    int f = 16 * s + 8 * g;
    if (f + 8 > k_in) f = 0;                     // past the end (or the zero tail): any valid address, the value is masked
    const float* src = source(f, rw);
    char* dst = slots + slot * 2048;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4),
                                     (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void read(float (&v)[8], int s, int slot) const {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(slots + slot * 2048 + lane * 16);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(slots + slot * 2048 + 1024 + lane * 16);
    const bool ok = 16 * s + 8 * g + 8 <= k_in;
    v[0] = ok ? v0.x : 0.0f; v[1] = ok ? v0.y : 0.0f; v[2] = ok ? v0.z : 0.0f; v[3] = ok ? v0.w : 0.0f;
    v[4] = ok ? v1.x : 0.0f; v[5] = ok ? v1.y : 0.0f; v[6] = ok ? v1.z : 0.0f; v[7] = ok ? v1.w : 0.0f;
  }
};

// acc[u] (u < NTILES) += W[tile u] x^T over all K steps.  On entry the x tiles of steps 0, 1, 2 have been issued into
// slots 0, 1, 2 (by the caller, possibly a whole row block earlier) and nothing younger than them is in flight except
// weight pieces.  Step s: wait for tile s + 1 (two tiles younger: vmcnt(4) before, vmcnt(2) after the issue of tile
// s + 3 -- the wait comes first), read it, issue tile s + 3 into the slot tile s left, split tile s + 1 in the shadow of
// step s's MFMAs.
template <int NTILES, bool PREFETCH, int PC, class X, class Stream>
__device__ __forceinline__ void rs_product_x(f32x16 (&acc)[NTILES], const X xs, const RsRow rw, int ks1, Stream& ws) {
  constexpr int SPC = rs_spc(NTILES, Stream::STG_);
  float xv[8];
  RS_WAIT(4, 15);                                    // tile 0 has landed (tiles 1, 2 may be in flight)
  __builtin_amdgcn_sched_barrier(0);
  xs.read(xv, 0, 0);
  RsFrag b = rs_split<PC>(xv);
  int slot = 1;                                      // slot of tile s + 1
  bool whole = true;                                 // the previous chunk had SPC steps (counted wait) or this is the first
  for (int s0 = 0; s0 < ks1; s0 += SPC) {
    // weight pieces of this chunk: issued in the first step of the previous chunk, 2 (SPC - 1) x-tile operations ago --
    // unless that chunk was not this loop's (first iteration: other phases issue no x tiles after their pieces)
    const u32x4* w = (s0 > 0 && whole) ? ws.template acquire<2 * (SPC - 1)>() : ws.template acquire<0>();
    whole = s0 + SPC <= ks1;
#pragma unroll
    for (int i = 0; i < SPC; ++i) {
      if (s0 + i < ks1) {
        RS_WAIT(2, 15);                              // x tile s + 1 has landed (tile s + 2 may be in flight)
        __builtin_amdgcn_sched_barrier(0);
        xs.read(xv, s0 + i + 1, slot);
        xs.issue(s0 + i + 3, slot == 0 ? 2 : slot - 1, rw);
        slot = slot == 2 ? 0 : slot + 1;
        RsFrag bn;
        rs_step<NTILES, PREFETCH, Stream::PMAX, PC>(acc, w + i * NTILES * PC * 64, b,
                                  [&](int p) { rs_split_pair<PC, true>(xv[2 * p], xv[2 * p + 1], bn.h[p], bn.m[p], bn.l[p]); },
                                  [&](int q) { if (i == 0) ws.piece(q); });
        b = bn;
      }
    }
  }
}

// dst[u] (u < NTILES) += W[tile u] src^T where src is NS accumulator tiles of this lane's row (the output of a previous
// product): step (t, sp) contracts over the lane's registers 8 sp .. 8 sp + 7 of src tile t -- the weights are packed in
// that feature order.  The split of step + 1 runs in the shadow of step's MFMAs (pinned there, which also keeps CSE from
// keeping a once-computed split alive across products at 24 registers per tile).
// (bs: the factor the source tiles carry and the B operand must not -- 1 for the bf16 triples, 1 / weight scale for the fp16 pairs)
template <int NS, int NTILES, bool PREFETCH, int PC, class Stream>
__device__ __forceinline__ void rs_product_h(f32x16 (&dst)[NTILES], const f32x16 (&src)[NS], Stream& ws, float bs) {
  constexpr int SPC = rs_spc(NTILES, Stream::STG_);
  static_assert((2 * NS) % SPC == 0, "whole chunks");
  RsFrag b;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    rs_split_pair<PC>(PC == 2 ? src[0][2 * p] * bs : src[0][2 * p], PC == 2 ? src[0][2 * p + 1] * bs : src[0][2 * p + 1], b.h[p], b.m[p], b.l[p]);
#pragma unroll
  for (int c = 0; c < 2 * NS / SPC; ++c) {
    const u32x4* w = ws.template acquire<0>();        // no x tiles in flight here (or old ones: harmless)
#pragma unroll
    for (int i = 0; i < SPC; ++i) {
      const int step = c * SPC + i, nx = (step + 1 < 2 * NS) ? step + 1 : step;
      RsFrag bn = b;
      rs_step<NTILES, PREFETCH, Stream::PMAX, PC>(dst, w + i * NTILES * PC * 64, b,
                                [&](int p) {
                                  if (step + 1 < 2 * NS) {
                                    const float s0 = src[nx >> 1][8 * (nx & 1) + 2 * p], s1 = src[nx >> 1][8 * (nx & 1) + 2 * p + 1];
                                    rs_split_pair<PC, true>(PC == 2 ? s0 * bs : s0, PC == 2 ? s1 * bs : s1, bn.h[p], bn.m[p], bn.l[p]);
                                  }
                                },
                                [&](int q) { if (i == 0) ws.piece(q); });
      b = bn;
    }
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

// the same with every value multiplied by sc (f16x2: the residual enters the accumulators in the weight-scaled domain)
__device__ __forceinline__ void rs_load_tile_scaled(f32x16& acc, const float* v, int g, float sc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * g);
    acc[4 * q + 0] = t.x * sc; acc[4 * q + 1] = t.y * sc; acc[4 * q + 2] = t.z * sc; acc[4 * q + 3] = t.w * sc;
  }
}

// The bin draw of eval.py:225-229 as the epilogue of the logit head's output layer (6 coordinates x 32 bins = the 6 output
// tiles): tile u holds the logits of coordinate u of this lane's row, registers e <-> bins (e & 3) + 8 (e >> 2) + 4 g, the other
// 16 bins sit in lane ^ 32.  Same arithmetic as decode_bins_kernel<32> (cppf_core.hip), bit for bit: logits (+ prior), max,
// softmax_exp, a float32 running sum IN BIN ORDER -- so the running total alternates between the two lanes every four bins --
// target = uniform * total, bin = #{cdf <= target} capped at 31.  The six coordinates' chains are independent and interleave.
struct RsDecode {
  const float* prior = nullptr;     // [rows, 192] additive logit prior or NULL
  // ... or the prior GENERATED here: -0.5 ((k - prior_pos[row, c]) * prior_inv_sigma)^2 for bin k of coordinate c (a Gaussian
  // bump in logit space around a per-coordinate bin position: 24 bytes per row read instead of 768; float32 operations in this
  // order, no contraction -- the values of the array a caller would build with the same three operations, bit for bit)
  const float* prior_pos = nullptr; // [rows, 6] or NULL
  float prior_inv_sigma = 0.0f;
  const float* uniforms = nullptr;  // [rows, 6]
  int32_t* bins = nullptr;          // [rows, 6] out
};

template <int NT, int PC>
__device__ __forceinline__ void rs_decode_epilogue(f32x16 (&o)[NT], const RsDecode& dc, int64_t row, bool in, int g, float bs) {
  static_assert(NT == 6, "6 coordinates x 32 bins");
  if (PC == 2) {                                 // back to the true scale (f16x2: the accumulators carry the weight scale)
#pragma unroll
    for (int u = 0; u < 6; ++u) o[u] *= bs;
  }
  if (dc.prior) {
    const float* pr = dc.prior + row * 192 + 4 * g;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pr + 32 * u + 8 * q);
        o[u][4 * q + 0] += v.x; o[u][4 * q + 1] += v.y; o[u][4 * q + 2] += v.z; o[u][4 * q + 3] += v.w;
      }
    }
  } else if (dc.prior_pos) {
    const float* pp = dc.prior_pos + row * 6;
    const float is = dc.prior_inv_sigma;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const float pos = pp[u];
#pragma unroll
      for (int e = 0; e < 16; ++e) {               // register e <-> bin (e & 3) + 8 (e >> 2) + 4 g
        const float z = ((float)((e & 3) + 8 * (e >> 2) + 4 * g) - pos) * is;
        o[u][e] += (z * z) * -0.5f;
      }
    }
  }
  float back[6], tot[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    float m = o[u][0];
#pragma unroll
    for (int e = 1; e < 16; ++e) m = fmaxf(m, o[u][e]);
    m = fmaxf(m, __shfl_xor(m, 32));
#pragma unroll
    for (int e = 0; e < 16; ++e) o[u][e] = softmax_exp(o[u][e] - m);
    back[u] = 0.0f;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float ta[6][4], tb[6][4], fwd[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {                // lanes g = 0: bins 8 q .. 8 q + 3 continue the total the partner sent back
      ta[u][0] = back[u] + o[u][4 * q + 0];
      ta[u][1] = ta[u][0] + o[u][4 * q + 1];
      ta[u][2] = ta[u][1] + o[u][4 * q + 2];
      ta[u][3] = ta[u][2] + o[u][4 * q + 3];
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) fwd[u] = __shfl_xor(ta[u][3], 32);
#pragma unroll
    for (int u = 0; u < 6; ++u) {                // lanes g = 1: bins 8 q + 4 .. 8 q + 7 continue from there
      tb[u][0] = fwd[u] + o[u][4 * q + 0];
      tb[u][1] = tb[u][0] + o[u][4 * q + 1];
      tb[u][2] = tb[u][1] + o[u][4 * q + 2];
      tb[u][3] = tb[u][2] + o[u][4 * q + 3];
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      back[u] = __shfl_xor(tb[u][3], 32);
      tot[u] = g ? tb[u][3] : back[u];           // after q = 3: the total, on both lanes
#pragma unroll
      for (int i = 0; i < 4; ++i) o[u][4 * q + i] = g ? tb[u][i] : ta[u][i];   // this lane's CDF values
    }
  }
  const float* ur = dc.uniforms + row * 6;
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const float target = ur[u] * tot[u];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) cnt += (o[u][e] <= target) ? 1 : 0;
    cnt += __shfl_xor(cnt, 32);
    if (in && g == 0) dc.bins[row * 6 + u] = cnt < 31 ? cnt : 31;
  }
}

// the claimed-row-block slots in LDS (dynamic scheduling): workgroup-scope relaxed atomics -- ordered by the workgroup barriers
// between the store and the loads, and not movable across them by the compiler
__device__ __forceinline__ void rs_claim_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int rs_claim_load(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct RsTap {                  // optional second output: the activation after the FIRST layer of a chained launch
  float* out = nullptr;         // [rows, >= n_out] or NULL
  int64_t ld = 0;
};

struct RsGather {               // GATHER launches: see RsX; RS_SUMGATHER: the per-point slot tables
  const int32_t* gidx = nullptr;   // [rows, slots] global point indices (RS_ENCODE: scene-local ones, as the sampler writes them)
  const float* table = nullptr;    // GATHER: [points, 1 << fshift]; RS_SUMGATHER: [points, slots, 2 x 32 NT] (row pitch tld floats)
  int slots = 0, head = 0, fshift = 0;
  int64_t tld = 0;
  // RS_ENCODE: the cloud's points and normals [points, 3], the scenes' point / tuple offsets [B + 1]
  const float* pts = nullptr;
  const float* nrm = nullptr;
  const int32_t* pt_off = nullptr;
  const int32_t* tup_off = nullptr;
  int B = 0;
  // dynamic row-block scheduling (nullptr: static partition): int32[2] = {claims, finished workgroups}, zero on entry and on exit
  int* sched = nullptr;
};

struct RsRow3 {                  // a 12-byte row of a [points, 3] array (4-byte aligned): one load instruction
  float x, y, z;
};

// RS_ENCODE: the x tiles of K steps 0, 1, 2 of this lane's row -- input columns 0 .. 47 = [30 coordinate differences p_i - p_j over
// the pairs (i, j) in itertools.combinations order | 10 max(n_i . n_j, -(n_i . n_j)) | the first 8 descriptor columns of point 0]
// -- computed and stored into the wavefront's three slots in the layout the LDS-DMA form leaves there (lane (r, g) holds columns
// 16 s + 8 g + 0..7 of K step s: two 16-byte halves 1 KiB apart).  Arithmetic = encode_shot_heads_tile_kernel's (cppf_core.hip):
// plain float32 subtractions; (x x' + y y') + z z' without contraction; fmaxf(s, -s).
__device__ __forceinline__ void rs_encode_tiles(char* slots, int lane, int g, const float* __restrict__ pts, const float* __restrict__ nrm,
                                                const float* __restrict__ table, int fshift, int g0, int g1, int g2, int g3, int g4) {
  constexpr int PI_[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, PJ_[10] = {1, 2, 3, 4, 2, 3, 4, 3, 4, 4};
  const int gi[5] = {g0, g1, g2, g3, g4};
  float v[48];
  {
    float p[5][3];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const RsRow3 r = *reinterpret_cast<const RsRow3*>(pts + 3 * (int64_t)gi[q]);
      p[q][0] = r.x; p[q][1] = r.y; p[q][2] = r.z;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 * q + c] = p[PI_[q]][c] - p[PJ_[q]][c];
  }
  {
    float n[5][3];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const RsRow3 r = *reinterpret_cast<const RsRow3*>(nrm + 3 * (int64_t)gi[q]);
      n[q][0] = r.x; n[q][1] = r.y; n[q][2] = r.z;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const float* ni = n[PI_[q]];
      const float* nj = n[PJ_[q]];
      const float s_ = (ni[0] * nj[0] + ni[1] * nj[1]) + ni[2] * nj[2];
      v[30 + q] = fmaxf(s_, -s_);
    }
  }
  {
    const float* d0 = table + ((int64_t)g0 << fshift);
    const f32x4 a = *reinterpret_cast<const f32x4*>(d0), b = *reinterpret_cast<const f32x4*>(d0 + 4);
    v[40] = a.x; v[41] = a.y; v[42] = a.z; v[43] = a.w; v[44] = b.x; v[45] = b.y; v[46] = b.z; v[47] = b.w;
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f32x4 lo, hi;
    lo.x = g ? v[16 * s + 8] : v[16 * s + 0]; lo.y = g ? v[16 * s + 9] : v[16 * s + 1];
    lo.z = g ? v[16 * s + 10] : v[16 * s + 2]; lo.w = g ? v[16 * s + 11] : v[16 * s + 3];
    hi.x = g ? v[16 * s + 12] : v[16 * s + 4]; hi.y = g ? v[16 * s + 13] : v[16 * s + 5];
    hi.z = g ? v[16 * s + 14] : v[16 * s + 6]; hi.w = g ? v[16 * s + 15] : v[16 * s + 7];
    *reinterpret_cast<f32x4*>(slots + s * 2048 + lane * 16) = lo;
    *reinterpret_cast<f32x4*>(slots + s * 2048 + 1024 + lane * 16) = hi;
  }
}

// RS_SUMENCODE: the x tiles of K steps 0 and 1 -- input columns 0 .. 31 = [30 coordinate differences p_i - p_j in combinations order |
// 0 0] (cppf_encode_tuples_coord_heads' columns, bit for bit: plain float32 subtractions).
__device__ __forceinline__ void rs_encode_coord_tiles(char* slots, int lane, int g, const float* __restrict__ pts, int g0, int g1, int g2,
                                                      int g3, int g4) {
  constexpr int PI_[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, PJ_[10] = {1, 2, 3, 4, 2, 3, 4, 3, 4, 4};
  const int gi[5] = {g0, g1, g2, g3, g4};
  float p[5][3], v[32];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const RsRow3 r = *reinterpret_cast<const RsRow3*>(pts + 3 * (int64_t)gi[q]);
    p[q][0] = r.x; p[q][1] = r.y; p[q][2] = r.z;
  }
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) v[3 * q + c] = p[PI_[q]][c] - p[PJ_[q]][c];
  v[30] = 0.0f; v[31] = 0.0f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f32x4 lo, hi;
    lo.x = g ? v[16 * s + 8] : v[16 * s + 0]; lo.y = g ? v[16 * s + 9] : v[16 * s + 1];
    lo.z = g ? v[16 * s + 10] : v[16 * s + 2]; lo.w = g ? v[16 * s + 11] : v[16 * s + 3];
    hi.x = g ? v[16 * s + 12] : v[16 * s + 4]; hi.y = g ? v[16 * s + 13] : v[16 * s + 5];
    hi.z = g ? v[16 * s + 14] : v[16 * s + 6]; hi.w = g ? v[16 * s + 15] : v[16 * s + 7];
    *reinterpret_cast<f32x4*>(slots + s * 2048 + lane * 16) = lo;
    *reinterpret_cast<f32x4*>(slots + s * 2048 + 1024 + lane * 16) = hi;
  }
}

// f16x2 (PC == 2, CPPF_MLP_ARITH=split16): every float32 operand as an fp16 pair hi + lo (22-23 significant bits), three
// products per K step, float32 accumulate.  fp16 has five exponent bits, so the weights are multiplied by a power of two
// `wscale` before they are split (the host picks it so that the largest weight of the launch sits near 2^13: the lo pieces stay
// normal) and the biases come pre-multiplied by it: every accumulator then holds wscale x its true value -- the residual stream
// included, so the skip additions need no rescaling -- and only the B operand of the next product (and the stored outputs) are
// multiplied by 1 / wscale (exact).  Activations are split unscaled: |x| < 65504 is required, subnormal lo pieces are honoured
// by the matrix core (absolute resolution 2^-25 for small activations).
template <int NT, bool PROJ, bool GATHER, bool DECODE = false, int PC = 3, int MODE = RS_RESLAYER>
__global__ __launch_bounds__(64 * rs_waves_mode(NT, MODE), rs_wgs_per_cu(MODE)) void reslayer_split_kernel(const float* x, int64_t ldx, int k_in, float* out,
                                                                              int64_t ldo, int64_t rows,
                                                                              const char* __restrict__ wq,
                                                                              const float* __restrict__ b1,
                                                                              const float* __restrict__ b0, int chain,
                                                                              RsGather ga, RsDecode dc, RsTap tap, float wscale) {
  const float bs = (PC == 2) ? 1.0f / wscale : 1.0f;        // B-operand / output factor (a power of two: exact)
  constexpr int WAVES = rs_waves_mode(NT, MODE), THREADS = 64 * WAVES, BLOCK_ROWS = 32 * WAVES;
  constexpr int T0 = PROJ ? 2 * NT : NT;        // tiles of the first product: W1 [and W0 behind it]
  constexpr bool LIN = MODE == RS_LINEAR;
  static_assert(!LIN || (!PROJ && !GATHER && !DECODE), "a plain Linear has one product");
  constexpr bool SUMG = MODE == RS_SUMGATHER || MODE == RS_SUMENCODE;      // the per-point parts come from slot tables
  constexpr bool LOCAL_IDX = MODE == RS_ENCODE || MODE == RS_SUMENCODE;    // gidx holds the sampler's scene-local indices
  static_assert(!SUMG || (PROJ && !GATHER), "the table sums stand for columns of a projection layer's input");
  static_assert(MODE != RS_ENCODE || (GATHER && !DECODE), "RS_ENCODE is a form of the gathering launch");
  constexpr bool PF = RS_DEEP_PREFETCH && WAVES == 4;   // LDS read-ahead: two tiles or (measured no slower) one
  extern __shared__ __attribute__((aligned(16))) char s_ring[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform: keeps its derived addresses scalar
  const int r = lane & 31, g = lane >> 5;
  const int ks1 = (k_in + 15) >> 4;
  const int64_t nblocks = (rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  // Row blocks of this workgroup: bfirst, bfirst + bstride, ... < bend.  Workgroup w runs on XCD w & 7 (round-robin dispatch, eight
  // L2 caches): with a grid that is a multiple of 8, XCD j takes the j-th eighth of the row blocks and its workgroups sweep it side
  // by side, so that the rows in flight on one XCD are neighbours -- the tuples of ONE scene, whose per-point table (1 MB for the
  // SHOT model) then stays in that XCD's L2 instead of being fetched by all eight.  Other grids: plain round-robin.
  int64_t bfirst = blockIdx.x, bstride = gridDim.x, bend = nblocks;
  if (RS_XCD_BLOCKS && (gridDim.x & 7) == 0 && nblocks >= 64) {
    const int64_t xcd = blockIdx.x & 7;
    bfirst = nblocks * xcd / 8 + (blockIdx.x >> 3);
    bend = nblocks * (xcd + 1) / 8;
    bstride = gridDim.x >> 3;
  }
  // Dynamic scheduling (ga.sched): workgroup w starts with row block w and claims every further one from a counter, one block
  // ahead (the claim issued at the top of block i is published through LDS in the middle of it and names the block after next), so
  // that workgroups that start late -- or share the chip with another stream's launch -- take fewer blocks instead of finishing last.
  int* const sched = ga.sched;
  const bool dyn = sched != nullptr && !LIN;
  // (the claim slots: 16 bytes of the dynamic LDS behind the biases -- a static array would not fit beside 160 KiB of dynamic LDS)
  // (accessed with workgroup-scope atomics only -- rs_claim_store / rs_claim_load: thread 0's stores and every lane's loads stay where
  // they are written relative to the barriers of the products; the s_barrier builtin carries no memory semantics the compiler would
  // have to respect for plain LDS accesses)
  int* const s_claim = reinterpret_cast<int*>(s_ring + 2 * rs_stage_bytes(rs_stage_tiles(MODE)) + WAVES * 3 * 2048 + (2 + chain) * 32 * NT * 4);
  if (dyn) { bfirst = blockIdx.x; bstride = 0; bend = nblocks; }
  const int64_t mine = dyn ? (bfirst < bend ? 1 : 0) : (bend > bfirst ? (bend - bfirst + bstride - 1) / bstride : 0);
  if (dyn && threadIdx.x == 0) rs_claim_store(&s_claim[0], (int)gridDim.x + atomicAdd(&sched[0], 1));

  constexpr int STG = rs_stage_tiles(MODE), RING_BYTES = 2 * rs_stage_bytes(STG);
  RsStream<NT, T0, WAVES, PC, LIN, STG> ws;
  // RS_LINEAR: `chain` is the number of column groups this workgroup evaluates (blockIdx.y selects which: grids with fewer row
  // blocks than CUs are spread over the groups as well)
  const int64_t seg_bytes = (int64_t)ks1 * T0 * rs_tile_bytes(PC);
  if (LIN) {
    wq += (int64_t)blockIdx.y * chain * seg_bytes;
    b1 = b1 ? b1 + (int64_t)blockIdx.y * chain * 32 * NT : b1;
    out += (int64_t)blockIdx.y * chain * 32 * NT;
  }
  ws.base = wq;
  ws.ring = s_ring;
  ws.ks1 = ks1;
  ws.seg = 0;
  ws.pos = 0;
  ws.off = 0;
  ws.shape(chain);
  ws.left = mine * ws.chunks;
  ws.stage = 0;
  ws.wave = wave;
  ws.lane = lane;
  ws.prime();
  // the biases live in LDS: as kernel-lifetime registers (where the compiler would hoist them) they cost 16 per tile
  float* s_b1 = reinterpret_cast<float*>(s_ring + RING_BYTES + WAVES * 3 * 2048);
  float* s_b0 = s_b1 + 32 * NT;                 // directly behind b1: the first product initialises both in one sweep
  if (LIN) {
    for (int i = threadIdx.x; i < 32 * NT * chain; i += THREADS) s_b1[i] = b1 ? b1[i] : 0.0f;      // one bias block per group
  } else {
    for (int i = threadIdx.x; i < 32 * NT; i += THREADS) {
      s_b1[i] = b1[i];
      if (PROJ) s_b0[i] = b0[i];
    }
    for (int i = threadIdx.x; i < 32 * NT * chain; i += THREADS) s_b1[2 * 32 * NT + i] = b1[32 * NT + i];   // chained layers
  }
  __syncthreads();

  RsX<GATHER> xs;
  xs.slots = s_ring + RING_BYTES + wave * (3 * 2048);
  xs.k_in = k_in;
  xs.g = g;
  xs.lane = lane;
  xs.table = ga.table;
  xs.head = ga.head;
  xs.fshift = ga.fshift;
  auto row_of = [&](int64_t blk) {                // this lane's row of a row block (+ its tuple's point indices)
    const int64_t row = blk * BLOCK_ROWS + wave * 32 + r;
    const int64_t rc = row < rows ? row : rows - 1;
    RsRow rw;
    rw.xrow = LOCAL_IDX ? ga.table : x + rc * ldx;      // (no head array in the encoding modes: any valid address for masked loads)
    rw.g0 = rw.g1 = rw.g2 = rw.g3 = rw.g4 = rw.g5 = rw.g6 = rw.g7 = 0;
    if (GATHER || SUMG) {
      const int32_t* p = ga.gidx + rc * ga.slots;
      int p0 = 0;
      if (LOCAL_IDX) {                      // scene-local indices: + the point offset of the row's scene
        int lo = 0, hi = ga.B;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if ((int64_t)ga.tup_off[mid] <= rc) lo = mid; else hi = mid;
        }
        p0 = ga.pt_off[lo];
      }
      rw.g0 = p[0];
      rw.g1 = ga.slots > 1 ? p[1] : 0; rw.g2 = ga.slots > 2 ? p[2] : 0; rw.g3 = ga.slots > 3 ? p[3] : 0;
      rw.g4 = ga.slots > 4 ? p[4] : 0; rw.g5 = ga.slots > 5 ? p[5] : 0; rw.g6 = ga.slots > 6 ? p[6] : 0;
      rw.g7 = ga.slots > 7 ? p[7] : 0;
      if (LOCAL_IDX) {
        rw.g0 += p0; rw.g1 += ga.slots > 1 ? p0 : 0; rw.g2 += ga.slots > 2 ? p0 : 0; rw.g3 += ga.slots > 3 ? p0 : 0;
        rw.g4 += ga.slots > 4 ? p0 : 0; rw.g5 += ga.slots > 5 ? p0 : 0; rw.g6 += ga.slots > 6 ? p0 : 0; rw.g7 += ga.slots > 7 ? p0 : 0;
      }
    }
    return rw;
  };
  int64_t next_blk = dyn ? (int64_t)rs_claim_load(&s_claim[0]) : bfirst + bstride;      // (after the barrier above)
  int par = 1;
  RsRow cur = row_of(bfirst < bend ? bfirst : 0);
  auto first_tiles = [&](const RsRow rw) {        // x tiles of K steps 0, 1, 2 of a row block into slots 0, 1, 2
    if constexpr (MODE == RS_ENCODE) {
      rs_encode_tiles(xs.slots, lane, g, ga.pts, ga.nrm, ga.table, ga.fshift, rw.g0, rw.g1, rw.g2, rw.g3, rw.g4);
    } else if constexpr (MODE == RS_SUMENCODE) {
      rs_encode_coord_tiles(xs.slots, lane, g, ga.pts, rw.g0, rw.g1, rw.g2, rw.g3, rw.g4);
    } else {
      xs.issue(0, 0, rw);
      xs.issue(1, 1, rw);
      xs.issue(2, 2, rw);
    }
  };
  if (bfirst < bend) first_tiles(cur);
  if constexpr (LIN) {
    // ---- plain Linear: per row block, one pass over x per column group; the accumulators go straight to memory -----------
    for (int64_t blk = bfirst; blk < bend; blk += bstride) {
      const int64_t row = blk * BLOCK_ROWS + wave * 32 + r;
      const bool in = row < rows;
      const bool more = blk + bstride < bend;
      const RsRow nxt = row_of(more ? blk + bstride : blk);
#pragma unroll 1
      for (int grp = 0; grp < chain; ++grp) {
        f32x16 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) rs_load_tile(acc[u], s_b1 + 32 * (grp * NT + u), g);
        rs_product_x<NT, false, PC>(acc, xs, cur, ks1, ws);
        const bool last = grp + 1 == chain;
        if (!last || more) {                      // the x tiles of the next pass: the same rows again, or the next row block's
          const RsRow nx = last ? nxt : cur;
          xs.issue(0, 0, nx);
          xs.issue(1, 1, nx);
          xs.issue(2, 2, nx);
        }
        if (in) {
          float* orow = out + row * ldo + (int64_t)grp * 32 * NT + 4 * g;
#pragma unroll
          for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v;
              v.x = acc[u][4 * q + 0]; v.y = acc[u][4 * q + 1]; v.z = acc[u][4 * q + 2]; v.w = acc[u][4 * q + 3];
              if (PC == 2) v *= bs;
              *reinterpret_cast<f32x4*>(orow + 32 * u + 8 * q) = v;
            }
          }
        }
      }
      cur = nxt;
    }
    RS_WAIT(0, 15);
    return;
  }
  for (int64_t blk = bfirst; blk < bend;) {
    const int64_t row = blk * BLOCK_ROWS + wave * 32 + r;
    const bool in = row < rows;
    const float* xrow = cur.xrow;
    const int64_t nb = dyn ? next_blk : blk + bstride;
    const bool more = nb < bend;
    if (dyn && more) ws.left += ws.chunks;                          // the stream runs on into the next block's weights
    const RsRow nxt = row_of(more ? nb : blk);                    // (GATHER: its index loads are used after the first product)
    int claim = 0;
    if (dyn && threadIdx.x == 0) claim = atomicAdd(&sched[0], 1);   // the block after next; published after the first product
    // ---- h^T = relu(W1 x^T + b1)  [and skip^T = W0 x^T + b0 of a projection layer] ----------------------------
    f32x16 acc[2 * NT];                         // h tiles, then the output tiles
    f32x16 (&h)[NT] = *reinterpret_cast<f32x16 (*)[NT]>(&acc[0]);
    f32x16 (&o)[NT] = *reinterpret_cast<f32x16 (*)[NT]>(&acc[NT]);
#pragma unroll
    for (int u = 0; u < T0; ++u) rs_load_tile(acc[u], s_b1 + 32 * u, g);
    if constexpr (SUMG) {
      // + sum_i table[gidx[row, i]][i][:]: the per-point parts of x W1^T (tiles 0 .. NT-1) and x W0^T (tiles NT .. 2 NT-1) of this
      // lane's row, in slot order (a fixed order: the sums do not depend on the launch geometry).  Lane (r, g) owns features
      // 32 u + 8 q + 4 g + (0..3) of tile u: one 16-byte load each, 16 of them (4 tiles) requested before the first is added.
#pragma unroll 1
      for (int i = 0; i < ((RS_DBG & 64) ? 0 : ga.slots); ++i) {
        int p = cur.g0;
        p = i == 1 ? cur.g1 : p; p = i == 2 ? cur.g2 : p; p = i == 3 ? cur.g3 : p; p = i == 4 ? cur.g4 : p;
        p = i == 5 ? cur.g5 : p; p = i == 6 ? cur.g6 : p; p = i == 7 ? cur.g7 : p;
        if (RS_DBG & 128) p &= 255;
        const float* trow = ga.table + ((int64_t)p * ga.tld + (int64_t)i * (32 * T0)) + 4 * g;
#pragma unroll
        for (int u0 = 0; u0 < T0; u0 += 4) {
          f32x4 t[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) t[j] = *reinterpret_cast<const f32x4*>(trow + 32 * (u0 + (j >> 2)) + 8 * (j & 3));
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            f32x16& a = acc[u0 + (j >> 2)];
            const int q = j & 3;
            if (PC == 2) {                           // (f16x2: the accumulators carry the weight scale)
              a[4 * q + 0] += t[j].x * wscale; a[4 * q + 1] += t[j].y * wscale; a[4 * q + 2] += t[j].z * wscale; a[4 * q + 3] += t[j].w * wscale;
            } else {
              a[4 * q + 0] += t[j].x; a[4 * q + 1] += t[j].y; a[4 * q + 2] += t[j].z; a[4 * q + 3] += t[j].w;
            }
          }
        }
      }
    }
    if (!PROJ && (WAVES == 8 || RS_EARLY_RESIDUAL)) {   // residual of an identity layer: requested before the first product (while
                                                        // the x tiles of the same rows are passing through L2)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        if (PC == 2) rs_load_tile_scaled(o[u], ((RS_DBG & 32) ? (const float*)s_b1 : xrow) + 32 * u, g, wscale);
        else rs_load_tile(o[u], ((RS_DBG & 32) ? (const float*)s_b1 : xrow) + 32 * u, g);
      }
    }
    {
      f32x16 (&first)[T0] = *reinterpret_cast<f32x16 (*)[T0]>(&acc[0]);
      rs_product_x<T0, PF, PC>(first, xs, cur, ks1, ws);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
#pragma unroll
      for (int e = 0; e < 16; ++e) h[u][e] = (h[u][e] < 0.0f) ? 0.0f : h[u][e];        // NaN stays NaN like torch.relu
    }
    if (!PROJ && WAVES == 4 && !RS_EARLY_RESIDUAL) {     // (measured alternative: after it; re-reads 1.2 GB per launch from HBM)
      if (PC == 2) {
        // scaled on the way in, two tiles at a time: the products are VALU results, which reach the accumulator half of the
        // register file through v_accvgpr_write -- all eight tiles at once would need 128 transient registers (spills)
#pragma unroll
        for (int u = 0; u < NT; u += 2) {
          rs_load_tile_scaled(o[u], ((RS_DBG & 32) ? (const float*)s_b1 : xrow) + 32 * u, g, wscale);
          if (u + 1 < NT) rs_load_tile_scaled(o[u + 1], ((RS_DBG & 32) ? (const float*)s_b1 : xrow) + 32 * (u + 1), g, wscale);
#pragma unroll
          for (int e = 0; e < 16; ++e) {         // parked in the accumulator half of the register file right away
            asm volatile("" : "+a"(o[u][e]));
            if (u + 1 < NT) asm volatile("" : "+a"(o[u + 1][e]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u) rs_load_tile(o[u], ((RS_DBG & 32) ? (const float*)s_b1 : xrow) + 32 * u, g);
      }
    }
    if (dyn && threadIdx.x == 0) rs_claim_store(&s_claim[par], (int)gridDim.x + claim);   // (read at the loop's end: barriers of the products between)
    // the next row block's first x tiles travel during the remaining products (their slots are free now)
    if (more) first_tiles(nxt);
    cur = nxt;
    // ---- y^T = skip^T + W2 h^T --------------------------------------------------------------------------
    rs_product_h<NT, NT, PF, PC>(o, h, ws, bs);
    if (tap.out && in) {                        // the first layer's output is needed in memory as well (the tuple features)
      float* trow = tap.out + row * tap.ld + 4 * g;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v.x = o[u][4 * q + 0]; v.y = o[u][4 * q + 1]; v.z = o[u][4 * q + 2]; v.w = o[u][4 * q + 3];
          if (PC == 2) v *= bs;
          *reinterpret_cast<f32x4*>(trow + 32 * u + 8 * q) = v;
        }
      }
    }
    // ---- the identity layers chained behind (same width): y <- y + relu(y W1^T + b1) W2^T with y = the output tiles,
    // never leaving the registers; their W1 is packed in accumulator feature order like every W2
#pragma unroll 1
    for (int l = 0; l < chain; ++l) {
      const float* bl = s_b1 + 32 * NT * (2 + l);
#pragma unroll
      for (int u = 0; u < NT; ++u) rs_load_tile(h[u], bl + 32 * u, g);
      rs_product_h<NT, NT, PF, PC>(h, o, ws, bs);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int e = 0; e < 16; ++e) h[u][e] = (h[u][e] < 0.0f) ? 0.0f : h[u][e];
      }
      rs_product_h<NT, NT, PF, PC>(o, h, ws, bs);
    }
    if (DECODE) {
      // (the draw as the row block's own epilogue.  Cut into pieces and hidden under the NEXT block's first product it was
      // slower -- round 4, 2.18 against 2.06 ms -- and left the source in round 6: docs/experiments/r6_removed_mlp_variants.patch)
      if constexpr (NT == 6) rs_decode_epilogue<NT, PC>(o, dc, in ? row : rows - 1, in, g, bs);
    } else if (in && (!(RS_DBG & 32) || o[0][0] == 1.2345e30f)) {
      float* orow = out + row * ldo + 4 * g;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v.x = o[u][4 * q + 0]; v.y = o[u][4 * q + 1]; v.z = o[u][4 * q + 2]; v.w = o[u][4 * q + 3];
          if (PC == 2) v *= bs;                  // back to the true scale (f16x2: the accumulators carry the weight scale)
          *reinterpret_cast<f32x4*>(orow + 32 * u + 8 * q) = v;
        }
      }
    }
    blk = nb;
    if (dyn) {
      next_blk = rs_claim_load(&s_claim[par]);
      par ^= 1;
    }
  }
  // Nothing may be in flight when the wavefront ends: the pieces issued during the last chunk (plan() past the end: a re-read of
  // the stream's first KiB, kept so that every chunk issues the same number of operations) are LDS-DMA loads, and an LDS-DMA
  // load that lands after s_endpgm writes into LDS that may already belong to ANOTHER workgroup (observed: a rotation-vote
  // workgroup starting on the CU while the last wavefronts of a 512-register MLP workgroup were leaving had the first KiB of
  // its LDS -- sphere bins 0..62 -- overwritten by weight bytes, scratch/rot_race_probe3.py).
  RS_WAIT(0, 15);
  if (dyn && threadIdx.x == 0) {                  // the last workgroup out leaves the counters zeroed for the next launch
    if (atomicAdd(&sched[1], 1) == (int)gridDim.x - 1) {
      __hip_atomic_store(&sched[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sched[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int64_t rs_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj, int32_t chain, int pc) {
  if (k_in <= 0 || !(n_out == 64 || n_out == 128 || n_out == 192 || n_out == 256) || chain < 0 || chain > 15) return -1;
  const int64_t nt = n_out / 32, ks1 = (k_in + 15) / 16;
  return (ks1 * (proj ? 2 : 1) + (1 + 2 * (int64_t)chain) * 2 * nt) * nt * rs_tile_bytes(pc);
}

extern "C" int64_t cppf_reslayer_split_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj, int32_t chain) {
  return rs_stream_bytes(k_in, n_out, proj, chain, 3);
}

// test hook (tests/test_mlp_split.py race screen): > 0 forces the number of persistent workgroups of every launch
static std::atomic<int> g_rs_debug_cus{0};
extern "C" int cppf_reslayer_split_debug_grid(int32_t workgroups) {
  g_rs_debug_cus.store(workgroups > 0 ? workgroups : 0);
  return CPPF_OK;
}

// Batch-mode knob (include/cppf_hip.h): CUs every persistent launch leaves to the kernels of other streams
static std::atomic<int> g_rs_reserved_cus{0};
// returns the previous reservation (>= 0); a negative argument only queries
extern "C" int cppf_mlp_reserve_cus(int32_t cus) {
  CPPF_CHECK_ARG(cus <= 1024);
  return cus < 0 ? g_rs_reserved_cus.load() : g_rs_reserved_cus.exchange(cus);
}

template <int NT, bool PROJ, bool GATHER = false, bool DECODE = false, int PC = 3, int MODE = RS_RESLAYER>
static int rs_launch(const float* x, int64_t ldx, int k_in, float* out, int64_t ldo, int64_t rows, const char* wq,
                     const float* b1, const float* b0, int chain, int cus, hipStream_t stream, RsGather ga = RsGather(),
                     RsDecode dc = RsDecode(), RsTap tap = RsTap(), float wscale = 1.0f) {
  constexpr int WAVES = rs_waves_mode(NT, MODE), THREADS = 64 * WAVES, BLOCK_ROWS = 32 * WAVES;
  const int64_t nblocks = (rows + BLOCK_ROWS - 1) / BLOCK_ROWS;
  const int forced = g_rs_debug_cus.load();
  if (forced > 0) cus = forced;
  else cus = cus - g_rs_reserved_cus.load() > cus / 2 ? cus - g_rs_reserved_cus.load() : (cus + 1) / 2;   // never below half the chip
  cus *= rs_wgs_per_cu(MODE);
  const unsigned grid = (unsigned)(nblocks < cus ? nblocks : cus);
  // RS_LINEAR: `chain` = column groups of 32 NT outputs; with fewer row blocks than CUs the groups are spread over blockIdx.y
  unsigned gy = 1;
  if (MODE == RS_LINEAR && (int64_t)grid * 2 <= cus) {
    for (int d = chain; d >= 1; --d)
      if (chain % d == 0 && (int64_t)grid * d <= cus) { gy = (unsigned)d; break; }
    chain /= (int)gy;
  }
  const int lds_bytes = 2 * rs_stage_bytes(rs_stage_tiles(MODE)) + WAVES * 3 * 2048 + (MODE == RS_LINEAR ? chain : 2 + chain) * 32 * NT * 4 + 16;
  if (lds_bytes > 160 * 1024 / rs_wgs_per_cu(MODE)) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "reslayer_split: %d bytes of LDS for the biases of this launch (too many column groups)", lds_bytes);
    return CPPF_EINVAL;
  }
  {
    // more than 64 KiB of dynamic LDS: declared once per kernel and device
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    CPPF_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev & 63]) {
      CPPF_HIP(hipFuncSetAttribute((const void*)reslayer_split_kernel<NT, PROJ, GATHER, DECODE, PC, MODE>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done[dev & 63] = true;
    }
  }
  hipLaunchKernelGGL((reslayer_split_kernel<NT, PROJ, GATHER, DECODE, PC, MODE>), dim3(grid, gy), dim3(THREADS), lds_bytes, stream, x,
                     ldx, k_in, out, ldo, rows, wq, b1, b0, chain, ga, dc, tap, wscale);
  CPPF_LAUNCH_CHECK();
  return CPPF_OK;
}

static int rs_cus() {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
  return n_cu > 0 ? n_cu : 256;
}

// out[rows, n_out] = L_chain(...L_1(L_0(x))): L_0(x) = skip(x) + relu(x[:, :k_in] W1^T + b1) W2^T with skip = x (b0 == NULL;
// then k_in == n_out and out may be x itself) or x W0^T + b0, followed by `chain` identity-skip layers of the same width
// evaluated on the output tiles in registers.  x float32 [rows, >= k_in] with row stride ldx, out float32
// with row stride ldo (device; 16-byte aligned rows: ldx, ldo multiples of 4); k_in a multiple of 8; n_out in {64, 128,
// 192, 256}.  wq = the layers' weights as the packed split stream (cppf_reslayer_split_stream_bytes bytes;
// cppf2_amd.models.pack_split documents the order); b1 = float32[(1 + chain) * n_out], the first-layer biases of L_0,
// L_1, ...  The second-layer biases are the caller's (carried as a pending offset by cppf2_amd.models.fused_stack).
static int rs_dispatch(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                       const void* wq, int64_t wq_bytes, const float* b1, const float* b0, int32_t chain, RsTap tap,
                       int32_t* sched, void* stream, const char* fn) {
  if (!(x && out && wq && b1 && rows >= 0) || !(k_in > 0 && (k_in & 7) == 0 && ldx >= k_in && (ldx & 3) == 0 && (ldo & 3) == 0 && ldo >= n_out) ||
      !(n_out == 64 || n_out == 128 || n_out == 192 || n_out == 256) || !(b0 != nullptr || k_in == n_out) ||
      !(chain >= 0 && chain <= 15) || (((uintptr_t)x | (uintptr_t)out | (uintptr_t)wq | (uintptr_t)tap.out) & 15) != 0 ||
      (tap.out && ((tap.ld & 3) != 0 || tap.ld < n_out)) ||
      wq_bytes != cppf_reslayer_split_stream_bytes(k_in, n_out, b0 != nullptr, chain)) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "%s: invalid argument (pointers, 16-byte alignment, k_in %% 8, strides %% 4, n_out in "
             "{64,128,192,256}, chain <= 15, stream size)", fn);
    return CPPF_EINVAL;
  }
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
  RsGather ga;
  ga.sched = sched;
  const RsDecode dc = RsDecode();
  switch (n_out / 32) {
    case 2: return proj ? rs_launch<2, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap)
                        : rs_launch<2, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap);
    case 4: return proj ? rs_launch<4, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap)
                        : rs_launch<4, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap);
    case 6: return proj ? rs_launch<6, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap)
                        : rs_launch<6, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap);
    default: return proj ? rs_launch<8, true>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap)
                         : rs_launch<8, false>(x, ldx, k_in, out, ldo, rows, w, b1, b0, chain, n_cu, st, ga, dc, tap);
  }
}

extern "C" int cppf_reslayer_split(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out,
                                   int64_t rows, const void* wq, int64_t wq_bytes, const float* b1, const float* b0,
                                   int32_t chain, int32_t* sched, void* stream) {
  return rs_dispatch(x, ldx, k_in, out, ldo, n_out, rows, wq, wq_bytes, b1, b0, chain, RsTap(), sched, stream, __func__);
}

// cppf_reslayer_split with a second output: first_out [rows, >= n_out] (row stride ld_first) receives the activation after the
// FIRST layer, out the one after the whole chain -- the first layer's result is needed in memory (the tuple features the scale
// head reads, train_shot.py:112-114) while the identity layers behind it (the logit head's, train_shot.py:62-64) go on in
// registers: the chain's input is never re-read.  first_out may not alias x or out.
extern "C" int cppf_reslayer_split_tap(const float* x, int64_t ldx, int32_t k_in, float* first_out, int64_t ld_first, float* out,
                                       int64_t ldo, int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes,
                                       const float* b1, const float* b0, int32_t chain, int32_t* sched, void* stream) {
  if (!first_out || first_out == out || first_out == x) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "%s: first_out must be a buffer of its own", __func__);
    return CPPF_EINVAL;
  }
  RsTap tap;
  tap.out = first_out;
  tap.ld = ld_first;
  return rs_dispatch(x, ldx, k_in, out, ldo, n_out, rows, wq, wq_bytes, b1, b0, chain, tap, sched, stream, __func__);
}

// The first ResLayer of the SHOT model's tuple encoder fed by the tuple encode itself (train_shot.py:75-83 never materialised):
// row t of the layer's input is [heads[t, 0:head_cols] | table[gidx[t, 0]] | ... | table[gidx[t, slots-1]]], i.e. k_in =
// head_cols + slots * fdim columns, gathered by the kernel's x-tile fetches.  heads float32 [rows, >= head_cols] with row
// stride ld_heads (cppf_encode_tuples_shot_heads writes it, with gidx int32 [rows, slots] = scene base + tuple index);
// table float32 [points, fdim], fdim a power of two >= 8, head_cols % 8 == 0, slots <= 8.  Everything else as
// cppf_reslayer_split (a projection layer, n_out = 128; `chain` identity layers behind it); the results are bit-identical
// to cppf_reslayer_split on the materialised rows.
extern "C" int cppf_reslayer_split_gather(const float* heads, int64_t ld_heads, int32_t head_cols, const int32_t* gidx,
                                          int32_t slots, const float* table, int32_t fdim, float* out, int64_t ldo,
                                          int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1,
                                          const float* b0, int32_t chain, int32_t* sched, void* stream) {
  CPPF_CHECK_ARG(heads && gidx && table && out && wq && b1 && b0 && rows >= 0);
  CPPF_CHECK_ARG(head_cols >= 0 && (head_cols & 7) == 0 && ld_heads >= head_cols && (ld_heads & 3) == 0);
  CPPF_CHECK_ARG(slots >= 1 && slots <= 8 && fdim >= 8 && (fdim & (fdim - 1)) == 0);
  CPPF_CHECK_ARG((ldo & 3) == 0 && ldo >= n_out && chain >= 0 && chain <= 15);
  CPPF_CHECK_ARG((((uintptr_t)heads | (uintptr_t)table | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  const int k_in = head_cols + slots * fdim;
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(k_in, n_out, 1, chain));
  if (n_out != 128) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_reslayer_split_gather: n_out = %d (only the 128-wide projection layer of the "
             "tuple encoder has a gathering kernel)", n_out);
    return CPPF_EUNSUPPORTED;
  }
  if (rows == 0) return CPPF_OK;
  int dev = 0, n_cu = 0;
  CPPF_HIP(hipGetDevice(&dev));
  CPPF_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  RsGather ga;
  ga.gidx = gidx;
  ga.table = table;
  ga.slots = slots;
  ga.head = head_cols;
  ga.fshift = __builtin_ctz((unsigned)fdim);
  ga.sched = sched;
  return rs_launch<4, true, true>(heads, ld_heads, k_in, out, ldo, rows, static_cast<const char*>(wq), b1, b0, chain,
                                  n_cu > 0 ? n_cu : 256, (hipStream_t)stream, ga);
}

// cppf_reslayer_split_gather with the pair features built inside the kernel (RS_ENCODE): prepare_tuple_inputs (train_shot.py:75-83)
// + the tuple encoder's first launch without any per-tuple array in between.  pts / normals float32 [points, 3], idx int32
// [rows, 5] scene-local tuple indices (the sampler's), pt_off / tup_off int32 [B + 1]; table / fdim / out / wq / b1 / b0 / chain as
// cppf_reslayer_split_gather with head_cols = 40 (wq = cppf_reslayer_split_stream_bytes(40 + 5 fdim, 128, 1, chain) bytes).
extern "C" int cppf_reslayer_split_encode(int B, const float* pts, const float* normals, const int32_t* idx, int32_t k,
                                          const int32_t* pt_off, const int32_t* tup_off, const float* table, int32_t fdim,
                                          float* out, int64_t ldo, int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes,
                                          const float* b1, const float* b0, int32_t chain, int32_t* sched, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && normals && idx && pt_off && tup_off && table && out && wq && b1 && b0 && rows >= 0);
  CPPF_CHECK_ARG(fdim >= 8 && (fdim & (fdim - 1)) == 0 && (ldo & 3) == 0 && ldo >= n_out && chain >= 0 && chain <= 15);
  CPPF_CHECK_ARG((((uintptr_t)table | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  if (k != 5 || n_out != 128) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_reslayer_split_encode: k = %d, n_out = %d (the kernel builds the 40 pair features of "
             "5-point tuples for the 128-wide projection layer of the tuple encoder)", k, n_out);
    return CPPF_EUNSUPPORTED;
  }
  const int k_in = 40 + k * fdim;
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(k_in, n_out, 1, chain));
  if (rows == 0) return CPPF_OK;
  RsGather ga;
  ga.gidx = idx;
  ga.table = table;
  ga.slots = k;
  ga.head = 40;
  ga.fshift = __builtin_ctz((unsigned)fdim);
  ga.pts = pts;
  ga.nrm = normals;
  ga.pt_off = pt_off;
  ga.tup_off = tup_off;
  ga.B = B;
  ga.sched = sched;
  return rs_launch<4, true, true, false, 3, RS_ENCODE>(nullptr, 0, k_in, out, ldo, rows, static_cast<const char*>(wq), b1, b0, chain,
                                                       rs_cus(), (hipStream_t)stream, ga);
}

// The output layer of the logit head (train_shot.py:62-66: a 192-wide projection ResLayer = 6 coordinates x 32 bins) with the
// bin draw of eval.py:225-229 as its epilogue: bins[t, c] = inverse-CDF draw of softmax(logits[t, c, :] (+ prior[t, c, :])) at
// uniforms[t, c] -- the same arithmetic as cppf_decode_bins, bit for bit -- so the logits are never written; follow with
// cppf_decode_from_bins for the vote parameters.  x / wq / b1 / b0 as cppf_reslayer_split with n_out = 192, chain = 0;
// logit_prior float32 [rows, 192] or NULL, uniforms float32 [rows, 6], bins int32 [rows, 6].
extern "C" int cppf_reslayer_split_decode_prior(const float* x, int64_t ldx, int32_t k_in, int64_t rows, const void* wq, int64_t wq_bytes,
                                                const float* b1, const float* b0, const float* logit_prior, const float* prior_pos,
                                                float prior_inv_sigma, const float* uniforms, int32_t* bins, int32_t* sched,
                                                void* stream) {
  CPPF_CHECK_ARG(x && wq && b1 && b0 && uniforms && bins && rows >= 0);
  CPPF_CHECK_ARG(!(logit_prior && prior_pos) && (!prior_pos || (prior_inv_sigma > 0.0f && prior_inv_sigma < INFINITY)));
  CPPF_CHECK_ARG(k_in > 0 && (k_in & 7) == 0 && ldx >= k_in && (ldx & 3) == 0);
  CPPF_CHECK_ARG((((uintptr_t)x | (uintptr_t)wq | (uintptr_t)logit_prior) & 15) == 0);
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(k_in, 192, 1, 0));
  if (rows == 0) return CPPF_OK;
  int dev = 0, n_cu = 0;
  CPPF_HIP(hipGetDevice(&dev));
  CPPF_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  RsDecode dc;
  dc.prior = logit_prior;
  dc.prior_pos = prior_pos;
  dc.prior_inv_sigma = prior_pos ? prior_inv_sigma : 0.0f;
  dc.uniforms = uniforms;
  dc.bins = bins;
  RsGather ga;
  ga.sched = sched;
  return rs_launch<6, true, false, true>(x, ldx, k_in, nullptr, 192, rows, static_cast<const char*>(wq), b1, b0, 0,
                                         n_cu > 0 ? n_cu : 256, (hipStream_t)stream, ga, dc);
}


// the stable form: the reference draws from the network's own logits (eval.py:225-229)
extern "C" int cppf_reslayer_split_decode(const float* x, int64_t ldx, int32_t k_in, int64_t rows, const void* wq, int64_t wq_bytes,
                                          const float* b1, const float* b0, const float* uniforms, int32_t* bins, int32_t* sched,
                                          void* stream) {
  return cppf_reslayer_split_decode_prior(x, ldx, k_in, rows, wq, wq_bytes, b1, b0, nullptr, nullptr, 0.0f, uniforms, bins, sched, stream);
}


// A plain nn.Linear on the matrix cores in the same split arithmetic: out[rows, n_out] = x[rows, :k_in] W^T + bias (bias may be
// NULL).  The DINO model's per-point transforms (train_dino.py:86-87: desc_transform over [N, 1024]; desc_pair_transform's
// slices, folded with the first ResLayer's weights into per-point slot tables, see cppf_reslayer_split_sumgather) run through
// it, so that no layer of either model is left to a BLAS library.  n_out a multiple of 256 (column groups of 256: one pass over
// x each), k_in a multiple of 8; x / out as cppf_reslayer_split.  wq = cppf_linear_split_stream_bytes(k_in, n_out) bytes: per
// column group the first-product segment of pack_split for that group's 256 rows of W (cppf2_amd.models.pack_linear).
extern "C" int64_t cppf_linear_split_stream_bytes(int32_t k_in, int32_t n_out) {
  if (k_in <= 0 || n_out <= 0 || (n_out & 255) != 0) return -1;
  return (int64_t)((k_in + 15) / 16) * (n_out / 32) * rs_tile_bytes(3);
}

extern "C" int cppf_linear_split(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                                 const void* wq, int64_t wq_bytes, const float* bias, void* stream) {
  CPPF_CHECK_ARG(x && out && wq && rows >= 0);
  CPPF_CHECK_ARG(k_in > 0 && (k_in & 7) == 0 && ldx >= k_in && (ldx & 3) == 0 && n_out > 0 && (n_out & 255) == 0 && ldo >= n_out && (ldo & 3) == 0);
  CPPF_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  CPPF_CHECK_ARG(wq_bytes == cppf_linear_split_stream_bytes(k_in, n_out));
  if (rows == 0) return CPPF_OK;
  return rs_launch<8, false, false, false, 3, RS_LINEAR>(x, ldx, k_in, out, ldo, rows, static_cast<const char*>(wq), bias, nullptr,
                                                         n_out / 256, rs_cus(), (hipStream_t)stream);
}

// The first ResLayer of a tuple encoder (a 128-wide projection layer, + `chain` identity layers) whose input row is
//   [heads[t, 0:head_cols] | s(t)],   s(t) = a sum / concatenation over the tuple's points of per-point vectors
// (DINO model, train_dino.py:91-97: s = desc_pair_transform(cat_i desc_transform(desc[idx_i])); SHOT model, train_shot.py:75-83:
// s = cat_i feat[idx_i]) WITHOUT forming the row: x W1^T and x W0^T are linear in s, so their per-point parts are evaluated once
// per point and slot -- tables[n][i] = [W1_i p_n | W0_i p_n], 128 + 128 floats, written by cppf_linear_split from host-folded
// weights (cppf2_amd.models) -- and the kernel starts its accumulators from bias + sum_i tables[gidx[t, i]][i] (slot order);
// only the head columns (pair coordinates: 30 -> 32 columns) go through the matrix cores in the first product.  The tuple rows
// are never written, and the first product shrinks from 18 (DINO) / 23 (SHOT) K steps to 2 / 3.
// heads / gidx / out / b1 / b0 / chain as cppf_reslayer_split_gather; tables float32 [points, slots, 256] with point pitch
// ld_tables floats (>= slots * 256, % 4); wq = cppf_reslayer_split_stream_bytes(head_cols, 128, 1, chain) bytes (the head
// columns' weights only); b1 / b0 must include the bias terms of the folded transforms.
extern "C" int cppf_reslayer_split_sumgather(const float* heads, int64_t ld_heads, int32_t head_cols, const int32_t* gidx,
                                             int32_t slots, const float* tables, int64_t ld_tables, float* out, int64_t ldo,
                                             int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1,
                                             const float* b0, int32_t chain, int32_t* sched, void* stream) {
  CPPF_CHECK_ARG(heads && gidx && tables && out && wq && b1 && b0 && rows >= 0);
  CPPF_CHECK_ARG(head_cols > 0 && (head_cols & 7) == 0 && ld_heads >= head_cols && (ld_heads & 3) == 0);
  CPPF_CHECK_ARG(slots >= 1 && slots <= 8 && ld_tables >= (int64_t)slots * 256 && (ld_tables & 3) == 0);
  CPPF_CHECK_ARG((ldo & 3) == 0 && ldo >= n_out && chain >= 0 && chain <= 15);
  CPPF_CHECK_ARG((((uintptr_t)heads | (uintptr_t)tables | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(head_cols, n_out, 1, chain));
  if (n_out != 128) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_reslayer_split_sumgather: n_out = %d (only the 128-wide projection layer of the "
             "tuple encoders has this kernel)", n_out);
    return CPPF_EUNSUPPORTED;
  }
  if (rows == 0) return CPPF_OK;
  RsGather ga;
  ga.gidx = gidx;
  ga.table = tables;
  ga.slots = slots;
  ga.tld = ld_tables;
  ga.sched = sched;
  return rs_launch<4, true, false, false, 3, RS_SUMGATHER>(heads, ld_heads, head_cols, out, ldo, rows, static_cast<const char*>(wq), b1, b0,
                                                           chain, rs_cus(), (hipStream_t)stream, ga);
}


// cppf_reslayer_split_sumgather with the coordinate columns built inside the kernel (RS_SUMENCODE): the DINO model's
// prepare_tuple_inputs (train_dino.py:91-97) + its tuple encoder's first launch with no per-tuple array in between.  pts float32
// [points, 3], idx int32 [rows, 5] scene-local (the sampler's), pt_off / tup_off int32 [B + 1]; tables / wq / b1 / b0 / chain as
// cppf_reslayer_split_sumgather with head_cols = 32.
extern "C" int cppf_reslayer_split_sumencode(int B, const float* pts, const int32_t* idx, int32_t k, const int32_t* pt_off,
                                             const int32_t* tup_off, const float* tables, int64_t ld_tables, float* out, int64_t ldo,
                                             int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1,
                                             const float* b0, int32_t chain, int32_t* sched, void* stream) {
  CPPF_CHECK_ARG(B > 0 && pts && idx && pt_off && tup_off && tables && out && wq && b1 && b0 && rows >= 0);
  CPPF_CHECK_ARG(ld_tables >= (int64_t)k * 256 && (ld_tables & 3) == 0 && (ldo & 3) == 0 && ldo >= n_out && chain >= 0 && chain <= 15);
  CPPF_CHECK_ARG((((uintptr_t)tables | (uintptr_t)out | (uintptr_t)wq) & 15) == 0);
  if (k != 5 || n_out != 128) {
    snprintf(g_cppf_err, sizeof(g_cppf_err), "cppf_reslayer_split_sumencode: k = %d, n_out = %d (5-point tuples, the 128-wide projection "
             "layer of the tuple encoder)", k, n_out);
    return CPPF_EUNSUPPORTED;
  }
  CPPF_CHECK_ARG(wq_bytes == cppf_reslayer_split_stream_bytes(32, n_out, 1, chain));
  if (rows == 0) return CPPF_OK;
  RsGather ga;
  ga.gidx = idx;
  ga.table = tables;
  ga.slots = k;
  ga.tld = ld_tables;
  ga.pts = pts;
  ga.pt_off = pt_off;
  ga.tup_off = tup_off;
  ga.B = B;
  ga.sched = sched;
  return rs_launch<4, true, false, false, 3, RS_SUMENCODE>(nullptr, 0, 32, out, ldo, rows, static_cast<const char*>(wq), b1, b0, chain,
                                                           rs_cus(), (hipStream_t)stream, ga);
}

// ---------------------------------------------------------------------------------------------------------------------
// f16x2 arithmetic (CPPF_MLP_ARITH=split16): one entry point for every launch form
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int64_t cppf_reslayer_split16_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj, int32_t chain) {
  return rs_stream_bytes(k_in, n_out, proj, chain, 2);
}

static RsGather rs16_gather(const CppfReslayerSplit16Args& a) {      // no gather: only the scheduling counters
  RsGather ga;
  ga.sched = a.sched;
  return ga;
}

template <int NT, bool PROJ>
static int rs16_plain(const CppfReslayerSplit16Args& a, int n_cu, RsTap tap) {
  return rs_launch<NT, PROJ, false, false, 2>(a.x, a.ldx, a.k_in, a.out, a.ldo, a.rows, static_cast<const char*>(a.wq), a.b1, a.b0,
                                              a.chain, n_cu, (hipStream_t)a.stream, rs16_gather(a), RsDecode(), tap, a.weight_scale);
}

// The ResLayer launch of cppf_reslayer_split / _tap / _gather / _decode in f16x2 arithmetic (see the kernel's comment): wq =
// the fp16 (hi, lo) pairs of weight_scale x the weights in the same fragment order (cppf2_amd.models.pack_split(arith="f16x2");
// cppf_reslayer_split16_stream_bytes bytes), b1 / b0 = weight_scale x the biases, weight_scale a power of two.  Exactly one of
// {plain, gather (gidx != NULL; x = heads, k_in = head columns), decode (uniforms != NULL)} ; first_out (tap) only with plain.
extern "C" int cppf_reslayer_split16(const CppfReslayerSplit16Args* args) {
  CPPF_CHECK_ARG(args != nullptr);
  const CppfReslayerSplit16Args& a = *args;
  CPPF_CHECK_ARG(a.x && a.wq && (a.b1 || a.mode == 1) && a.rows >= 0 && a.chain >= 0 && a.chain <= 15);
  {
    int e = 0;
    CPPF_CHECK_ARG(a.weight_scale > 0.0f && frexpf(a.weight_scale, &e) == 0.5f);        // a power of two
  }
  CPPF_CHECK_ARG((((uintptr_t)a.x | (uintptr_t)a.out | (uintptr_t)a.wq | (uintptr_t)a.first_out | (uintptr_t)a.table |
                   (uintptr_t)a.logit_prior) & 15) == 0);
  CPPF_CHECK_ARG((a.ldx & 3) == 0 && (a.ldo & 3) == 0 && (a.ld_first & 3) == 0);
  const bool gather = a.gidx != nullptr, decode = a.uniforms != nullptr, proj = a.b0 != nullptr;
  CPPF_CHECK_ARG(!(gather && decode) && !((gather || decode) && a.first_out));
  CPPF_CHECK_ARG(a.mode == 0 || (a.mode == 1 && !gather && !decode && !proj && !a.first_out) || (a.mode == 2 && gather));
  if (a.rows == 0) return CPPF_OK;
  int dev = 0, n_cu = 0;
  CPPF_HIP(hipGetDevice(&dev));
  CPPF_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  const char* w = static_cast<const char*>(a.wq);
  hipStream_t st = (hipStream_t)a.stream;
  if (a.mode == 1) {                             // plain Linear (cppf_linear_split): b1 = weight_scale x the bias or NULL
    CPPF_CHECK_ARG(a.out && a.k_in > 0 && (a.k_in & 7) == 0 && a.ldx >= a.k_in && a.n_out > 0 && (a.n_out & 255) == 0 && a.ldo >= a.n_out);
    CPPF_CHECK_ARG(a.wq_bytes == (int64_t)((a.k_in + 15) / 16) * (a.n_out / 32) * rs_tile_bytes(2));
    return rs_launch<8, false, false, false, 2, RS_LINEAR>(a.x, a.ldx, a.k_in, a.out, a.ldo, a.rows, w, a.b1, nullptr, a.n_out / 256, n_cu,
                                                           st, RsGather(), RsDecode(), RsTap(), a.weight_scale);
  }
  if (a.mode == 2) {                             // per-point slot tables summed into the accumulators (cppf_reslayer_split_sumgather)
    CPPF_CHECK_ARG(a.table && proj && a.out && a.n_out == 128 && a.ldo >= 128);
    CPPF_CHECK_ARG(a.k_in > 0 && (a.k_in & 7) == 0 && a.ldx >= a.k_in);
    CPPF_CHECK_ARG(a.slots >= 1 && a.slots <= 8 && a.ld_table >= (int64_t)a.slots * 256 && (a.ld_table & 3) == 0);
    CPPF_CHECK_ARG(a.wq_bytes == rs_stream_bytes(a.k_in, 128, 1, a.chain, 2));
    RsGather ga;
    ga.gidx = a.gidx;
    ga.table = a.table;
    ga.slots = a.slots;
    ga.tld = a.ld_table;
    ga.sched = a.sched;
    return rs_launch<4, true, false, false, 2, RS_SUMGATHER>(a.x, a.ldx, a.k_in, a.out, a.ldo, a.rows, w, a.b1, a.b0, a.chain, n_cu, st, ga,
                                                             RsDecode(), RsTap(), a.weight_scale);
  }
  if (gather) {
    CPPF_CHECK_ARG(a.table && proj && a.out && a.n_out == 128 && a.ldo >= 128);
    CPPF_CHECK_ARG(a.k_in >= 0 && (a.k_in & 7) == 0 && (a.k_in == 0 || a.ldx >= a.k_in));
    CPPF_CHECK_ARG(a.slots >= 1 && a.slots <= 8 && a.fdim >= 8 && (a.fdim & (a.fdim - 1)) == 0);
    const int k_tot = a.k_in + a.slots * a.fdim;
    CPPF_CHECK_ARG(a.wq_bytes == rs_stream_bytes(k_tot, 128, 1, a.chain, 2));
    RsGather ga;
    ga.gidx = a.gidx;
    ga.table = a.table;
    ga.slots = a.slots;
    ga.head = a.k_in;
    ga.fshift = __builtin_ctz((unsigned)a.fdim);
    ga.sched = a.sched;
    return rs_launch<4, true, true, false, 2>(a.x, a.ldx, k_tot, a.out, a.ldo, a.rows, w, a.b1, a.b0, a.chain, n_cu, st, ga,
                                              RsDecode(), RsTap(), a.weight_scale);
  }
  CPPF_CHECK_ARG(a.k_in > 0 && (a.k_in & 7) == 0 && a.ldx >= a.k_in);
  if (decode) {
    CPPF_CHECK_ARG(proj && a.bins && a.chain == 0 && a.wq_bytes == rs_stream_bytes(a.k_in, 192, 1, 0, 2));
    RsDecode dc;
    CPPF_CHECK_ARG(!(a.logit_prior && a.prior_pos) && (!a.prior_pos || (a.prior_inv_sigma > 0.0f && a.prior_inv_sigma < INFINITY)));
    dc.prior = a.logit_prior;
    dc.prior_pos = a.prior_pos;
    dc.prior_inv_sigma = a.prior_pos ? a.prior_inv_sigma : 0.0f;
    dc.uniforms = a.uniforms;
    dc.bins = a.bins;
    return rs_launch<6, true, false, true, 2>(a.x, a.ldx, a.k_in, nullptr, 192, a.rows, w, a.b1, a.b0, 0, n_cu, st, rs16_gather(a), dc,
                                              RsTap(), a.weight_scale);
  }
  CPPF_CHECK_ARG(a.out && (a.n_out == 64 || a.n_out == 128 || a.n_out == 192 || a.n_out == 256) && a.ldo >= a.n_out);
  CPPF_CHECK_ARG(proj || a.k_in == a.n_out);
  CPPF_CHECK_ARG(a.wq_bytes == rs_stream_bytes(a.k_in, a.n_out, proj, a.chain, 2));
  RsTap tap;
  if (a.first_out) {
    CPPF_CHECK_ARG(a.first_out != a.out && a.first_out != a.x && a.ld_first >= a.n_out);
    tap.out = a.first_out;
    tap.ld = a.ld_first;
  }
  switch (a.n_out / 32) {
    case 2: return proj ? rs16_plain<2, true>(a, n_cu, tap) : rs16_plain<2, false>(a, n_cu, tap);
    case 4: return proj ? rs16_plain<4, true>(a, n_cu, tap) : rs16_plain<4, false>(a, n_cu, tap);
    case 6: return proj ? rs16_plain<6, true>(a, n_cu, tap) : rs16_plain<6, false>(a, n_cu, tap);
    default: return proj ? rs16_plain<8, true>(a, n_cu, tap) : rs16_plain<8, false>(a, n_cu, tap);
  }
}
