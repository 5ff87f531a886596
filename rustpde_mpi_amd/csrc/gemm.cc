// f64 MFMA GEMM of the Poisson eigen-transforms (launch_gemm_nt / _nn / _pair, kernels.h) -- a translation unit of its own:
// the kernel is tuned on its ISA, and kernels.cc takes two minutes to compile.  HIP build only (the host emulation of
// these launchers is in kernels.cc).
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>

#ifndef RPDE_EMU
namespace rpde {

// ------------------------------------------------------------------------------- f64 MFMA GEMM
// C (M x N) = A (M x K, k contiguous) * B, with B either (N x K, k contiguous)  [NN = false]
// or (K x N, n contiguous) [NN = true].  128 x 128 x 16 block tile, 4 waves in a 2 x 2 grid,
// each wave 4 x 4 tiles of v_mfma_f64_16x16x4_f64.  LDS layout [k/4][row][k%4] makes every
// fragment read one contiguous 512-byte ds_read_b64 per wave (conflict free).
typedef double dbl4 __attribute__((ext_vector_type(4)));
typedef double dbl2v __attribute__((ext_vector_type(2)));
// D = A B + D with the accumulator PINNED TO VGPRs.  Measured on MI355X (rpde_microbench mfma_peak /
// mfma_peak_a): v_mfma_f64_16x16x4_f64 runs at 77.4 TFLOP/s with a VGPR accumulator and at 38.3 with an
// AGPR one; left to itself the register allocator parks part of the 128 accumulator registers in AGPRs.
// Inline asm is opaque to the hazard recognizer: readers of the accumulators outside the MFMA stream wait
// explicitly (mfma_drain).
__device__ __forceinline__ void mfma_f64_vgpr(dbl4& acc, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
constexpr int kGemmLdsDefault = 1;          // LDS layout of the operand stages (gemm_f64_db_tile LAY), one greppable line: flipped by measurement
constexpr long kGemmSmallTileBelow = 512;   // fewer 128 x 128 tiles than this (two per CU): use 64 x 64 tiles

// two independent products in one launch (blockIdx.z picks): the even and the odd block of the Poisson
// eigen-transforms.  512 tiles are exactly one round on 256 CUs with two workgroups each -- prologue, epilogue
// and the store burst of a round are not hidden; 1024 tiles give every CU a second round to overlap them with
struct GemmArgs { int M, N, K; const double* A; long lda; const double* B; long ldb; double* C; long ldc; bool ct; bool z00; };
// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: in the linear
// order an XCD works on a few n-tiles of EVERY m-tile, so each of the 8 L2s pulls the whole A operand (measured:
// 2.4 x the operand bytes).  With rx * ry = 8 rectangular regions of bw x bh tiles, one per XCD, the panels an L2
// has to hold are bw + bh instead of gx / 8 + gy (32 x 16 tiles: 8 + 8 instead of 4 + 16).  bw == 0: linear order.
struct GemmSwizzle { int rx = 0, bw = 0, bh = 0; };
static GemmSwizzle gemm_swizzle(int gx, int gy) {
  static const bool on = [] { const char* e = std::getenv("RPDE_GEMM_SWIZZLE"); return !e || std::atoi(e) != 0; }();
  GemmSwizzle best;
  if (!on || (gx * gy) % 8) return best;
  int cost = 1 << 30;
  for (int rx = 1; rx <= 8; rx *= 2) {
    const int ry = 8 / rx;
    if (gx % rx || gy % ry) continue;
    if (gx / rx + gy / ry < cost) { cost = gx / rx + gy / ry; best = GemmSwizzle{rx, gx / rx, gy / ry}; }
  }
  return best;
}
__device__ __forceinline__ void gemm_tile_of_block(const GemmSwizzle& z, int& tx, int& ty) {
  tx = (int)blockIdx.x; ty = (int)blockIdx.y;
  if (z.bw == 0) return;
  const int l = tx + (int)gridDim.x * ty, xcd = l & 7, s = l >> 3;
  tx = (xcd % z.rx) * z.bw + s % z.bw;
  ty = (xcd / z.rx) * z.bh + s / z.bw;
}
template <bool NN, int DB>
__global__ __launch_bounds__(256) void gemm_f64_pair_kernel(const GemmArgs g0, const GemmArgs g1, const GemmSwizzle z);

// Variant 4: the same tiling with the pipeline written out.  Two LDS stages (64 KB): the operands of stage
// t + 1 are written while stage t feeds the MFMAs, so one barrier per stage instead of two; the fragments of
// k sub-step s + 1 are read from LDS before the 16 MFMAs of sub-step s issue; the global loads of stage t + 2
// are in flight during all of stage t + 1.
// TM x TM block tile (128 or 64), 4 waves in a 2 x 2 grid, each wave MT x MT MFMA tiles (MT = TM / 32).  The
// 64-tile serves problems that would not give every CU a 128-tile (small grids; the local products of a
// pencil-sharded run, 512 x 2048 x 2048 per GPU at 4097^2 on 8 GPUs).
// LAY: the LDS layout of an operand stage.  0 (rounds 1 - 4): [k/4][row][k%4] -- a fragment (lane l: row l % 16, k l / 16) is
// one contiguous 512-byte block, but the compiler fetches two fragments per ds_read2st64_b64, which the LDS serves in groups
// of 16 lanes over 32 banks: the 16 rows of a group are 32 bytes apart, a 4-way bank conflict on every fragment read
// (SQ_LDS_BANK_CONFLICT: 70 % of the kernel's LDS cycles, profiles/r05_lds_counters.txt).  1 (round 5): [k/4][k%4][row]
// with 144 doubles per k plane -- the 16 rows of a lane group are 128 contiguous bytes, the next k plane starts 32 banks
// further (conflict free also as a 32-lane ds_read_b64); the staging stores become 8-byte stores (two lanes of a row hit
// the same bank: the k/4 planes are 580 doubles apart, which puts them 16 banks apart).
// WAVES: 4 (2 x 2 waves of TM/2 x TM/2) or -- TM = 128 only -- 8 (2 x 4 waves of 64 x 32: half the accumulators per wave, 128 VGPRs,
// so that the two workgroups of a CU put FOUR waves on a SIMD instead of two; round 6, the pair launcher's default)
// CTS (round 6, the transposed store of G2): the product is accumulated TRANSPOSED -- the MFMA takes the B fragment as its first and
// the A fragment as its second operand, so a lane holds C(m = lane % 16, n = lane / 16 + 4 r) and the 16 lanes of a register write
// 128 contiguous bytes of row n of C^T, like the plain store does for C (the runtime `ct` store of the untransposed accumulators
// writes 32-byte pieces: four times the write requests).  Every element accumulates the same products in the same order (a b = b a).
template <bool NN, int TM, bool OLD = true, int LAY = 0, int WAVES = 4, bool CTS = false>
__device__ __forceinline__ void gemm_f64_db_tile(int M, int N, int K, const double* __restrict__ A, long lda,
                                                 const double* __restrict__ B, long ldb, double* __restrict__ C, long ldc,
                                                 int tile_m, int tile_n, bool ct = false, bool z00 = false) {
  constexpr int BK = 16, KS = 4;
  constexpr int NT = 64 * WAVES;       // threads
  constexpr int WN = WAVES / 2;        // waves along n (2 or 4); two along m
  constexpr int MT = TM / 32;          // MFMA tiles per wave along m
  constexpr int MTN = TM / (16 * WN);  // ... along n
  constexpr int TPR = NT / TM;         // threads per operand row (k contiguous)
  constexpr int KT = BK / TPR;         // doubles per thread and operand per stage (8 or 4)
  constexpr int VPE = 4 * TM / NT;     // NN: k-values per thread and k sub-step (2 or 1)
  static_assert(TM == 128 || TM == 64, "tile sizes 128 and 64");
  static_assert(WAVES == 4 || (WAVES == 8 && TM == 128), "4 waves, or 8 waves on the 128-tile");
  constexpr int RS = LAY ? TM + 16 : 4;              // LAY 1: doubles per k plane; LAY 0: per row
  constexpr int SS = LAY ? 4 * RS + 4 : 4 * TM;      // doubles per k/4 block
  __shared__ __attribute__((aligned(16))) double Asf[2 * KS * SS];
  __shared__ __attribute__((aligned(16))) double Bsf[2 * KS * SS];
  auto at = [](int st, int s, int row, int kk) { return LAY ? (st * KS + s) * SS + kk * RS + row : (st * KS + s) * SS + row * 4 + kk; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = tile_m * TM, n0 = tile_n * TM;
  const int l15 = lane & 15, l4 = lane >> 4;
  dbl4 acc[MT][MTN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MTN; ++j) acc[i][j] = dbl4{0.0, 0.0, 0.0, 0.0};
  double ra[KT], rb[KT];
  const int arow = tid / TPR, akk = (tid % TPR) * KT;
  const int bn = tid % TM, bkg = tid / TM;                                   // NN: column n, k group
  const bool vec16 = ((lda | ldb) & 1) == 0 && (((size_t)A | (size_t)B) & 15) == 0;
  // edge tiles read the last valid row / column again (results never stored): no predicated loads
  const double* pa = A + (long)min(m0 + arow, M - 1) * lda + akk;
  const double* pbt = B + (long)min(n0 + arow, N - 1) * ldb + akk;          // !NN
  const double* pbn = B + (long)(VPE * bkg) * ldb + min(n0 + bn, N - 1);     // NN: k = 4 e + VPE bkg + v

  // Full stages (k0 + BK <= K) of aligned operands load without a bounds test, from a UNIFORM base pointer (scalar
  // registers, advanced by scalar adds) plus one 32-bit byte offset per thread: `global_load v, v_off, s[base]` -- no 64-bit
  // vector address arithmetic and no branch in the steady-state loop (round 5; before: a vector add per load and the
  // bounds-tested form of the partial last stage inlined behind a branch in every stage, 375 instructions around 64 MFMAs).
  // The launchers check that the operands stay below 2 GB (32-bit byte offsets).
  const char* ab8 = reinterpret_cast<const char*>(A);
  const char* bb8 = reinterpret_cast<const char*>(B);
  const unsigned offa = 8u * (unsigned)(min(m0 + arow, M - 1) * (int)lda + akk);
  const unsigned offbt = 8u * (unsigned)(min(n0 + arow, N - 1) * (int)ldb + akk);               // !NN
  const unsigned offbn = 8u * (unsigned)(VPE * bkg * (int)ldb + min(n0 + bn, N - 1));           // NN
  const size_t ldb8 = (size_t)ldb * 8;
  auto gload_fast = [&](int k0) {
    using gp = const __attribute__((address_space(1))) char*;
    gp pa8 = (gp)(ab8 + (size_t)k0 * 8);
#pragma unroll
    for (int e = 0; e < KT / 2; ++e) {
      const dbl2v v = *reinterpret_cast<const __attribute__((address_space(1))) dbl2v*>(pa8 + offa + 16 * e);
      ra[2 * e] = v.x; ra[2 * e + 1] = v.y;
    }
    if constexpr (!NN) {
      gp pb8 = (gp)(bb8 + (size_t)k0 * 8);
#pragma unroll
      for (int e = 0; e < KT / 2; ++e) {
        const dbl2v v = *reinterpret_cast<const __attribute__((address_space(1))) dbl2v*>(pb8 + offbt + 16 * e);
        rb[2 * e] = v.x; rb[2 * e + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < KS; ++e)
#pragma unroll
        for (int v = 0; v < VPE; ++v) {
          gp q8 = (gp)(bb8 + (size_t)(k0 + 4 * e + v) * ldb8);     // uniform: row k0 + 4 e + v (+ VPE bkg in the offset)
          rb[VPE * e + v] = *reinterpret_cast<const __attribute__((address_space(1))) double*>(q8 + offbn);
        }
    }
  };
  auto gload = [&](int k0) {
    if (k0 + BK <= K && vec16) {
      if constexpr (!OLD) { gload_fast(k0); return; }
      const dbl2v* p = reinterpret_cast<const dbl2v*>(pa + k0);
#pragma unroll
      for (int e = 0; e < KT / 2; ++e) { const dbl2v v = p[e]; ra[2 * e] = v.x; ra[2 * e + 1] = v.y; }
      if constexpr (!NN) {
        const dbl2v* q = reinterpret_cast<const dbl2v*>(pbt + k0);
#pragma unroll
        for (int e = 0; e < KT / 2; ++e) { const dbl2v v = q[e]; rb[2 * e] = v.x; rb[2 * e + 1] = v.y; }
      } else {
        const double* q = pbn + (long)k0 * ldb;
#pragma unroll
        for (int e = 0; e < KS; ++e)
#pragma unroll
          for (int v = 0; v < VPE; ++v) rb[VPE * e + v] = q[(long)(4 * e + v) * ldb];
      }
      return;
    }
    {
      const int r = m0 + arow;
#pragma unroll
      for (int e = 0; e < KT; ++e) ra[e] = (r < M && k0 + akk + e < K) ? pa[k0 + e] : 0.0;
    }
    if constexpr (!NN) {
      const int r = n0 + arow;
#pragma unroll
      for (int e = 0; e < KT; ++e) rb[e] = (r < N && k0 + akk + e < K) ? pbt[k0 + e] : 0.0;
    } else {
      const int n = n0 + bn;
#pragma unroll
      for (int e = 0; e < KS; ++e)
#pragma unroll
        for (int v = 0; v < VPE; ++v) {
          const int k = k0 + 4 * e + VPE * bkg + v;
          rb[VPE * e + v] = (k < K && n < N) ? B[(long)k * ldb + n] : 0.0;
        }
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int h = 0; h < KT / 4; ++h) {
      if constexpr (LAY) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) Asf[at(st, (akk >> 2) + h, arow, kk)] = ra[4 * h + kk];
      } else {
        dbl2v* d = reinterpret_cast<dbl2v*>(&Asf[at(st, (akk >> 2) + h, arow, 0)]);
        d[0] = dbl2v{ra[4 * h], ra[4 * h + 1]};
        d[1] = dbl2v{ra[4 * h + 2], ra[4 * h + 3]};
      }
    }
    if constexpr (!NN) {
#pragma unroll
      for (int h = 0; h < KT / 4; ++h) {
        if constexpr (LAY) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) Bsf[at(st, (akk >> 2) + h, arow, kk)] = rb[4 * h + kk];
        } else {
          dbl2v* d = reinterpret_cast<dbl2v*>(&Bsf[at(st, (akk >> 2) + h, arow, 0)]);
          d[0] = dbl2v{rb[4 * h], rb[4 * h + 1]};
          d[1] = dbl2v{rb[4 * h + 2], rb[4 * h + 3]};
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < KS; ++e) {
        if constexpr (VPE == 2) {
          if constexpr (LAY) { Bsf[at(st, e, bn, 2 * bkg)] = rb[2 * e]; Bsf[at(st, e, bn, 2 * bkg + 1)] = rb[2 * e + 1]; }
          else *reinterpret_cast<dbl2v*>(&Bsf[at(st, e, bn, 2 * bkg)]) = dbl2v{rb[2 * e], rb[2 * e + 1]};
        } else Bsf[at(st, e, bn, bkg)] = rb[e];
      }
    }
  };
  auto frag = [&](int st, int s, double (&a)[MT], double (&b)[MTN]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = Asf[at(st, s, wm * (TM / 2) + i * 16 + l15, l4)];
#pragma unroll
    for (int j = 0; j < MTN; ++j) b[j] = Bsf[at(st, s, wn * (TM / WN) + j * 16 + l15, l4)];
  };
  auto mma = [&](const double (&a)[MT], const double (&b)[MTN]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < MTN; ++j) {
        if constexpr (CTS) mfma_f64_vgpr(acc[i][j], b[j], a[i]);
        else mfma_f64_vgpr(acc[i][j], a[i], b[j]);
      }
  };

  gload(0);
  lstore(0);
  if (BK < K) gload(BK);
  __syncthreads();
  int st = 0, k0 = 0;
  // the barrier of a stage sits in front of its LAST group of MFMAs, and the first fragments of the next stage are read
  // behind it: the barrier and the LDS latency of a stage change hide under 16 MFMAs instead of idling the pipe
  {
    double a0[MT], b0[MTN], a1[MT], b1[MTN];
    frag(0, 0, a0, b0);
    // steady state: stage t stores the operands of stage t + 1 and loads those of stage t + 2, all three full stages
    if (!OLD && vec16)   // (OLD: the round-4 loop -- every stage through the bounds-tested form below; the default, see launch_gemm_pair)
      for (; k0 + 3 * BK <= K; k0 += BK, st ^= 1) {
        frag(st, 1, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        lstore(st ^ 1);
        frag(st, 2, a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        gload_fast(k0 + 2 * BK);
        frag(st, 3, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                          // stage t + 1 is complete; every read of stage t has been issued
        frag(st ^ 1, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    // the last two stages (and everything, for unaligned operands): bounds-tested loads, nothing to load behind the end
    for (; k0 < K; k0 += BK, st ^= 1) {
      const bool more = k0 + BK < K;
      frag(st, 1, a1, b1);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) lstore(st ^ 1);
      frag(st, 2, a0, b0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (k0 + 2 * BK < K) gload(k0 + 2 * BK);   // (one MFMA group earlier, right behind the LDS stores: 1.5 % slower, round 5 call 5)
      frag(st, 3, a1, b1);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      if (more) frag(st ^ 1, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mfma_drain();
  if constexpr (CTS) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < MTN; ++j) {
        const int m = m0 + wm * (TM / 2) + i * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * (TM / WN) + j * 16 + l4 + 4 * r;
          if (m < M && n < N) C[(long)n * ldc + m] = (z00 && (m | n) == 0) ? 0.0 : acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MTN; ++j) {
      const int n = n0 + wn * (TM / WN) + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (TM / 2) + i * 16 + l4 + 4 * r;
        // ct: C^T -- the four lanes of a column write 32 contiguous bytes of row n, the four registers the 128 bytes
        // z00: the element (0, 0) leaves as 0 (GemmProblem::zero00 -- pseu[0, 0] = 0 without a launch of its own)
        if (m < M && n < N) C[ct ? (long)n * ldc + m : (long)m * ldc + n] = (z00 && (m | n) == 0) ? 0.0 : acc[i][j][r];
      }
    }
}

template <bool NN, int TM, int LAY>
__global__ __launch_bounds__(256) void gemm_f64_db_kernel(int M, int N, int K,
                                                          const double* __restrict__ A, long lda,
                                                          const double* __restrict__ B, long ldb,
                                                          double* __restrict__ C, long ldc) {
  gemm_f64_db_tile<NN, TM, true, LAY>(M, N, K, A, lda, B, ldb, C, ldc, (int)blockIdx.y, (int)blockIdx.x);
}
// RPDE_GEMM_LDS=0 | 1 (A/B): the LDS layout of the operand stages (gemm_f64_db_tile LAY); kGemmLdsDefault is set by measurement
static bool gemm_lay1() {
  static const bool on = [] { const char* e = std::getenv("RPDE_GEMM_LDS"); return e ? std::atoi(e) != 0 : kGemmLdsDefault != 0; }();
  return on;
}
// DB: 1 = 128-tile, 2 = 64-tile (both two-stage)
template <bool NN, int DB>
__global__ __launch_bounds__(256) void gemm_f64_pair_kernel(const GemmArgs g0, const GemmArgs g1, const GemmSwizzle z) {
  constexpr int TM = (DB == 2 || DB == 5) ? 64 : 128;   // DB 3 / 4: the 128-tile with the round-4 loop, LDS layout 0 / 1 (DB 1: the peeled loop, A/B); DB 2 / 5: the 64-tile
  int tx, ty;
  gemm_tile_of_block(z, tx, ty);
  const GemmArgs& g = blockIdx.z ? g1 : g0;
  if (ty * TM >= g.M || tx * TM >= g.N) return;
  if constexpr (DB == 2) gemm_f64_db_tile<NN, 64>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
  else if constexpr (DB == 3) gemm_f64_db_tile<NN, 128, true>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
  else if constexpr (DB == 4) gemm_f64_db_tile<NN, 128, true, 1>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);   // LDS layout 1
  else if constexpr (DB == 5) gemm_f64_db_tile<NN, 64, true, 1>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
  else gemm_f64_db_tile<NN, 128, false>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
}

// The 128-tile by eight waves of 64 x 32 (four waves per SIMD with two workgroups per CU): the default of the pair launcher since
// round 6 (RPDE_GEMM_WAVES=4: the four-wave kernel, A/B)
template <bool NN, bool CTS = false>
__global__ __launch_bounds__(512, 4) void gemm_f64_pair8_kernel(const GemmArgs g0, const GemmArgs g1, const GemmSwizzle z) {
  const GemmArgs& g = blockIdx.z ? g1 : g0;
  int tx, ty;
  gemm_tile_of_block(z, tx, ty);
  if (ty * 128 >= g.M || tx * 128 >= g.N) return;
  gemm_f64_db_tile<NN, 128, true, 1, 8, CTS>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
}

// RPDE_GEMM_PERSIST=1 (A/B, round 6): a workgroup works off its tile of BOTH problems one after the other instead of the two
// problems being two rounds of workgroups -- no drain / relaunch between the rounds.  A kernel of its own: inside the pair kernel
// the loop cost 90 more registers (315: one workgroup per CU).
template <bool NN>
__global__ __launch_bounds__(256, 2) void gemm_f64_pair_persist_kernel(const GemmArgs g0, const GemmArgs g1, const GemmSwizzle z) {
  int tx, ty;
  gemm_tile_of_block(z, tx, ty);
#pragma unroll 1
  for (int zz = 0; zz < 2; ++zz) {
    const GemmArgs& g = zz ? g1 : g0;
    if (!(ty * 128 >= g.M || tx * 128 >= g.N))
      gemm_f64_db_tile<NN, 128, true, 1>(g.M, g.N, g.K, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, ty, tx, g.ct, g.z00);
    __syncthreads();
  }
}

// the steady-state loop addresses its operands with 32-bit byte offsets from a scalar base (gemm_f64_db_tile)
template <bool NN>
static void gemm_require_32bit(int M, int N, int K, long lda, long ldb) {
  const long lim = (1L << 31) / 8;
  const long ea = (long)(M - 1) * lda + K, eb = NN ? (long)(K - 1) * ldb + N : (long)(N - 1) * ldb + K;
  RPDE_REQUIRE(ea < lim && eb < lim, "gemm: an operand of 2 GB or more (32-bit byte offsets in the kernel)");
}
template <bool NN>
static void launch_gemm(int M, int N, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                        Stream& st) {
  if (M <= 0 || N <= 0) return;
  gemm_require_32bit<NN>(M, N, K, lda, ldb);
  dim3 grid((N + 127) / 128, (M + 127) / 128);
  if ((long)grid.x * grid.y < kGemmSmallTileBelow) {   // too few 128-tiles for the chip
    dim3 g64((N + 63) / 64, (M + 63) / 64);
    if (gemm_lay1()) hipLaunchKernelGGL((gemm_f64_db_kernel<NN, 64, 1>), g64, dim3(256), 0, st.s, M, N, K, A, lda, B, ldb, C, ldc);
    else hipLaunchKernelGGL((gemm_f64_db_kernel<NN, 64, 0>), g64, dim3(256), 0, st.s, M, N, K, A, lda, B, ldb, C, ldc);
    RPDE_HIP(hipGetLastError());
    return;
  }
  if (gemm_lay1()) hipLaunchKernelGGL((gemm_f64_db_kernel<NN, 128, 1>), grid, dim3(256), 0, st.s, M, N, K, A, lda, B, ldb, C, ldc);
  else hipLaunchKernelGGL((gemm_f64_db_kernel<NN, 128, 0>), grid, dim3(256), 0, st.s, M, N, K, A, lda, B, ldb, C, ldc);
  RPDE_HIP(hipGetLastError());
}
void launch_gemm_pair(bool nn, const GemmProblem& p0, const GemmProblem& p1, Stream& st) {
  if (p0.M <= 0 || p0.N <= 0 || p1.M <= 0 || p1.N <= 0) {
    RPDE_REQUIRE(!(p0.ct || p1.ct || p0.zero00 || p1.zero00), "transposed store / zeroed element: both problems must be non-empty");
    for (const GemmProblem* p : {&p0, &p1}) {
      if (nn) launch_gemm<true>(p->M, p->N, p->K, p->A, p->lda, p->B, p->ldb, p->C, p->ldc, st);
      else launch_gemm<false>(p->M, p->N, p->K, p->A, p->lda, p->B, p->ldb, p->C, p->ldc, st);
    }
    return;
  }
  for (const GemmProblem* p : {&p0, &p1}) {
    if (nn) gemm_require_32bit<true>(p->M, p->N, p->K, p->lda, p->ldb);
    else gemm_require_32bit<false>(p->M, p->N, p->K, p->lda, p->ldb);
  }
  const GemmArgs g0{p0.M, p0.N, p0.K, p0.A, p0.lda, p0.B, p0.ldb, p0.C, p0.ldc, p0.ct, p0.zero00}, g1{p1.M, p1.N, p1.K, p1.A, p1.lda, p1.B, p1.ldb, p1.C, p1.ldc, p1.ct, p1.zero00};
  const int Mx = std::max(p0.M, p1.M), Nx = std::max(p0.N, p1.N);
  dim3 grid((Nx + 127) / 128, (Mx + 127) / 128, 2);
  const bool lay1 = gemm_lay1();
  if ((long)grid.x * grid.y * 2 < kGemmSmallTileBelow) {
    dim3 g64((Nx + 63) / 64, (Mx + 63) / 64, 2);
    const GemmSwizzle z = gemm_swizzle((int)g64.x, (int)g64.y);
    if (lay1) {
      if (nn) hipLaunchKernelGGL((gemm_f64_pair_kernel<true, 5>), g64, dim3(256), 0, st.s, g0, g1, z);
      else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, 5>), g64, dim3(256), 0, st.s, g0, g1, z);
    } else if (nn) hipLaunchKernelGGL((gemm_f64_pair_kernel<true, 2>), g64, dim3(256), 0, st.s, g0, g1, z);
    else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, 2>), g64, dim3(256), 0, st.s, g0, g1, z);
  } else {
    const GemmSwizzle z = gemm_swizzle((int)grid.x, (int)grid.y);
    // RPDE_GEMM_PEEL=1 (A/B): the steady-state loop without bounds tests and 64-bit vector addresses (gemm_f64_db_tile<.., OLD =
    // false>).  Measured in round 5 (profiles/r05_experiments): G1 / G2 1.081 / 1.100 ms with it, 1.071 / 1.092 without -- the
    // 375 instructions around the 64 MFMAs of a stage were never what kept the pipe at 0.81; the round-4 loop stays the default
    static const bool peel = [] { const char* e = std::getenv("RPDE_GEMM_PEEL"); return e && std::atoi(e) != 0; }();
    static const bool persist = [] { const char* e = std::getenv("RPDE_GEMM_PERSIST"); return e && std::atoi(e) != 0; }();
    // eight waves per 128-tile: the default since round 6 (G1 / G2 1.097 / 1.145 -> 1.045 / 1.083 ms, three alternating runs in one
    // call, profiles/r06_experiments/call9_ab_gemm_waves.txt; MFMA pipe 92 % -> 96 % busy); RPDE_GEMM_WAVES=4: four waves (A/B)
    static const bool waves8 = [] { const char* e = std::getenv("RPDE_GEMM_WAVES"); return !e || std::atoi(e) != 4; }();
    // both products stored transposed (G2): accumulate them transposed (gemm_f64_db_tile CTS); RPDE_GEMM_CTSWAP=0: the 32-byte store (A/B)
    static const bool ctswap = [] { const char* e = std::getenv("RPDE_GEMM_CTSWAP"); return !e || std::atoi(e) != 0; }();
    if (!peel && lay1 && waves8 && !persist) {
      if (ctswap && p0.ct && p1.ct) {
        if (nn) hipLaunchKernelGGL((gemm_f64_pair8_kernel<true, true>), grid, dim3(512), 0, st.s, g0, g1, z);
        else hipLaunchKernelGGL((gemm_f64_pair8_kernel<false, true>), grid, dim3(512), 0, st.s, g0, g1, z);
      } else if (nn) hipLaunchKernelGGL((gemm_f64_pair8_kernel<true>), grid, dim3(512), 0, st.s, g0, g1, z);
      else hipLaunchKernelGGL((gemm_f64_pair8_kernel<false>), grid, dim3(512), 0, st.s, g0, g1, z);
    } else if (!peel && lay1 && persist) {
      grid.z = 1;
      if (nn) hipLaunchKernelGGL((gemm_f64_pair_persist_kernel<true>), grid, dim3(256), 0, st.s, g0, g1, z);
      else hipLaunchKernelGGL((gemm_f64_pair_persist_kernel<false>), grid, dim3(256), 0, st.s, g0, g1, z);
    } else if (!peel && lay1) {
      if (nn) hipLaunchKernelGGL((gemm_f64_pair_kernel<true, 4>), grid, dim3(256), 0, st.s, g0, g1, z);
      else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, 4>), grid, dim3(256), 0, st.s, g0, g1, z);
    } else if (!peel) {
      if (nn) hipLaunchKernelGGL((gemm_f64_pair_kernel<true, 3>), grid, dim3(256), 0, st.s, g0, g1, z);
      else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, 3>), grid, dim3(256), 0, st.s, g0, g1, z);
    } else if (nn) hipLaunchKernelGGL((gemm_f64_pair_kernel<true, 1>), grid, dim3(256), 0, st.s, g0, g1, z);
    else hipLaunchKernelGGL((gemm_f64_pair_kernel<false, 1>), grid, dim3(256), 0, st.s, g0, g1, z);
  }
  RPDE_HIP(hipGetLastError());
}
void launch_gemm_nt(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream& st) {
  launch_gemm<false>(M, N, K, A, lda, B, ldb, C, ldc, st);
}
void launch_gemm_nn(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream& st) {
  launch_gemm<true>(M, N, K, A, lda, B, ldb, C, ldc, st);
}


}  // namespace rpde
#endif
