// Microbenchmark (round 6): can the f64 VALU (v_fma_f64) run beside the f64 MFMA pipe (v_mfma_f64_16x16x4_f64) on gfx950?
// The two have the same nominal rate (78.6 TFLOP/s each at 2.4 GHz); if they are separate hardware, a GEMM could put part of
// its tile on the vector unit.  Per wave: NM MFMAs on independent accumulators followed by NV v_fma_f64 on independent
// registers, `iters` times; the register-only loop has no memory traffic.  Reported: shader cycles per iteration (s_memtime
// of wave 0 of workgroup 0, and the mean over all waves), wall time, TFLOP/s of each pipe.
//   hipcc --offload-arch=gfx950 -O3 -o tools/native/bin/f64_coissue tools/native/f64_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double dbl4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// MODE 0: every wave runs NM MFMAs + NV FMAs per iteration (interleaved by the order written: MFMAs first, the FMAs issue in their shadow).
// MODE 1: even waves of a workgroup run the MFMAs only, odd waves the FMAs only (wave w and w + 4 share a SIMD with 512 threads).
template <int NM, int NV, int MODE>
__global__ __launch_bounds__(512) void mix_kernel(double* out, long long* cyc, int iters) {
  dbl4 acc[8];
  double f[16];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12 * (threadIdx.x + 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 16; ++j) f[j] = 1e-3 * j + b;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || (wave >> 2) % 2 == 0 || blockDim.x <= 256;
  const bool do_v = MODE == 0 || !do_m;
  const bool m_on = MODE == 0 ? true : ((wave >> 2) % 2 == 0);
  const bool v_on = MODE == 0 ? true : !m_on;
  (void)do_m; (void)do_v;
  long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {        // four groups: NM / 4 MFMAs then NV / 4 FMAs
#pragma unroll
        for (int j = 0; j < NM / 4; ++j)
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[(s * (NM / 4) + j) % 8]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV / 4; ++j)
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[(s * (NV / 4) + j) % 16]) : "v"(a), "v"(b));
      }
    }
  } else {
    if (m_on) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NM; ++j)
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j % 8]) : "v"(a), "v"(b));
      }
    }
    if (v_on) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[j % 16]) : "v"(a), "v"(b));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
  for (int j = 0; j < 16; ++j) s += f[j];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int NM, int NV, int MODE>
static void run(const char* label, int threads, int blocks, int iters, double* out, long long* cyc) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix_kernel<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters / 10 + 1);   // warm-up
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<NM, NV, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const int nw = blocks * (threads / 64);
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double mean = 0; long long mx = 0;
  for (long long c : h) { mean += (double)c; if (c > mx) mx = c; }
  mean /= nw;
  const double waves_m = MODE == 0 ? nw : nw / 2.0, waves_v = MODE == 0 ? nw : nw / 2.0;
  const double tf_m = waves_m * (double)iters * NM * 2048.0 / (best * 1e-3) / 1e12;
  const double tf_v = waves_v * (double)iters * NV * 128.0 / (best * 1e-3) / 1e12;
  // s_memtime / readcyclecounter ticks at a constant 100 MHz on this chip (not shader cycles): report the wall-derived figures
  printf("%-34s thr %4d blk %5d  NM %2d NV %2d  %8.3f ms  ticks/iter mean %8.2f max %8.2f  MFMA %6.2f TF  VALU %6.2f TF  sum %6.2f\n",
         label, threads, blocks, NM, NV, best, mean / iters, (double)mx / iters, tf_m, tf_v, tf_m + tf_v);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  double* out; long long* cyc;
  CK(hipMalloc(&out, sizeof(double) * 512 * 4096));
  CK(hipMalloc(&cyc, sizeof(long long) * 8 * 4096));
  // one wave per SIMD (256 threads, one workgroup per CU)
  run<8, 0, 0>("mfma only, 1 wave/SIMD", 256, 256, iters, out, cyc);
  run<0, 64, 0>("valu only, 1 wave/SIMD", 256, 256, iters, out, cyc);
  run<8, 16, 0>("mix 8:16, 1 wave/SIMD", 256, 256, iters, out, cyc);
  run<8, 32, 0>("mix 8:32, 1 wave/SIMD", 256, 256, iters, out, cyc);
  run<8, 64, 0>("mix 8:64, 1 wave/SIMD", 256, 256, iters, out, cyc);
  run<8, 128, 0>("mix 8:128, 1 wave/SIMD", 256, 256, iters, out, cyc);
  // two waves per SIMD
  run<8, 0, 0>("mfma only, 2 waves/SIMD", 512, 256, iters, out, cyc);
  run<0, 64, 0>("valu only, 2 waves/SIMD", 512, 256, iters, out, cyc);
  run<8, 32, 0>("mix 8:32, 2 waves/SIMD", 512, 256, iters, out, cyc);
  run<8, 64, 0>("mix 8:64, 2 waves/SIMD", 512, 256, iters, out, cyc);
  run<8, 128, 0>("mix 8:128, 2 waves/SIMD", 512, 256, iters, out, cyc);
  // specialised waves: one MFMA wave and one VALU wave per SIMD
  run<8, 32, 1>("split waves 8 | 32", 512, 256, iters, out, cyc);
  run<8, 64, 1>("split waves 8 | 64", 512, 256, iters, out, cyc);
  run<8, 128, 1>("split waves 8 | 128", 512, 256, iters, out, cyc);
  // four waves per SIMD (two workgroups of 512 per CU)
  run<8, 0, 0>("mfma only, 4 waves/SIMD", 512, 512, iters, out, cyc);
  run<8, 64, 0>("mix 8:64, 4 waves/SIMD", 512, 512, iters, out, cyc);
  run<8, 128, 0>("mix 8:128, 4 waves/SIMD", 512, 512, iters, out, cyc);
  return 0;
}
