#include "kernels.h"
#include "dct_line.h"
#include "hdct_line.h"
#include "rhs_line.h"
namespace rpde {
#define LINE_OF_BLOCK const int chunk = (int)gridDim.x >> 3; const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3)
template <int N, int MODE> __global__ __launch_bounds__(N / 16, 4) void pure_transform(const DctLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS]; LINE_OF_BLOCK; if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0}; hdct_bwd_line<N, MODE>(blk, a);
}
template <int N> __global__ __launch_bounds__(N / 16, 3) void conv_term(const ConvLineArgs c) {
  __shared__ __attribute__((aligned(16))) double buf[DctGeom<N>::LDS]; LINE_OF_BLOCK; if (line >= c.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0}; conv_line<N>(blk, c);
}
template <int N, int WHICH> __global__ __launch_bounds__(N / 16, 3) void rhs_hholtz_x(const RhsLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS]; LINE_OF_BLOCK; if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0}; rhs_line<N, WHICH>(blk, a);
}
template __global__ void pure_transform<4096, -1>(const DctLineArgs);
template __global__ void pure_transform<4096, kHdctSten2>(const DctLineArgs);
template __global__ void pure_transform<4096, kHdctSten2 | kHdctDeriv>(const DctLineArgs);
template __global__ void conv_term<4096>(const ConvLineArgs);
template __global__ void rhs_hholtz_x<4096, 0>(const RhsLineArgs);
template __global__ void rhs_hholtz_x<4096, 1>(const RhsLineArgs);
}
