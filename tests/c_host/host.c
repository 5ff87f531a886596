/* A plain C99 host of the C ABI (include/rustpde_hip.h): what a C caller -- or the Rust FFI
 * sketched in INTEGRATION.md -- does.  Build: gcc -std=c99 -I include host.c -L<dir> -l<lib>.
 * Usage: host <nx> <ny> <steps>; prints "time <t> sum <sum of T> sumsq <sum of T^2> exit <flag>".
 * Test infrastructure: tests/test_c_host.py compiles it against the emulation build (CPU) and,
 * on a GPU box, against librustpde_hip.so, and compares with the Python mirror. */
#include <stdio.h>
#include <stdlib.h>

#include "rustpde_hip.h"

#define CHECK(call)                                                         \
  do {                                                                      \
    if ((call) != 0) {                                                      \
      fprintf(stderr, "%s failed: %s\n", #call, rpde_last_error());         \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: host nx ny steps\n"); return 2; }
  const int nx = atoi(argv[1]), ny = atoi(argv[2]), steps = atoi(argv[3]);
  rpde_navier2d* nav = NULL;
  CHECK(rpde_navier2d_create_confined(nx, ny, 1e5, 1.0, 0.01, 1.0, "rbc", 0, &nav));
  CHECK(rpde_navier2d_set_velocity(nav, 0.2, 1.0, 1.0));
  CHECK(rpde_navier2d_set_temperature(nav, 0.2, 1.0, 1.0));
  CHECK(rpde_navier2d_update(nav, steps));
  double t = 0.0;
  int flag = -1;
  CHECK(rpde_navier2d_time(nav, &t));
  CHECK(rpde_navier2d_exit(nav, &flag));
  const size_t n = (size_t)nx * (size_t)ny;
  double* temp = (double*)malloc(n * sizeof(double));
  if (!temp) return 3;
  CHECK(rpde_navier2d_get_field(nav, "temp", RPDE_PHYSICAL, temp, n));
  double s = 0.0, s2 = 0.0;
  for (size_t i = 0; i < n; ++i) { s += temp[i]; s2 += temp[i] * temp[i]; }
  /* misuse must come back as an error code with a message, never as a crash */
  if (rpde_navier2d_get_field(nav, "temp", RPDE_PHYSICAL, temp, n - 1) == 0) return 4;
  if (rpde_navier2d_get_field(nav, "no such field", RPDE_PHYSICAL, temp, n) == 0) return 5;
  printf("time %.17g sum %.17g sumsq %.17g exit %d\n", t, s, s2, flag);
  free(temp);
  CHECK(rpde_navier2d_destroy(nav));
  return 0;
}
