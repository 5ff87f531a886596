// The stages of the PERIODIC step that are element-wise along x on the YX arrays (a spectral x-line = kx complex numbers,
// Navier2D::new_periodic, navier.rs:336-428): no transform, no recurrence -- only the y stencil (rows j and j - 2), a factor i k
// and sums.  Rounds 1 - 5 ran them as line programs (one workgroup stages a line in LDS, an op at a time); here every thread owns
// one complex number of one row: 16-byte loads and stores, nothing staged.
//   kPerDiv   S5  div = d/dx velx.to_ortho() + d/dy vely                      (`div`, navier_eq.rs:19-24; d/dy vely from C4)
//   kPerCorr  S8  velx -= d/dx (from_ortho_y part of pseu), vely -= (its d/dy part)   (`correct_velocity`, navier_eq.rs:117-125; y part from C7)
//   kPerPres  S9  pres += -nu div + pseu.to_ortho() / dt                      (`update_pres`, navier_eq.rs:137-143)
// Per element the arithmetic is that of the line programs they replace (OP_LOADX, OP_CIK, accumulating OP_LOAD, guarded OP_STORE).
#pragma once
#include "platform.h"

namespace rpde {

enum PerRowsKind : int { kPerDiv = 0, kPerCorr = 1, kPerPres = 2 };

struct PerRowsArgs {
  int kind;
  int nlines, line0;        // local rows, global index of local row 0 (pencil-sharded: the rows in front are halo rows)
  int kx;                   // complex numbers per row
  long ld;                  // doubles between rows (all arrays)
  const double* a0;         // kPerDiv: velx (composite y)      kPerCorr: x part's input (Y2)   kPerPres: pseudo-pressure (composite y)
  const double* a1;         // kPerDiv: d/dy vely               kPerCorr: y part (Y3)           kPerPres: div
  double* o0;               // kPerDiv: div                     kPerCorr: velx (in place)       kPerPres: pres (in place)
  double* o1;               //                                  kPerCorr: vely (in place)
  const double* low;        // y stencil S[j, j - 2] of a0's base (kPerDiv, kPerPres)
  int my;                   // rows of a0 (composite y)
  double s0, s1;            // kPerDiv: s0 = 1 / sx             kPerCorr: s0 = -1 / sx          kPerPres: s0 = 1 / dt, s1 = -nu
  int* nanflag;             // kPerCorr, kPerPres: raised when a NaN is stored (Integrate::exit, navier.rs:482-489); may be null
};

// complex number k of local row `line`
RPDE_HD inline void per_rows_point(const PerRowsArgs& a, int line, int k) {
  const int gl = line + a.line0;
  const long o = (long)line * a.ld + 2 * (long)k;
  bool bad = false;
  if (a.kind == kPerCorr) {
    const double f = a.s0 * (double)k;
    const double gr = a.a0[o], gi = a.a0[o + 1];
    const double ur = a.o0[o] + (-f * gi), ui = a.o0[o + 1] + f * gr;       // + (i k s0) g
    a.o0[o] = ur; a.o0[o + 1] = ui;
    const double vr = a.a1[o] + a.o1[o], vi = a.a1[o + 1] + a.o1[o + 1];
    a.o1[o] = vr; a.o1[o + 1] = vi;
    bad = (ur != ur) || (ui != ui) || (vr != vr) || (vi != vi);
  } else {
    // S_y a0: rows gl (if it exists in the composite space) and gl - 2 times the stencil coefficient
    const bool has0 = gl < a.my, has2 = gl >= 2 && gl - 2 < a.my;
    const double c2 = has2 ? a.low[gl - 2] : 0.0;
    const double zr = (has0 ? a.a0[o] : 0.0) + c2 * (has2 ? a.a0[o - 2 * a.ld] : 0.0);
    const double zi = (has0 ? a.a0[o + 1] : 0.0) + c2 * (has2 ? a.a0[o + 1 - 2 * a.ld] : 0.0);
    if (a.kind == kPerDiv) {
      const double f = a.s0 * (double)k;
      a.o0[o] = -f * zi + a.a1[o];
      a.o0[o + 1] = f * zr + a.a1[o + 1];
    } else {
      const double pr = a.s0 * zr + a.s1 * a.a1[o] + a.o0[o], pi = a.s0 * zi + a.s1 * a.a1[o + 1] + a.o0[o + 1];
      a.o0[o] = pr; a.o0[o + 1] = pi;
      bad = (pr != pr) || (pi != pi);
    }
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

}  // namespace rpde
