"""Parity diagnostics at the headline size (GPU): where do engine and oracle differ, and by how much?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R
from tests import checks as K
from oracle import solver as S, navier as N

lib = R.lib()
rel = K.rel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
sp, osp = K.spaces(lib, "cheb_dirichlet", n, "cheb_dirichlet", 65)
rng = np.random.default_rng(1)
rhs = rng.standard_normal(osp.shape_ortho)
k = np.arange(n)[:, None]
smooth = np.exp(-k / 40.0) * rng.standard_normal(osp.shape_ortho)
for c in ([2e-8, 2e-8], [5.7e-8, 5.7e-8], [1e-5, 1e-5], [1e-3, 1e-3]):
    h, oh = R.HholtzAdi(sp, c), S.HholtzAdi(osp, c)
    print(f"HholtzAdi n={n} c={c[0]:g}: random rhs {rel(h.solve(rhs), oh.solve(rhs)):.2e}   smooth rhs {rel(h.solve(smooth), oh.solve(smooth)):.2e}", flush=True)
vh = osp.forward(rng.standard_normal(osp.shape_physical))
print("from_ortho", rel(sp.from_ortho(osp.to_ortho(vh)), osp.from_ortho(osp.to_ortho(vh))), "gradient x", rel(sp.gradient(vh, [1, 0]), osp.gradient(vh, [1, 0])), flush=True)
sp, osp = K.spaces(lib, "cheb_neumann", n, "cheb_neumann", 65)
rhs = rng.standard_normal(osp.shape_ortho)
p = R.Poisson(sp, [1.0, 1.0])
t0 = time.time(); ind = S.Poisson(osp, [1.0, 1.0], eig_mode="parity"); t1 = time.time()
sh = S.Poisson(osp, [1.0, 1.0], eig_override=p.eigenbasis())
got = p.solve(rhs)
print(f"Poisson n={n}: independent basis {rel(got, ind.solve(rhs)):.2e}   shared basis {rel(got, sh.solve(rhs)):.2e}   (oracle eig {t1-t0:.1f}s)", flush=True)
del p, ind, sh
# steps
nav = R.Navier2D.new_confined(n, n, 1e8, 1.0, 2e-4, 1.0, "rbc")
ora = N.Navier2D.new_confined(n, n, 1e8, 1.0, 2e-4, 1.0, "rbc", eig_override=nav.poisson_eigenbasis())
for z in (nav, ora):
    z.set_velocity(0.2, 1.0, 1.0); z.set_temperature(0.2, 1.0, 1.0)
for s in range(1, 4):
    nav.update(); ora.update()
    a, b = nav.physical_fields(), ora.physical_fields()
    print(f"step {s} (shared basis):", {k: f"{rel(a[k], b[k]):.2e}" for k in b}, " pseu:", f"{rel(nav.pseu.vhat, ora.pseu.vhat):.2e}", flush=True)
