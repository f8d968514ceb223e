import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from cmpt_eigenex_amd import capi
from oracle import cref
ctx = capi.Context()
n = 16; N = n**3
rowptr, col, val = cref.laplacian3d(n)
A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=8)
init = np.random.default_rng(0).standard_normal(N)
for kind, calls, mode in (("lanczos", 256, 0), ("lanczos", 256, 2), ("arnoldi", 250, 3), ("arnoldi", 250, 2)):
    b = capi.Basis(ctx, A, N, 260); b.configure(ortho_mode=mode)
    res = []
    for rep in range(3):
        b.clear(); b.upload(capi.VEC_W, init)
        if kind == "lanczos":
            b.lanczos_enqueue(calls); st, al, be = b.lanczos_state(); res.append(al.copy())
        else:
            b.arnoldi_enqueue(calls); st, H = b.arnoldi_state(); res.append(H.copy())
    assert all(np.array_equal(res[0], r) for r in res[1:]), (kind, mode)
    print(kind, calls, mode, "nvec", st.nvec, "ok")
    b.close()
print("done")
