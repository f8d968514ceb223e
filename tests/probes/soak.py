"""Leak / stability soak: create and destroy contexts, operators (CSR, column-blocked, blocks, device-resident),
bases and solvers of every kind many times; device memory must return to where it started.
usage: python tests/probes/soak.py [rounds=40]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from cmpt_eigenex_amd import capi, solver
from oracle import cref

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
n = 20
N = n ** 3
rowptr, col, val = cref.laplacian3d(n)
sizes = [int(v) for v in rng.integers(4, 30, 60)]
blocks = {}
for q in range(len(sizes)):
    D = rng.standard_normal((sizes[q], sizes[q]))
    blocks[(q, q)] = (D + D.T) / 2
Nb = sum(sizes)
# an operator that takes the column-sorted row tiles (needs >= 2 input slices of 32768 columns)
Ns = 90_001
cnt = rng.integers(0, 9, Ns)
rps = np.zeros(Ns + 1, np.int64); np.cumsum(cnt, out=rps[1:])
cls = rng.integers(0, Ns, int(rps[-1]))
cls = cls[np.lexsort((cls, np.repeat(np.arange(Ns), cnt)))].astype(np.int32)
vls = rng.uniform(-1, 1, cls.size)
torch.cuda.init()
free0 = None
for r in range(rounds):
    shards = (1, 2, 3)[r % 3]
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=(None, 0, 3)[r % 3])
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(tolerance=1e-8, maxIterations=60, maxEigenvalues=3)
    es.compute(); es.compute()
    out = es.expWithLanczos(-0.3, N)
    ar = solver.ArnoldiEigenSolver()
    ar.setDeviceOperator(A).set(minIterations=20, maxIterations=20, maxEigenvalues=2)
    ar.compute()
    tr = solver.ThickRestartLanczosEigenSolver()
    tr.setDeviceOperator(A).set(numberOfEigenvalues=3, maxBasisSize=24, tolerance=1e-8)
    tr.compute()
    B = capi.Csr.upload_blocks(ctx, sizes, sizes, blocks)
    eb = solver.LanczosEigenSolver()
    eb.setDeviceOperator(B).set(minIterations=30, maxIterations=30)
    eb.compute()
    if shards == 1:
        t = [torch.from_numpy(a).cuda() for a in (rowptr, col, val)]
        D = capi.Csr.from_device(ctx, N, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())
        ed = solver.LanczosEigenSolver(); ed.setDeviceOperator(D).set(minIterations=10, maxIterations=10); ed.compute(); ed.close(); D.close()
        del t
    S = capi.Csr.upload(ctx, Ns, rps.astype(np.int32), cls, vls, column_blocks=-2)
    assert S.layout() == "sorted_tiles"
    sa = solver.ArnoldiEigenSolver()
    sa.setDeviceOperator(S).set(minIterations=12, maxIterations=12, maxEigenvalues=2)
    sa.compute(); sa.close()
    for cplx in (False, True):  # split tiles, real and complex
        vv = vls if not cplx else (vls + 1j * vls[::-1]).astype(np.complex128)
        P = capi.Csr.upload(ctx, Ns, rps.astype(np.int32), cls, vv, column_blocks=-3)
        assert P.layout() == "split_tiles"
        pa = solver.ArnoldiEigenSolver(np.complex128) if cplx else solver.ArnoldiEigenSolver()
        pa.setDeviceOperator(P).set(minIterations=12, maxIterations=12, maxEigenvalues=2)
        pa.compute(); pa.close(); P.close()
    bb = capi.Basis(ctx, S, Ns, 6)
    bb.upload(capi.VEC_W, rng.standard_normal(Ns)); bb.lanczos_enqueue(5)
    import ctypes as C
    h2 = C.c_void_p()
    assert capi.lib().eigenex_basis_clone(bb.h, C.byref(h2)) == 0  # deep copy of a Krylov state
    assert capi.lib().eigenex_basis_destroy(h2) == 0
    bb.close(); S.close()
    zc = solver.LanczosEigenSolver(np.complex128)
    zc.setMatrixMultiplication(lambda x: 2.0 * x, 64).set(maxIterations=5)
    zc.compute()
    for o in (es, ar, tr, eb, zc):
        o.close()
    A.close(); B.close(); ctx.close()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if r == 2:
        free0 = free  # after the first rounds: allocator pools and lazy runtime state are warm
    if r % 10 == 9 or r == rounds - 1:
        print(f"round {r+1}: free {free/2**20:.0f} MiB" + (f"  (delta vs round 3: {(free-free0)/2**20:+.1f} MiB)" if free0 else ""), flush=True)
assert free0 is not None and free0 - free < 64 * 2**20, "device memory is leaking"
print("soak ok")
