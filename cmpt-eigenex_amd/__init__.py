"""cmpt-eigenex_amd: MI355X-native Krylov (Lanczos/Arnoldi) inner loop behind the
solver API of versmc/cmpt-eigenex.

Layout
  csrc/            hand-written gfx950 HIP kernels + the C ABI (include/eigenex_hip.h)
  include/cmpt/eigen_ex/   header-only C++ host: LanczosEigenSolver / ArnoldiEigenSolver
                   with the reference's interface, calling the C ABI
  capi.py          ctypes view of the C ABI (plumbing for tests/ and bench.py)
  solver.py        ctypes view of the C++ solver classes (libeigenex_solver.so)
  synthetic.py     synthetic operators of BASELINE.md (host-side CSR builders)
  build.py         hipcc build recipe (in-tree .so files)
"""
__version__ = "0.1.0"
