"""CPU oracle for the Krylov hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (see oracle/krylov_oracle.py for the pinning statement).
"""
