"""CPU oracle for the cuRobo hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package (see the header of ``oracle/curobo_oracle.c``).  The product package
``curobo_amd`` never imports it.
"""

from .oracle import Oracle, build_native_oracle, build_oracle, load_native_oracle, load_oracle  # noqa: F401
