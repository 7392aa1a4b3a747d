"""CPU oracle for the SandCrate per-timestep particle update.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the algorithm
the reference runs in ``src/crate/crate.py:91-129`` (``Crate.physics_tick``) and
its callees.  It exists to check the HIP path, never to replace it:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
  leg may import it;
* nothing under ``sand_crate_amd/`` imports it, and the product path raises when
  the HIP library is missing instead of falling back to this code.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the unmodified
reference (in the build container only) and writes input/output vectors to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against them, and ``tests/test_reference_kats.py`` re-expresses the
reference's own known-answer tests (``tests/test_distance.py:16-70``).

Modules
-------
world       rigid bodies, particle sources, YAML config (host-side world state)
neighbors   strip sort + neighbor lists (collision_detector.py:9-128)
tick        vectorised single-tick update over padded P x 20 neighbor arrays
tick_loops  the same tick written with the reference's per-particle loop
            structure; it is what ``bench.py`` times as "the reference NumPy path"
scene       OracleCrate: YAML-driven scene runner built on ``tick``
"""
