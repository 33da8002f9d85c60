"""CPU oracle for the spectra->SMILES hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``multimodal_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker.

Parity status: PINNED.  ``oracle/restate.py`` is checked against golden vectors
in ``tests/golden/*.npz`` that ``oracle/gen_golden.py`` produced by importing
the reference's own ``models`` package in the build container (the reference
holds no tests or golden vectors of its own: SURVEY.md section 4 / 8c).
"""
