"""CPU oracle for the Crane hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a CPU restatement of the reference's
algorithm (lucasjinreal/Crane, ``crane-core/src/models/qwen3*`` and
``crane-core/src/ops/gdn``).  It exists to *check* the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import or execute anything from here.  The product
package ``crane_amd`` must never import it; the product path fails loudly if
the HIP library is missing instead of falling back to this code.
"""
