"""CPU oracle for the Crane hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a CPU restatement of the reference's
algorithm (lucasjinreal/Crane, ``crane-core/src/models/qwen3*`` and
``crane-core/src/ops/gdn``).  It exists to *check* the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import or execute anything from here.  The product
package ``crane_amd`` must never import it; the product path fails loudly if
the HIP library is missing instead of falling back to this code.

``oracle/_ref/`` (git-ignored) holds what of the reference itself builds here: its
``gdn.cu`` / ``topk.cu`` kernels as gfx950 code objects (``build_ref.sh``,
``ref_kernels.py``), used by ``tests/test_gpu_ref_kernels.py`` to pin the restatement.
"""
