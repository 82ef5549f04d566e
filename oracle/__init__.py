"""ORACLE — TEST INFRASTRUCTURE ONLY.

A CPU (torch fp32, eager) restatement of the reference's distill-step arithmetic
(ForJadeForest/DistillCLIP: model/component/*, model/_loss.py, model/loss_component/*).
Every function cites the reference file:line it follows.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline.  Nothing under
distillclip_amd/ imports it; the product path fails loudly when the HIP library is missing.

Parity status: PINNED.  tests/golden/*.npz were produced by running the reference's own
modules in the build container (tools/golden/gen_golden.py) on deterministic synthetic
weights/inputs (distillclip_amd/synth.py); tests/test_oracle_golden.py checks this
restatement against them.  The timm boundary (Mlp / PatchEmbed / DropPath / trunc_normal_,
version unpinned by the reference) is pinned against tools/golden/ref_shims, our
restatement of timm's published semantics (see that README).
Exception: oracle/metrics.py (validation retrieval metrics) is pinned by known-answer cases only —
the reference computes top-k accuracy with torchmetrics, which this image lacks, so no reference-run golden exists.
"""
from .encoders import (teacher_image_forward, teacher_text_forward, student_image_forward,   # noqa: F401
                       student_text_forward, clip_forward, bf16_matched, clip_student_image_forward,
                       clip_student_text_forward)
from .loss import LossOracle, LOSS_FUNCS   # noqa: F401
