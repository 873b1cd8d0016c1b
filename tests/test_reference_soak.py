"""A bounded run of tools/reference_fuzz.py: the oracle (x86 approximation mode) against the reference's own compiled renderer (oracle/_ref/ref_render) on the soak's
random case stream, every pixel and ray counter bit for bit.  Build container only (oracle/_ref is built from /root/reference by oracle/ref_harness); elsewhere skipped.
Round 6: this comparison is what found the host mirror's differently-rounded Matrix4::Inverse (tests/test_host_kat.py pins it now)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_oracle_equals_the_reference_renderer_on_random_cases(built):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_render")):
        pytest.skip("oracle/_ref/ref_render is built from /root/reference (build container only)")
    import oracle_lib
    ok, _ = oracle_lib.set_x86_approximations(True)
    oracle_lib.set_x86_approximations(False)
    if not ok:
        pytest.skip("this CPU cannot evaluate the reference's approximate instructions")
    import reference_fuzz
    lines = []
    cases, bad, _ = reference_fuzz.run(budget=20.0, seed=3, kinds=8, min_cases=12, log=lambda *a: lines.append(" ".join(str(x) for x in a)))
    assert cases >= 12
    assert bad == 0, "\n".join(lines)
