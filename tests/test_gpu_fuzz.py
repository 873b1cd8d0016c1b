"""A bounded slice of the soak test (tools/oracle_fuzz.py) inside `pytest -m gpu`: about half a minute of random scenes (Sponza-class meshes of
300 ... 20 000 triangles seen from random cameras, the mesh + analytic scene, the Cornell box, the sphere), frame sizes, depths, roulette
settings, both light sampling strategies, sampler settings and pass counts -- with the library's default walk (intersection counters off)
in half of the cases and the reference's counting walk in the others.  Device and CPU oracle must agree bit for bit on both sum buffers
and on every compared ray counter."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_scenes_against_the_oracle(built):
    import oracle_fuzz
    messages = []
    cases, bad, default_walk = oracle_fuzz.run(budget=30.0, seed=20260928, min_cases=24, log=lambda *a: messages.append(a))
    assert bad == 0, messages
    assert cases >= 24 and default_walk >= 6, (cases, default_walk)
