"""The CPU checker's index primitives (PCG32 streams, Morton codes, sRGB transfer, ray / box, march helpers) against the outputs of the
REFERENCE's own host-compilable fragments (tests/golden/int_fixtures.json, written by tests/golden/make_int_fixtures.py in the build
container): bit for bit, floats included. The same items run through the HIP library in tests/test_gpu_parity.py."""
import json
import os

from tests import int_fixture_cases, oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_primitives_match_the_reference_fragments():
    c = oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10)
    try:
        n = int_fixture_cases.check(c, exact_pow=True)
    finally:
        c.close()
    assert n == {"pcg32": 44, "morton": 64, "srgb": 256, "ray_box": 96, "march": 128}


def test_fixture_agrees_with_the_older_known_answers():
    fx = int_fixture_cases.load()
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "pcg32_kat.json")))
    assert fx["pcg32_next_uint_seeds_1337_42_0_deadbeefcafe_x6"][:3] == kat["reference_seed_1337"]["draws"]
    assert "make_int_fixtures.py" in fx["_source"]
