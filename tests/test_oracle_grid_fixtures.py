"""tests/golden/f11_grids/ -- the committed outputs of the CPU oracle's LARGE grid sweeps, which the GPU suite compares the kernels
with instead of re-running tens of seconds of oracle per case (oracle/pyoracle.py Oracle.acq_grid, oracle/gen_golden_grids.py).
Here, without a GPU: every fixture's name is the hash of the inputs it holds, one one-block fixture is recomputed in full by the
live oracle, cells of it by the reference's own C (when oracle/_ref is built), and the lookup itself hits, misses and honours
live=True as documented.  `python oracle/gen_golden_grids.py --verify` recomputes ALL of them (minutes)."""
import glob
import os

import numpy as np

from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "f11_grids")
FILES = sorted(glob.glob(os.path.join(DIR, "*.npz")))


def test_every_fixture_is_named_by_the_hash_of_its_inputs_and_has_the_grid_shape():
    assert len(FILES) >= 30
    kinds = set()
    for f in FILES:
        with np.load(f) as z:
            args = tuple(int(a) for a in z["args"])
            assert os.path.basename(f)[:-4] == pyoracle.grid_key(z["blocks"], z["prns"], args)
            n_ms, _, _, n_dopp, n_bits = args
            assert z["blocks"].shape == (n_ms * 2046,) and z["peaks"].shape == (len(z["prns"]), n_dopp, n_bits)
            assert z["peaks"].dtype == pyoracle.PEAK_DTYPE and n_ms * z["peaks"].size >= pyoracle.FIXTURE_MIN_UNITS
            assert int(z["peaks"]["max_val"].max()) > 0
            kinds.add(n_ms)
    assert {1, 10} <= kinds          # one-block grids and BASELINE configs[3]'s ten-block searches


def test_a_one_block_fixture_equals_the_live_oracle_and_the_reference(oracle):
    f = next(f for f in FILES if int(np.load(f)["args"][0]) == 1)
    with np.load(f) as z:
        args = tuple(int(a) for a in z["args"])
        live = oracle.acq_grid(z["blocks"], args[0], z["prns"], *args[1:], n_threads=max(4, len(os.sched_getaffinity(0))), live=True)
        assert np.array_equal(live, z["peaks"])
        if pyoracle.RefPM.available():
            ref = pyoracle.RefPM()
            rng = np.random.default_rng(3)
            for _ in range(8):
                p, d, b = int(rng.integers(len(z["prns"]))), int(rng.integers(args[3])), int(rng.integers(args[4]))
                di, dq = ref.wipeoff(np.ascontiguousarray(z["blocks"]), float(4092000 + args[1] + d * args[2]))
                mx, avr, ph = ref.correlation_search(ref.replica(ref.ca_code(int(z["prns"][p])), b), di, dq, 0, 2046)
                got = z["peaks"][p, d, b]
                assert (int(got["max_val"]), int(got["avr"]), int(got["phase"])) == (mx, avr, ph)


def test_lookup_hits_only_on_identical_inputs_and_never_when_live(tmp_path):
    orc = pyoracle.Oracle()
    orc.grid_fixtures = DIR
    f = next(f for f in FILES if int(np.load(f)["args"][0]) == 1)
    with np.load(f) as z:
        blocks, prns, args, want = z["blocks"].copy(), z["prns"].copy(), tuple(int(a) for a in z["args"]), z["peaks"].copy()
    got = orc.acq_grid(blocks, args[0], prns, *args[1:])
    assert orc.fixture_hits == 1 and np.array_equal(got, want)
    # a smaller grid of the same block is below the fixture threshold: computed live, no lookup
    small = orc.acq_grid(blocks, 1, prns[:2], args[1], args[2], 2, 8)
    assert orc.fixture_hits == 1 and np.array_equal(small, want[:2, :2])
    # one flipped input bit: another hash, a miss -> live (only the first PRNs, to keep this test short, would change the key
    # as well: the whole call is made once, on 8 threads)
    blocks[100] ^= 1
    assert pyoracle.load_grid_fixture(DIR, pyoracle.grid_key(blocks, prns, args), blocks, prns, args) is None
    # a file under the right name holding OTHER inputs is not trusted
    blocks[100] ^= 1
    pyoracle.save_grid_fixture(str(tmp_path), pyoracle.grid_key(blocks, prns, args), blocks[::-1].copy(), prns, args, want)
    assert pyoracle.load_grid_fixture(str(tmp_path), pyoracle.grid_key(blocks, prns, args), blocks, prns, args) is None
