"""Host logic of matmul(...ms) (matmul.js:150-236) without a GPU: the chain planner is pinned to the reference by
evaluating its parenthesisation with the oracle's bit-exact matmul2 — any other order changes the rounding."""
import numpy as np
import pytest

import oracle
from conftest import golden_cases
from nd4js_amd import la, rng


def operands(g):
    if g.meta.get("hand"):
        return [g["M%d" % k] for k in range(g.n)]
    return [rng.matrix(g.seed0 + k, *s) for k, s in enumerate(g.shapes)]


@pytest.mark.parametrize("name", golden_cases(op="matmul"))
def test_chain_order_bit_identical_to_reference(golden, name):
    g = golden(name)
    c = la.matmul(*operands(g), _matmul2=oracle.matmul2)
    ref = g["C"]
    assert c.shape == ref.shape and np.array_equal(c, ref)


def test_plan_textbook_example_and_ties():
    # CLRS 15.2: dims 30,35,15,5,10,20,25 -> ((A1(A2A3))((A4A5)A6))
    dims = [30, 35, 15, 5, 10, 20, 25]
    cut = la.chain_plan([(dims[i], dims[i + 1]) for i in range(6)])
    assert cut[0][5] == 2 and cut[0][2] == 0 and cut[3][5] == 4
    cut = la.chain_plan([(16, 16)] * 4)            # all orders equal: the first split wins at every level
    assert cut[0][3] == 0 and cut[1][3] == 1


def test_plan_errors():
    with pytest.raises(ValueError, match="Shape mismatch."):
        la.chain_plan([(2, 3), (4, 2), (2, 2)])
    with pytest.raises(ValueError, match="broadcast-compatible"):
        la.chain_plan([(2, 2, 3), (3, 3, 2), (2, 2)])
    with pytest.raises(ValueError, match="Integer overflow"):
        la.chain_plan([(2 ** 31 - 1,) * 40] * 3)           # numel*K = inf in double arithmetic: no split is cheaper than inf
