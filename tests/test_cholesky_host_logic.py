"""Host logic of the blocked inverse-Cholesky factor (two-level blocking, triangular-aware doubling) on the CPU, on the Python + torch.mm
form of the algorithm (tests/ab_partners.py: the A/B partner of inc_gptq_inverse_factor, which runs the same blocking from C++).

The only device kernel of that form is `inc_chol_diag_block` (a 128 x 128 diagonal block: factor + inverse of the
factor).  Here it is replaced by a torch.linalg stand-in, so that everything AROUND it -- outer blocks, panel solves, lower-only
trailing updates, recursive doubling with triangular products, padding -- is checked against the definition the reference
uses:  U = cholesky(cholesky_inverse(cholesky(H)), upper=True)  (reference gptq.py:1228-1230), in fp64.
"""

import pytest
import torch

from tests import ab_partners as P


def _stand_in(A_view, Linv_view, info, tag):
    full = torch.tril(A_view) + torch.tril(A_view, -1).t()  # only the lower triangle is a valid input
    L = torch.linalg.cholesky(full.double())
    A_view.copy_(L.float())
    Linv_view.copy_(torch.linalg.inv(L).float())


def _factor(H, outer, depth, tri_min):
    return P.inverse_cholesky_upper_python(H, check=False, outer=outer, tri_depth=depth, tri_min=tri_min, diag_block=_stand_in)


def _spd(K, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(3 * K, K, generator=g, dtype=torch.float64)
    H = X.t() @ X / (3 * K)
    H += 0.01 * H.diagonal().mean() * torch.eye(K, dtype=torch.float64)
    return H.float()


@pytest.mark.parametrize("K,outer,depth,tri_min", [
    (128, 1024, 2, 512),    # a single diagonal block
    (300, 256, 2, 128),     # padded to 384: identity tail, three inner blocks in two outer blocks
    (1024, 256, 2, 128),    # four outer blocks, doubling inside and across them
    (1500, 512, 1, 128),    # ragged last outer block, one level of triangular splitting
    (2200, 1024, 3, 128),   # deeper splitting than the matrix allows everywhere
])
def test_two_level_factorisation_matches_the_reference_definition(K, outer, depth, tri_min):
    H = _spd(K, K)
    U, info = _factor(H.clone(), outer, depth, tri_min)
    assert int(info.item()) == 0
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H.double())), upper=True)
    assert U.shape == (K, K) and torch.equal(torch.triu(U), U), "U is upper triangular"
    assert (U.diagonal() > 0).all()
    rel = float((U.double() - ref).norm() / ref.norm())
    assert rel < 5e-6, rel  # fp32 factorisation of a matrix with condition number ~1e3
    # and the property the column loop relies on: H^-1 = U^T U
    resid = U.double() @ H.double() @ U.double().t() - torch.eye(K, dtype=torch.float64)
    assert float(resid.norm() / K ** 0.5) < 1e-4


def test_block_sizes_do_not_change_the_result_beyond_rounding():
    H = _spd(900, 7)
    outs = []
    for outer, depth in ((128, 0), (512, 2), (1024, 2)):
        outs.append(_factor(H.clone(), outer, depth, 128)[0])
    for U in outs[1:]:
        assert float((U - outs[0]).norm() / outs[0].norm()) < 2e-6
