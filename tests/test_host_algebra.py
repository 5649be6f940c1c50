"""The information-form step of the host filter (esekfom.hpp:1782-1809) without either 23x23 inverse.

The reference inverts P / R, adds H^T H to the leading 12 x 12 block and inverts again, then reads the first 12 columns.  The
product's mirror (include/fastlio_amd/esekfom.hpp, info_cols) forms the same columns as B[:, :12] (I + HTH B[:12, :12])^-1 with
B = P / R -- one 12 x 12 elimination, the covariance is never inverted.  The oracle follows the reference's sequence.  Checked
here, without a GPU (the measurement model is the oracle's, injected through flh_esekf_set_meas_model):
  * the two agree to what the reference's sequence itself resolves (1e-11 m without, 1e-7 m with extrinsic estimation: the
    12-column system is poorly conditioned and the double inversion amplifies the last bit of the normal equations to ~1e-8);
  * the product's form does not amplify: normal equations perturbed by 2e-16 relative -- what a different summation order of
    the GPU's partial sums does -- move its posterior by less than 1e-12 m / 1e-10 max|P| (the reference-sequence build
    of the same filter, -DFASTLIO_AMD_REFERENCE_ALGEBRA, moves by 1e-9 .. 1e-8 m / 2e-7 .. 6e-7 max|P| on these problems)."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth
from oracle import pyoracle as po


def _update(pr, m, xp, P, ext, noise, rng):
    sc = po.Scan(pr.body, nthreads=2)

    def fn(x, converge):
        if not sc.h_share_model(m, x, converge, ext):
            return {"valid": False, "n_eff": 0}
        HTH, HTh = sc.normal_equations()
        if noise:
            E = 1 + noise * rng.standard_normal(HTH.shape)
            HTH = HTH * ((E + E.T) / 2)
            HTh = HTh * (1 + noise * rng.standard_normal(HTh.shape))
        return {"valid": True, "n_eff": sc.n_eff, "HTH": HTH, "HTh": HTh, "total_residual": sc.total_residual}

    kf = capi.Esekf(None, max_iter=3, extrinsic_est_en=ext)
    kf.set_meas_model(fn)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    out = kf.get_x(), kf.get_P(), st.passes
    kf.close()
    return out


@pytest.mark.parametrize("cfg,n_scan", [(102, 4000), (103, 3000)])
@pytest.mark.parametrize("ext", [False, True])
def test_gain_form_columns_agree_with_the_reference_sequence_and_do_not_amplify(cfg, n_scan, ext):
    pr = synth.make_problem(60000, n_scan, "avia", cfg=cfg)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
    rng = np.random.default_rng(7)
    x0, P0, passes0 = _update(pr, m, xp, P, ext, 0.0, rng)
    x_ref, P_ref, st_ref = po.Scan(pr.body, nthreads=2).update_iterated(m, xp, P, extrinsic_est_en=ext)
    assert passes0 == st_ref.passes
    np.testing.assert_allclose(x0, x_ref, rtol=0, atol=1e-7 if ext else 1e-11)
    np.testing.assert_allclose(P0, P_ref, rtol=0, atol=1e-6 * np.abs(P_ref).max())
    for _ in range(6):
        x1, P1, passes1 = _update(pr, m, xp, P, ext, 2e-16, rng)
        assert passes1 == passes0
        assert np.abs(x1 - x0).max() < 1e-12
        assert np.abs(P1 - P0).max() < 1e-10 * np.abs(P0).max()
