"""parked with the experiment (EXPERIMENTS.md 5.2): was part of tests/test_oracle_rbfull.py"""

def test_cooperative_form_of_the_recursion_is_the_sequential_one_bit_for_bit():
    """csrc/shared/llpf_rbfull_coop.h (round 5): the last batches of a k_rbfull launch are shared by four waves — one for the nonlinear
    state, three Kalman waves owning the columns c = k mod 3 of every matrix — instead of running as a fourth batch on a few SIMDs.
    Emulated on the host (four threads, a barrier, a plain exchange buffer) on random particles of every instantiated shape, with and
    without the measurement update: every word of xl, the packed covariance and the log-likelihood increment equals what
    llpf_rbf_predict + llpf_rbf_correct (the form the rest of the kernel and the device-order oracle run) produce."""
    cases = [M.linear_case(1, 2, 1, seed=0)[0], M.linear_case(2, 2, 2, seed=1)[0], M.linear_case(4, 8, 2, seed=2)[0], M.quadtank_case()]
    for m in cases:
        o = ob.OracleFilter(_cfg(m, 16), ob.ORDER_DEVICE)
        for has_corr in (1, 0):
            bad, xl, R, ll = o.rbf_coop_check(has_corr, 300, seed=7 + has_corr)
            assert bad == 0, (m.nx, m.rb.nxl, m.ny, has_corr, bad)
            assert np.all(np.isfinite(xl)) and np.all(np.isfinite(R)) and np.isfinite(ll) and (ll != 0.0) == bool(has_corr)
