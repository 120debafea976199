def lagrange_weights(o, r):
    """Montgomery-form Lagrange basis on {0,1,2} at r, via the oracle's field ops"""
    L = o.lib
    one = o.f.r_mod_p
    two = L.sco_add(o.fp, one, one)
    inv2 = L.sco_inv(o.fp, two)
    rm1, rm2 = L.sco_sub(o.fp, r, one), L.sco_sub(o.fp, r, two)
    l0 = L.sco_mul(o.fp, L.sco_mul(o.fp, rm1, rm2), inv2)
    l1 = L.sco_sub(o.fp, 0, L.sco_mul(o.fp, r, rm2))
    l2 = L.sco_mul(o.fp, L.sco_mul(o.fp, r, rm1), inv2)
    return l0, l1, l2
