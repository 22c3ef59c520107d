"""The duplicate-group cluster rule of fullrank_verify_kernel (csrc/kernels_fullverify.inc, DUP instantiations), as a host
model: the kernel's bit-string arithmetic restated word for word in Python integers and checked against the rule's
definition on random inputs.  No device, no oracle: this pins the TRICK (one multi-word addition fills every run of close
pairs upward from its seeds; runs that cross lanes hand a carry to the lane before them; a depth that cuts the list keeps a
zone mask), the GPU parity tests pin the kernel.

Definition.  A candidate's sorted documents 0 .. n-1 give pairs j = (j, j+1).  Every pair carries three bits: F (the gap is
proven), D (the classes differ), N (the two are NOT members of one duplicate group).  A cluster is a maximal run of pairs
without F.  The candidate fails iff some cluster that reaches into the first `cut` pairs holds a D pair and an N pair
(no cut: every cluster counts)."""
import random

import pytest

M32 = 0xFFFFFFFF


def brute(F, D, N, cut):
    n, j, fail = len(F), 0, False
    while j < n:
        if F[j]:
            j += 1
            continue
        k = j
        while k < n and not F[k]:
            k += 1
        if j < cut and any(D[j:k]) and any(N[j:k]):  # the run j .. k-1 starts inside the cut (runs are contiguous)
            fail = True
        j = k
    return fail


def lane_strings(F, D, N, NL):
    """One lane's NL pairs as the kernel gathers them: W words, pair 0 at the top bit of word 0 (alignbit shifts left); a part
    full last word moved to the top with the padding below repeating the last F bit."""
    W, Z = (NL + 31) // 32, ((NL + 31) // 32) * 32 - NL
    Fw, Dw, Nw = [0] * W, [0] * W, [0] * W
    for j in range(NL):
        b = j // 32
        Fw[b] = ((Fw[b] << 1) | F[j]) & M32
        Dw[b] = ((Dw[b] << 1) | D[j]) & M32
        Nw[b] = ((Nw[b] << 1) | N[j]) & M32
    if Z:
        pad = (-(Fw[W - 1] & 1)) & ((1 << Z) - 1)
        Fw[W - 1] = ((Fw[W - 1] << Z) | pad) & M32
        Dw[W - 1] = (Dw[W - 1] << Z) & M32
        Nw[W - 1] = (Nw[W - 1] << Z) & M32
    Cw = [~f & M32 for f in Fw]
    return Cw, [d & c for d, c in zip(Dw, Cw)], [x & c for x, c in zip(Nw, Cw)]


def kernel_model(F, D, N, NL, PL, cut):
    """F, D, N: PL * NL pair bits of one candidate (the last pair of the last lane compares against -inf: always F)."""
    W = (NL + 31) // 32
    lanes = [lane_strings(F[h * NL:(h + 1) * NL], D[h * NL:(h + 1) * NL], N[h * NL:(h + 1) * NL], NL) for h in range(PL)]
    # stage 1: (generate D, generate N, propagate) per lane, then the suffix scan over the lanes below
    t = []
    for Cw, Dw, Nw in lanes:
        cd = cn = 0
        allones = M32
        for b in range(W - 1, -1, -1):
            sd, sn = Cw[b] + Dw[b] + cd, Cw[b] + Nw[b] + cn
            cd, cn = sd >> 32, sn >> 32
            allones &= Cw[b]
        t.append(cd | (cn << 1) | ((1 if allones == M32 else 0) << 2))
    v = [t[(h + 1) % PL] for h in range(PL)]  # (the kernel's lanes wrap inside the wave: a candidate's last lane never propagates)
    s2 = 1
    while s2 < PL:
        o = [v[(h + s2) % PL] for h in range(PL)]
        v = [(v[h] & 3) | (o[h] & (M32 if (v[h] >> 2) & 1 else 0)) for h in range(PL)]
        s2 <<= 1
    fail = False
    for h, (Cw, Dw, Nw) in enumerate(lanes):
        Dw, Nw = list(Dw), list(Nw)
        if PL > 1:
            Dw[W - 1] |= v[h] & 1 & Cw[W - 1]
            Nw[W - 1] |= (v[h] >> 1) & 1 & Cw[W - 1]
        lim = (cut if cut is not None else PL * NL) - h * NL
        cd = cn = bad = 0
        for b in range(W - 1, -1, -1):
            sd, sn = Cw[b] + Dw[b] + cd, Cw[b] + Nw[b] + cn
            cd, cn = sd >> 32, sn >> 32
            up_d = (((sd & M32) ^ Cw[b]) & Cw[b]) | Dw[b]
            up_n = (((sn & M32) ^ Cw[b]) & Cw[b]) | Nw[b]
            bw = (up_d & Nw[b]) | (up_n & Dw[b])
            if cut is not None and cut < PL * NL:
                c = max(0, min(32, lim - 32 * b))
                bw = (bw | (up_d & up_n)) & ((0xFFFFFFFF00000000 >> c) & M32)
            bad |= bw
        fail |= bad != 0
    return fail


@pytest.mark.parametrize("NL,PL", [(16, 1), (32, 1), (48, 1), (64, 1), (80, 1), (96, 1), (64, 2), (80, 2), (96, 4), (64, 8), (80, 16), (16, 4)])
def test_bit_string_rule_equals_its_definition(NL, PL):
    rng = random.Random(NL * 100 + PL)
    n = NL * PL
    for it in range(400):
        pf = rng.choice([0.02, 0.1, 0.3, 0.6])  # few far pairs: long clusters that cross words and lanes
        pd, pn = rng.choice([0.0, 0.02, 0.2]), rng.choice([0.0, 0.02, 0.2, 0.9])
        F = [1 if rng.random() < pf else 0 for _ in range(n)]
        F[n - 1] = 1  # against the -inf behind the candidate's last lane
        D = [1 if rng.random() < pd else 0 for _ in range(n)]
        N = [1 if rng.random() < pn else 0 for _ in range(n)]
        for cut in (None, rng.randrange(1, n), rng.randrange(1, min(n, 40))):
            got = kernel_model(F, D, N, NL, PL, cut)
            assert got == brute(F, D, N, n if cut is None else cut), (it, cut, F, D, N)


def test_a_cluster_with_both_kinds_of_pair_beyond_the_cut_counts_only_if_it_reaches_it():
    NL, PL = 32, 1
    F = [1] * 32
    D, N = [0] * 32, [0] * 32
    for j in range(8, 14):
        F[j] = 0
    D[12], N[13] = 1, 1  # the differing pair and the non-duplicate pair: pairs 12 and 13
    assert kernel_model(F, D, N, NL, PL, 9) is True    # the cluster 8..13 starts inside the first 9 pairs
    assert kernel_model(F, D, N, NL, PL, 8) is False   # it starts beyond them: any order of its documents leaves the first 8 ranks alone
    assert kernel_model(F, D, N, NL, PL, None) is True
    assert brute(F, D, N, 9) and not brute(F, D, N, 8)
