"""Deterministic synthetic tables shaped like the flattened ACE structures
(SURVEY.md 8d): no ACE files exist in the image.  Shared by the golden
generator and the tests (TEST INFRASTRUCTURE)."""
import numpy as np


def mu_grid(M):
    dmu = 2.0 / float(M - 1)
    mu = -1.0 + np.arange(M, dtype=np.float64) * dmu
    mu[-1] = 1.0
    return mu


def kalbach_rows(M, n_rows, np_lo, np_hi, e_lo, e_hi, seed, dup_last=False, intt=2):
    """A law-44-like ScattData: n_rows incoming energies, each with NP outgoing
    energies (Eout starts at 0), a normalised pdf and Kalbach-Mann f(mu) columns
    f = A/(2 sinh A) (cosh(A mu) + R sinh(A mu)), R in [0,0.5], A in [0.5,3]
    (scattdata_header.F90:822-831).  Returns the CSR tables of the C ABI."""
    rng = np.random.default_rng(seed)
    mu = mu_grid(M)
    e_grid = np.logspace(np.log10(e_lo), np.log10(e_hi), n_rows)
    row_ptr = [0]
    eout, pdf, f, intts = [], [], [], []
    for k in range(n_rows):
        NP = int(rng.integers(np_lo, np_hi + 1))
        emax = 0.9 * e_grid[k]
        eo = np.concatenate([[0.0], np.sort(rng.uniform(0.0, emax, NP - 2)), [emax]])
        if dup_last and k % 2 == 1:
            eo[-2] = eo[-1]          # Zr-90-like duplicate end points (:1127-1130)
        p = np.exp(-eo / (0.3 * emax + 1e-30)) * (eo + 0.05 * emax)
        de = np.diff(eo)
        norm = np.sum(0.5 * (p[1:] + p[:-1]) * de)
        p = p / norm
        R = rng.uniform(0.0, 0.5, NP)
        A = rng.uniform(0.5, 3.0, NP)
        cols = 0.5 * A[:, None] / np.sinh(A[:, None]) * (
            np.cosh(A[:, None] * mu[None, :]) + R[:, None] * np.sinh(A[:, None] * mu[None, :]))
        eout.append(eo)
        pdf.append(p)
        f.append(cols)
        intts.append(intt)
        row_ptr.append(row_ptr[-1] + NP)
    return dict(e_grid=e_grid, row_ptr=np.array(row_ptr, dtype=np.int32),
                eout=np.concatenate(eout), pdf=np.concatenate(pdf),
                intt=np.array(intts, dtype=np.int32),
                f=np.ascontiguousarray(np.concatenate(f, axis=0)))


def law9_edata(e_lo, e_hi, n=8, U=0.5):
    """edist%data of an evaporation spectrum (ACE law 9): TAB1 of T(E) with NR=0
    followed by the restriction energy U (scattdata_header.F90:1289-1302)."""
    E = np.logspace(np.log10(e_lo), np.log10(e_hi), n)
    T = 0.2 + 0.1 * np.sqrt(E)
    return np.concatenate([[0.0, float(n)], E, T, [U]])


def sab_table(mode, seed, NEi=24, NEo=16, NMU=8, elastic=None, NEo_range=(10, 30)):
    """A synthetic thermal-scattering table shaped like the ACE data NDPP consumes
    (ace_header.F90:201-235).  mode 0/1: equal/skewed discrete E_out x mu;
    mode 2: continuous E_out pdf with discrete mu, NEo_range[0] .. NEo_range[1] outgoing energies per
    incoming energy ((50, 300)).  elastic: None, "coherent"
    (Bragg edges, exact mode) or "incoherent" (discrete cosines)."""
    rng = np.random.default_rng(seed)
    kT = 2.53e-8
    ei = np.logspace(-11, np.log10(4e-6), NEi)
    sig = 20.0 + 60.0 / (1.0 + ei / 1e-8)
    t = dict(threshold_inelastic=float(ei[-1]), threshold_elastic=0.0, NEi=NEi, NEo=NEo, NMU=NMU,
             mode=mode, ei=ei, sig=sig,
             e_out=np.zeros(1), mu=np.zeros(1), cptr=np.zeros(NEi + 1, dtype=np.int32),
             ce_out=np.zeros(1), cpdf=np.zeros(1), cmu=np.zeros(1),
             el_mode=3, NEe=0, NMUe=0, ee=np.zeros(1), eP=np.zeros(1), emu=np.zeros(1))
    if mode in (0, 1):
        q = (np.arange(NEo) + 0.5) / NEo
        e_out = np.empty((NEi, NEo))
        mu = np.empty((NEi, NEo, NMU))
        for k in range(NEi):
            e_out[k] = np.sort(ei[k] * (0.2 + 1.6 * q) + kT * (-np.log(1 - q)) * 1.5)
            base = np.clip(1.0 - 2.0 * kT / (ei[k] + kT), -0.9, 0.9)
            mu[k] = np.sort(np.clip(base * 0.3 + rng.uniform(-0.95, 0.95, (NEo, NMU)), -1, 1), axis=1)
        t["e_out"], t["mu"] = e_out.ravel(), mu.ravel()
    else:
        ptr, ce, cp, cm = [0], [], [], []
        for k in range(NEi):
            n = int(rng.integers(NEo_range[0], NEo_range[1] + 1))
            emax = 3.0 * ei[k] + 12 * kT
            eo = np.concatenate([[0.0], np.sort(rng.uniform(0, emax, n - 2)), [emax]])
            pdf = (eo + 0.02 * emax) * np.exp(-eo / (ei[k] + 2 * kT))
            pdf /= np.sum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(eo))
            cm.append(np.sort(rng.uniform(-1, 1, (n, NMU)), axis=1))
            ce.append(eo)
            cp.append(pdf)
            ptr.append(ptr[-1] + n)
        t.update(cptr=np.array(ptr, dtype=np.int32), ce_out=np.concatenate(ce),
                 cpdf=np.concatenate(cp), cmu=np.concatenate(cm).ravel())
    if elastic == "coherent":
        ee = np.array([1.8e-9, 4.5e-9, 6.3e-9, 1.1e-8, 2.2e-8, 5.0e-8, 1.2e-7, 4.0e-6])
        t.update(el_mode=4, NEe=len(ee), NMUe=0, ee=ee, eP=np.cumsum(1e-9 * rng.uniform(0.5, 2, len(ee))),
                 threshold_elastic=float(ee[-1]))
    elif elastic == "incoherent":
        ee = np.logspace(-11, np.log10(4e-6), 12)
        NMUe = 6
        t.update(el_mode=3, NEe=len(ee), NMUe=NMUe, ee=ee, eP=5.0 / (1 + ee / 1e-7),
                 emu=np.sort(rng.uniform(-1, 1, (len(ee), NMUe)), axis=1).ravel(),
                 threshold_elastic=float(ee[-1]))
    return t


def sab_ein_grid(t, n=40, seed=3):
    """Test E_in points: inside, at and beyond the table, incl. exact table points."""
    rng = np.random.default_rng(seed)
    e = np.concatenate([10 ** rng.uniform(-11.3, np.log10(t["threshold_inelastic"]), n),
                        t["ei"][[0, 3, -1]], [t["threshold_inelastic"] * 1.5, 5e-12]])
    return np.sort(e)


# ---- fission spectra (chi) ---------------------------------------------------
def law4_block(e_in, np_pts, e_max, seed, hist=False):
    """edist%data of a continuous tabular distribution (ACE law 4): NR, [NBT,INT], NE,
    E_in(NE), L(NE), then per E_in: INTT', NP, E_out(NP), pdf(NP), cdf(NP)
    (layout read at chidata_header.F90:258-350)."""
    rng = np.random.default_rng(seed)
    NE = len(e_in)
    head = ([1.0, float(NE), 1.0] if hist else [0.0]) + [float(NE)] + list(e_in)
    blocks, locs = [], []
    pos = len(head) + NE  # 0-based offset of the first block == its 1-based index - 1
    for k in range(NE):
        eo = np.concatenate([[0.0], np.sort(rng.uniform(0, e_max, np_pts - 2)), [e_max]])
        T = 1.2 + 0.05 * k
        pdf = np.sqrt(eo + 1e-3) * np.exp(-eo / T)
        cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(eo))])
        pdf, cdf = pdf / cdf[-1], cdf / cdf[-1]
        blk = [2.0, float(np_pts)] + list(eo) + list(pdf) + list(cdf)
        locs.append(float(pos))
        pos += len(blk)
        blocks += blk
    return np.array(head + locs + blocks)


def tab1_block(x, y):
    return [0.0, float(len(x))] + list(x) + list(y)


def chi_case(seed=18):
    """A synthetic fissionable nuclide (SURVEY 8d config 5 shape): three fission
    reactions -- MT 19 with two nested spectra (law 4 then law 7), MT 20 law 11 (Watt),
    MT 21 law 9 -- and 3 delayed precursor groups (law 4, law 4 histogram, law 7)."""
    rng = np.random.default_rng(seed)
    n_grid = 50
    energy = np.logspace(-11, np.log10(20.0), n_grid)
    part = [2.0 / (1 + energy) ** 0.3, 0.6 * np.ones(n_grid), 0.3 * np.ones(n_grid)]
    thr = [1, 36, 42]
    sig = [part[0][thr[0] - 1:], part[1][thr[1] - 1:] * np.linspace(0, 1, n_grid - thr[1] + 1),
           part[2][thr[2] - 1:] * np.linspace(0, 1, n_grid - thr[2] + 1)]
    fission = part[0].copy()
    fission[thr[1] - 1:] += sig[1]
    fission[thr[2] - 1:] += sig[2]
    e3 = [1e-11, 1.0, 20.0]
    spectra = [  # (law, data) in chain order
        (4, law4_block(e3, 12, 15.0, seed)),
        (7, np.array(tab1_block([1e-11, 20.0], [1.30, 1.45]) + [-20.0])),        # T(E), U
        (11, np.array(tab1_block([1e-11, 20.0], [0.95, 1.05]) +                  # Watt a(E)
                      tab1_block([1e-11, 20.0], [2.2, 2.6]) + [3.0])),           # b(E), U
        (9, np.array(tab1_block([5.0, 20.0], [0.5, 0.9]) + [5.5])),
    ]
    nnest = [2, 1, 1]
    delayed = [(4, law4_block([1e-11, 20.0], 8, 3.0, seed + 1)),
               (4, law4_block([1e-11, 20.0], 8, 2.0, seed + 2, hist=True)),
               (7, np.array(tab1_block([1e-11, 20.0], [0.4, 0.45]) + [-20.0]))]
    prec = []
    for j in range(3):
        prec += [0.01 * (j + 1)] + tab1_block([1e-11, 20.0], [0.2 + 0.1 * j, 0.25 + 0.1 * j])
    return dict(n_grid=n_grid, energy=energy, fission=fission,
                nu_t_type=1, nu_t_data=np.array([2.0, 2.4, 0.12]),
                nu_d_type=2, nu_d_data=np.array(tab1_block([1e-11, 4.0, 20.0], [0.016, 0.016, 0.009])),
                n_prec=3, prec_data=np.array(prec), mts=[19, 20, 21], thr=thr, sig=sig,
                nnest=nnest, spectra=spectra, delayed=delayed,
                bins=np.concatenate([[0.0], np.logspace(-3, np.log10(20.0), 7)]))


# ---- raw ACE blocks for the ACE -> tabular conversion (convert_file4 / convert_file6) ----
def _ang_table(rng, interp, npts, positive=False):
    """[JJ, NP, cosines(NP), pdf(NP), cdf(NP)] of one ACE tabular angular distribution."""
    cs = np.concatenate([[-1.0], np.sort(rng.uniform(-1, 1, npts - 2)), [1.0]])
    if positive:                       # log interpolation in mu needs mu > 0 to be finite
        cs = np.linspace(1e-3, 1.0, npts)
    pdf = 0.5 * (1 + rng.uniform(-0.8, 0.8) * cs + rng.uniform(0, 0.4) * cs ** 2) + 0.05
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(cs))])
    return [float(interp), float(npts)] + list(cs) + list(pdf / cdf[-1]) + list(cdf / cdf[-1])


def ace_adist(energies, kinds, seed):
    """DistAngle (ace_header.F90:14-24): kinds[k] in {"iso", "equi", "hist", "lin"}.
    Returns energy, type, location, data (locations are 0-based offsets lc with
    data(lc+1) the first word, as ace.F90 stores them)."""
    rng = np.random.default_rng(seed)
    data, typ, loc = [0.0], [], []     # one pad word so that no table sits at lc = 0
    for k in kinds:
        if k == "iso":
            typ.append(1)
            loc.append(0)
        elif k == "equi":
            typ.append(2)
            loc.append(len(data))       # data(lc+1) is the first edge; data(lc) must exist
            edges = np.concatenate([[-1.0], np.sort(rng.uniform(-1, 1, 31)), [1.0]])
            data += list(edges)
        else:
            typ.append(3)
            loc.append(len(data))
            data += _ang_table(rng, 1 if k == "hist" else 2, int(rng.integers(3, 40)))
    return (np.asarray(energies, dtype=np.float64), np.array(typ, dtype=np.int32),
            np.array(loc, dtype=np.int32), np.array(data))


def ace_edist(law, e_in, np_lo, np_hi, seed, interps=(1, 2), inttp=2):
    """edist%data of ACE law 4 / 44 / 61 (layout read at scattdata_header.F90:799-865):
    NR=0, NE, E_in(NE), L(NE), then per E_in INTT', NP, E_out, pdf, cdf and law 44: R, A;
    law 61: LC(NP) locators (0 = isotropic) followed by the angular tables."""
    rng = np.random.default_rng(seed)
    NE = len(e_in)
    head = [0.0, float(NE)] + list(e_in)
    body, locs = [], []
    pos = len(head) + NE
    for k in range(NE):
        NP = int(rng.integers(np_lo, np_hi + 1))
        emax = 0.9 * e_in[k]
        eo = np.concatenate([[0.0], np.sort(rng.uniform(0, emax, NP - 2)), [emax]])
        pdf = np.exp(-eo / (0.3 * emax)) * (eo + 0.05 * emax)
        cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(eo))])
        blk = [float(inttp), float(NP)] + list(eo) + list(pdf / cdf[-1]) + list(cdf / cdf[-1])
        locs.append(float(pos))
        if law == 44:
            blk += list(rng.uniform(0, 0.5, NP)) + list(rng.uniform(0.5, 3.0, NP))
        elif law == 61:
            lc_at = len(blk)
            blk += [0.0] * NP
            for j in range(NP):
                if rng.uniform() < 0.2:
                    continue                                  # LC = 0: isotropic
                interp = int(interps[int(rng.integers(len(interps)))])
                blk[lc_at + j] = float(pos + len(blk))
                blk += _ang_table(rng, interp, int(rng.integers(3, 30)), positive=interp in (3, 5))
        pos += len(blk)
        body += blk
    return np.array(head + locs + body)


# ---- nuclide-level inputs of the E_in grid builders (create_Ein_grid) --------------------
def grid_cases():
    """(name, dict) pairs: what create_Ein_grid reads -- nuclide grid, group structure,
    awr, kT, free-gas cutoff, inelastic threshold and per ScattData (is_init, MT, Q, E_grid)."""
    kT = 2.5301e-8
    bins2 = np.array([0.0, 6.25e-7, 20.0])
    bins8 = np.concatenate([[1e-11], np.logspace(-7, np.log10(20.0), 8)])
    bins5 = np.array([0.0, 1e-3, 0.05, 0.5, 3.0, 20.0])
    h1 = dict(awr=0.999167, kT=kT, cutoff=400.0 * kT, thresh=20.0, bins=bins2,
              nuc=np.logspace(-11, np.log10(20.0), 300),
              sds=[(1, 2, 0.0, np.array([1e-11, 1e-6, 1.0, 20.0]))])
    nuc_u = np.unique(np.concatenate([np.logspace(-11, np.log10(30.0), 400), [0.0449, 0.148, 1.0]]))
    u_sds = [(1, 2, 0.0, np.logspace(-5, np.log10(20.0), 40)),
             (1, 51, -0.0449, np.array([0.0451, 1.0, 20.0])),
             (1, 52, -0.148, np.array([0.1486, 2.0, 30.0])),
             (0, 18, 190.0, np.array([1e-11, 20.0])),
             (1, 91, -1.2, np.array([1.3, 2.5, 6.0, 12.0, 20.0])),
             (1, 16, -6.15, np.array([6.2, 9.0, 14.0, 20.0]))]
    u2 = dict(awr=236.0058, kT=kT, cutoff=400.0 * kT, thresh=0.0449, bins=bins2, nuc=nuc_u, sds=u_sds)
    u5 = dict(u2, bins=bins5)
    u8 = dict(u2, bins=bins8, cutoff=0.0)        # free-gas treatment off, first edge > 0
    o16 = dict(awr=15.8575, kT=kT, cutoff=400.0 * kT, thresh=6.4, bins=bins5,
               nuc=np.logspace(-11, np.log10(20.0), 150),
               sds=[(1, 2, 0.0, np.logspace(-6, np.log10(20.0), 25)),
                    (1, 51, -6.05, np.array([6.4, 10.0, 20.0]))])
    return [("h1_g2", h1), ("u238_g2", u2), ("u238_g5", u5), ("u238_g8_nofg", u8), ("o16_g5", o16)]


# ---- a whole nuclide as raw ACE blocks (calc_scatt / ndpp_scatt_nuclide) -------------------
def nuclide_case():
    """An O-16-like nuclide: elastic with isotropic / tabular / 32-equiprobable angular tables
    and free gas below 4 kT; MT 51 level (law 3 + tabular angles); MT 91 continuum (law 44, CM,
    p_valid 1 -> 0.8); MT 22 (law 61, lab, energy-dependent multiplicity); MT 102 capture and an
    MT 18 fission entry that ScattData%init must skip.  Three groups."""
    kT = 2.5301e-8
    n_grid = 30
    energy = 1e-11 * (20.0 / 1e-11) ** (np.arange(n_grid) / (n_grid - 1.0))
    elastic = 3.8 + 0.2 / (1.0 + energy)
    el_ad = ace_adist([1e-11, 1e-3, 20.0], ["iso", "lin", "equi"], seed=16)
    thr = {51: 29, 91: 29, 22: 28}
    def sigma(MT, scale):
        n = n_grid - thr[MT] + 1
        return scale * np.arange(n) / max(n - 1, 1) + 0.01 * np.arange(n)
    lvl_ad = ace_adist([energy[thr[51] - 1], 20.0], ["lin", "hist"], seed=51)
    pv = ([1e-5, 20.0], [1.0, 0.8])
    reactions = [
        dict(MT=2, Q=0.0, mult=1, thr=1, in_cm=1, sigma=None, adist=el_ad, edists=[]),
        dict(MT=102, Q=4.1, mult=0, thr=1, in_cm=0, sigma=0.1 / np.sqrt(energy / 1e-11), adist=None, edists=[]),
        dict(MT=51, Q=-6.05, mult=1, thr=thr[51], in_cm=1, sigma=sigma(51, 0.2), adist=lvl_ad,
             edists=[dict(law=3, data=np.array([6.43, 0.885]), pv_x=None, pv_y=None)]),
        dict(MT=91, Q=-7.2, mult=1, thr=thr[91], in_cm=1, sigma=sigma(91, 0.3), adist=None,
             edists=[dict(law=44, data=ace_edist(44, np.array([1.0, 5.0, 20.0]), 5, 9, seed=91, inttp=2),
                          pv_x=pv[0], pv_y=pv[1])]),
        dict(MT=22, Q=-2.5, mult=1, thr=thr[22], in_cm=0, sigma=sigma(22, 0.4), adist=None,
             edists=[dict(law=61, data=ace_edist(61, np.array([1.0, 5.0, 20.0]), 5, 9, seed=22),
                          pv_x=[1e-5, 20.0], pv_y=[0.9, 1.0])],
             mult_E=([1e-5, 10.0, 20.0], [1.0, 1.5, 2.0])),
        dict(MT=18, Q=190.0, mult=1, thr=1, in_cm=0, sigma=np.ones(n_grid), adist=None, edists=[]),
    ]
    return dict(awr=15.8575, kT=kT, freegas_cutoff=4.0 * kT, energy=energy, elastic=elastic,
                reactions=reactions, bins=np.array([0.0, 6.25e-7, 0.1, 20.0]),
                order=2, mu_bins=129, extend_pts=3, inel_extend_pts=4)


def pack_nuclide(d):
    """Flat (ints, doubles) encoding of nuclide_case() for oracle/ref_shim.f90:ref_calc_scatt."""
    I, D = [len(d["energy"]), len(d["reactions"])], [d["awr"], d["kT"], d["freegas_cutoff"]]
    D += list(d["energy"]) + list(d["elastic"])
    for r in d["reactions"]:
        sig = [] if r["sigma"] is None else list(r["sigma"])
        ad = r["adist"]
        mE = r.get("mult_E")
        I += [r["MT"], r["mult"], r["thr"], r["in_cm"], len(sig), 0 if ad is None else 1,
              0 if ad is None else len(ad[0]), 0 if ad is None else len(ad[3]), len(r["edists"]),
              0 if mE is None else len(mE[0])]
        D += [r["Q"]] + sig
        if ad is not None:
            I += list(ad[1]) + list(ad[2])
            D += list(ad[0]) + list(ad[3])
        if mE is not None:
            D += list(mE[0]) + list(mE[1])
        for ed in r["edists"]:
            npv = 0 if ed["pv_x"] is None else len(ed["pv_x"])
            I += [ed["law"], len(ed["data"]), npv]
            D += list(ed["data"])
            if npv:
                D += list(ed["pv_x"]) + list(ed["pv_y"])
    return np.array(I, dtype=np.int32), np.array(D, dtype=np.float64)


# ---- BASELINE configs[2] / SURVEY 8(d) #3: a U-238-like nuclide -----------------------------
def _lin_table(cs, pdf):
    """[JJ=2, NP, cosines, pdf, cdf] of one lin-lin ACE angular table, normalised."""
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(cs))])
    return [2.0, float(len(cs))] + list(cs) + list(pdf / cdf[-1]) + list(cdf / cdf[-1])


def _forward_adist(energies, a_of, b_of, npts=33):
    """DistAngle with an isotropic first row and tabular lin-lin rows
    f = 1/2 (1 + a(E) mu + b(E) P2(mu)) on npts cosines."""
    cs = np.linspace(-1.0, 1.0, npts)
    data, typ, loc = [0.0], [1], [0]
    for E in energies[1:]:
        typ.append(3)
        loc.append(len(data))
        data += _lin_table(cs, 0.5 * (1 + a_of(E) * cs + b_of(E) * (1.5 * cs * cs - 0.5)))
    return (np.asarray(energies, dtype=np.float64), np.array(typ, dtype=np.int32),
            np.array(loc, dtype=np.int32), np.array(data))


def u238_case(n_grid=50000, n_levels=40, n_el_rows=200, groups=2, order=7, mu_bins=2001,
              freegas_cutoff_kT=400.0, extend_pts=50, inel_extend_pts=30):
    """The nuclide of BASELINE configs[2] as SURVEY 8(d) #3 makes it concrete: A = 236.0058,
    293.6 K; elastic with n_el_rows tabular lin-lin angular tables (33 cosines) on 1e-5..20 MeV;
    n_levels level reactions (MT 51...), Q_k = -(0.0449 + 0.05 k) MeV, CM, isotropic -> mildly
    forward; MT 91 law-44 continuum in CM (30 incoming energies x 40 outgoing points); MT 22
    law 4 + angular table in the lab; MT 16 (n,2n) evaporation (law 9), multiplicity 2.
    n_grid log-spaced nuclide energies.  groups = 2 (the shipped structure) or 70 (log grid)."""
    awr, kT = 236.0058, 2.53e-8
    energy = 1e-11 * (20.0 / 1e-11) ** (np.arange(n_grid) / (n_grid - 1.0))
    energy[-1] = 20.0
    elastic = 9.0 + 3.0 / (1.0 + 50.0 * energy)
    el_E = np.concatenate([[1e-11], np.logspace(-5, np.log10(20.0), n_el_rows)])
    el_E[-1] = 20.0
    el_ad = _forward_adist(el_E, lambda E: 0.8 * E / 20.0, lambda E: 0.5 * (E / 20.0) ** 2)

    def thr_of(Q):  # first grid point at or above the kinematic threshold
        return int(np.searchsorted(energy, -Q * (awr + 1.0) / awr, side="left")) + 1

    def sigma_of(Q, thr, step):
        E = energy[thr - 1:]
        return step * (1.0 - np.exp(-(E + Q * (awr + 1.0) / awr).clip(0.0) / 0.1)) + 1e-6

    reactions = [dict(MT=2, Q=0.0, mult=1, thr=1, in_cm=1, sigma=None, adist=el_ad, edists=[]),
                 dict(MT=102, Q=4.8, mult=0, thr=1, in_cm=0, sigma=2.7 / np.sqrt(energy / 2.53e-8), adist=None,
                      edists=[])]
    for k in range(n_levels):
        Q = -(0.0449 + 0.05 * k)
        thr = thr_of(Q)
        ad = _forward_adist([energy[thr - 1], 20.0], lambda E: 0.3, lambda E: 0.1)
        reactions.append(dict(MT=51 + k, Q=Q, mult=1, thr=thr, in_cm=1, sigma=sigma_of(Q, thr, 0.05 + 0.002 * k),
                              adist=ad, edists=[dict(law=3, data=np.array([-Q * (awr + 1.0) / awr,
                                                                           (awr / (awr + 1.0)) ** 2]),
                                                     pv_x=None, pv_y=None)]))
    pv = ([1e-11, 20.0], [1.0, 1.0])      # every energy distribution carries its p_valid TAB1
    Qc = -(0.0449 + 0.05 * n_levels)
    thr = thr_of(Qc)
    e44 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), 30)
    e44[0], e44[-1] = energy[thr - 1], 20.0
    reactions.append(dict(MT=91, Q=Qc, mult=1, thr=thr, in_cm=1, sigma=sigma_of(Qc, thr, 1.2), adist=None,
                          edists=[dict(law=44, data=ace_edist(44, e44, 40, 40, seed=238), pv_x=pv[0], pv_y=pv[1])]))
    Q22 = -4.0
    thr = thr_of(Q22)
    e4 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), 12)
    e4[0], e4[-1] = energy[thr - 1], 20.0
    # (a reaction with both an angular and a law-4 / law-9 energy distribution is converted row by
    # row on the ENERGY distribution's incoming grid, scattdata_header.F90:236-250: the angular
    # tables must sit on the same energies)
    ad22 = _forward_adist(list(e4), lambda E: 0.4 * E / 20.0, lambda E: 0.0)
    reactions.append(dict(MT=22, Q=Q22, mult=1, thr=thr, in_cm=0, sigma=sigma_of(Q22, thr, 0.1), adist=ad22,
                          edists=[dict(law=4, data=ace_edist(4, e4, 20, 20, seed=22), pv_x=pv[0], pv_y=pv[1])]))
    Q16 = -6.15
    thr = thr_of(Q16)
    e9 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), 8)
    ad16 = _forward_adist(list(e9), lambda E: 0.2, lambda E: 0.0)
    reactions.append(dict(MT=16, Q=Q16, mult=2, thr=thr, in_cm=0, sigma=sigma_of(Q16, thr, 0.8), adist=ad16,
                          edists=[dict(law=9, data=law9_edata(energy[thr - 1], 20.0, n=8, U=-Q16 * (awr + 1) / awr),
                                       pv_x=pv[0], pv_y=pv[1])]))
    reactions.append(dict(MT=18, Q=190.0, mult=1, thr=1, in_cm=0, sigma=np.full(n_grid, 1e-5), adist=None, edists=[]))
    bins = np.array([0.0, 6.25e-7, 20.0]) if groups == 2 else \
        np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), groups)])
    bins[-1] = 20.0
    return dict(awr=awr, kT=kT, freegas_cutoff=freegas_cutoff_kT * kT, energy=energy, elastic=elastic,
                reactions=reactions, bins=bins, order=order, mu_bins=mu_bins, extend_pts=extend_pts,
                inel_extend_pts=inel_extend_pts)


def library_nuclide(awr, seed, n_grid=None, kT=2.53e-8, groups=2, order=5, mu_bins=2001,
                    freegas_cutoff_kT=400.0, scale=1.0, extend_pts=50, inel_extend_pts=30):
    """One nuclide of the synthetic library of BASELINE configs[4] / SURVEY 8(d) #5: the shape of
    u238_case with table sizes drawn (seeded) around the config-3 sizes and scaled with the mass:
    elastic angular tables on every nuclide; level reactions, a law-44 continuum (CM), a law-4
    reaction with an angular table (lab) and an (n,2n) evaporation spectrum appear with
    increasing mass.  The nuclide grid has n_grid log-spaced energies, about half of them below
    the free-gas cutoff, so that the free-gas part of the incoming grid -- where the time goes --
    has 200..800 points with the ~150 points the grid builder adds around the group edges."""
    rng = np.random.default_rng(seed)
    if n_grid is None:
        n_grid = int(rng.integers(100, 1301) * scale) + 20
    energy = 1e-11 * (20.0 / 1e-11) ** (np.arange(n_grid) / (n_grid - 1.0))
    energy[-1] = 20.0
    elastic = 4.0 + 16.0 * rng.uniform(0.2, 1.0) / (1.0 + 30.0 * energy)
    n_el_rows = int(np.clip(rng.lognormal(np.log(10 + 150 * awr / 240.0), 0.4), 3, 300))
    el_E = np.concatenate([[1e-11], np.logspace(-5, np.log10(20.0), n_el_rows)])
    el_E[-1] = 20.0
    a1, b1 = rng.uniform(0.3, 0.9), rng.uniform(0.1, 0.5)
    el_ad = _forward_adist(el_E, lambda E: a1 * E / 20.0, lambda E: b1 * (E / 20.0) ** 2)

    def thr_of(Q):
        return min(int(np.searchsorted(energy, -Q * (awr + 1.0) / awr, side="left")) + 1, n_grid - 2)

    def sigma_of(Q, thr, step):
        E = energy[thr - 1:]
        return step * (1.0 - np.exp(-(E + Q * (awr + 1.0) / awr).clip(0.0) / 0.1)) + 1e-6

    reactions = [dict(MT=2, Q=0.0, mult=1, thr=1, in_cm=1, sigma=None, adist=el_ad, edists=[]),
                 dict(MT=102, Q=4.8, mult=0, thr=1, in_cm=0, sigma=2.7 / np.sqrt(energy / 2.53e-8), adist=None,
                      edists=[])]
    pv = ([1e-11, 20.0], [1.0, 1.0])
    n_levels = 0 if awr < 4.0 else int(np.clip(rng.lognormal(np.log(2 + 38 * awr / 240.0), 0.3), 1, 40))
    q0 = 0.0449 + 2.0 / max(awr, 4.0)
    for k in range(n_levels):
        Q = -(q0 + 0.05 * k)
        thr = thr_of(Q)
        ad = _forward_adist([energy[thr - 1], 20.0], lambda E: 0.3, lambda E: 0.1)
        reactions.append(dict(MT=51 + k, Q=Q, mult=1, thr=thr, in_cm=1, sigma=sigma_of(Q, thr, 0.05 + 0.002 * k),
                              adist=ad, edists=[dict(law=3, data=np.array([-Q * (awr + 1.0) / awr,
                                                                           (awr / (awr + 1.0)) ** 2]),
                                                     pv_x=None, pv_y=None)]))
    if awr >= 10.0:
        Qc = -(q0 + 0.05 * n_levels)
        thr = thr_of(Qc)
        ne44 = int(np.clip(rng.lognormal(np.log(8 + 22 * awr / 240.0), 0.3), 4, 40))
        np44 = int(np.clip(rng.lognormal(np.log(10 + 30 * awr / 240.0), 0.3), 6, 60))
        e44 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), ne44)
        e44[0], e44[-1] = energy[thr - 1], 20.0
        reactions.append(dict(MT=91, Q=Qc, mult=1, thr=thr, in_cm=1, sigma=sigma_of(Qc, thr, 1.2), adist=None,
                              edists=[dict(law=44, data=ace_edist(44, e44, np44, np44, seed=seed + 1), pv_x=pv[0],
                                           pv_y=pv[1])]))
    if awr >= 20.0:
        Q22 = -4.0
        thr = thr_of(Q22)
        e4 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), 8)
        e4[0], e4[-1] = energy[thr - 1], 20.0
        ad22 = _forward_adist(list(e4), lambda E: 0.4 * E / 20.0, lambda E: 0.0)
        reactions.append(dict(MT=22, Q=Q22, mult=1, thr=thr, in_cm=0, sigma=sigma_of(Q22, thr, 0.1), adist=ad22,
                              edists=[dict(law=4, data=ace_edist(4, e4, 12, 12, seed=seed + 2), pv_x=pv[0],
                                           pv_y=pv[1])]))
    if awr >= 30.0:
        Q16 = -6.15
        thr = thr_of(Q16)
        e9 = np.logspace(np.log10(energy[thr - 1]), np.log10(20.0), 6)
        ad16 = _forward_adist(list(e9), lambda E: 0.2, lambda E: 0.0)
        reactions.append(dict(MT=16, Q=Q16, mult=2, thr=thr, in_cm=0, sigma=sigma_of(Q16, thr, 0.8), adist=ad16,
                              edists=[dict(law=9, data=law9_edata(energy[thr - 1], 20.0, n=6,
                                                                  U=-Q16 * (awr + 1) / awr),
                                           pv_x=pv[0], pv_y=pv[1])]))
    bins = np.array([0.0, 6.25e-7, 20.0]) if groups == 2 else \
        np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), groups)])
    bins[-1] = 20.0
    return dict(awr=float(awr), kT=kT, freegas_cutoff=freegas_cutoff_kT * kT, energy=energy, elastic=elastic,
                reactions=reactions, bins=bins, order=order, mu_bins=mu_bins, extend_pts=extend_pts,
                inel_extend_pts=inel_extend_pts)


def synthetic_library(n_nuclides=423, n_thermal=20, n_fissionable=30, seed=2024, scale=1.0, order=5, **nuc_kw):
    """The library of SURVEY 8(d) #5: n_nuclides nuclide descriptors (masses log-uniform in
    [1, 250], the .71c count of the reference's NNDC listing), n_thermal thermal tables (half
    discrete, half continuous, two with elastic parts) and chi inputs for the n_fissionable
    heaviest nuclides.  Everything seeded; `scale` shrinks the grids for tests, nuc_kw goes to
    library_nuclide (e.g. a small free-gas region so that the reference can afford goldens)."""
    rng = np.random.default_rng(seed)
    awr = np.sort(np.exp(rng.uniform(np.log(1.0), np.log(250.0), n_nuclides)))
    nucs = [library_nuclide(awr[k], seed=seed * 1000 + k, scale=scale, order=order, **nuc_kw) for k in range(n_nuclides)]
    thermal = []
    for k in range(n_thermal):
        mode = 1 if k % 2 == 0 else 2
        el = "coherent" if k == 3 else ("incoherent" if k == 7 else None)
        nei = max(8, int(116 * scale))
        thermal.append(sab_table(mode, seed=seed + 100 + k, NEi=nei, NEo=max(8, int(64 * scale)) if mode == 1 else 16,
                                 NMU=16 if mode == 1 else 20, elastic=el))
    fissionable = list(range(n_nuclides - n_fissionable, n_nuclides))
    chis = [chi_case(seed=seed + 500 + k) for k in range(n_fissionable)]
    return dict(awr=awr, nuclides=nucs, thermal=thermal, fissionable=fissionable, chi=chis)
