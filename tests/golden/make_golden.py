"""Mint golden vectors from the genuine reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/pycmf (with the one in-process alias its 2018-era
sklearn import needs), runs its solver on small seeded inputs
and writes inputs + outputs to tests/golden/*.npz.  The reference itself never
travels; only these data files do.  Versions used: see ``meta`` in each file.
"""
import os
import sys
import warnings

import numpy as np
import scipy
import scipy.sparse as sp
import sklearn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CMF_REFERENCE", "/root/reference")


def _import_reference():
    import sklearn.decomposition._nmf as _nmf
    sys.modules.setdefault("sklearn.decomposition.nmf", _nmf)
    sys.path.insert(0, REF)
    import pycmf  # noqa
    from pycmf import cmf_solvers, cmf
    return pycmf, cmf_solvers, cmf


pycmf, RS, RC = _import_reference()
META = "numpy %s / scipy %s / sklearn %s" % (np.__version__, scipy.__version__, sklearn.__version__)


def save(name, **arrs):
    arrs["meta"] = np.array(META)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items() if k != "meta"})


def base_problem(seed, m, d, p, k, signed=False):
    rng = np.random.RandomState(seed)
    f = (lambda a: a) if signed else np.abs
    X = f(rng.randn(m, d))
    Y = f(rng.randn(d, p))
    U = f(rng.randn(m, k))
    V = f(rng.randn(d, k))
    Z = f(rng.randn(p, k))
    return X, Y, U, V, Z


# ---------------------------------------------------------------- G1 README
def g1_readme():
    rng = np.random.RandomState(0)
    X = rng.randn(5, 4) ** 2
    Y = rng.randn(4, 1) ** 2
    model = pycmf.CMF(n_components=4, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U, V, Z = model.fit_transform(X, Y)
    save("g1_readme", X=X, Y=Y, U=U, V=V, Z=Z, n_iter=np.array(model.n_iter_),
         err=np.array(model.reconstruction_err_))


# ---------------------------------------------------------------- G2 MU steps
def g2_mu():
    out = {}
    X, Y, U0, V0, Z0 = base_problem(7, 9, 8, 5, 3)
    out.update(X=X, Y=Y, U0=U0, V0=V0, Z0=Z0)
    for tag, (l1, l2) in {"plain": (0.0, 0.0), "reg": (0.3, 0.7)}.items():
        for fmt in ("dense", "csr"):
            Xi = sp.csr_matrix(X) if fmt == "csr" else X
            s = RS.MUSolver(l1_reg=l1, l2_reg=l2)
            U, V, Z = U0.copy(), V0.copy(), Z0.copy()
            for it in range(1, 11):
                s.update_step(Xi, Y, U, V, Z, l1, l2, 0.5)
                if it in (1, 10):
                    out["%s_%s_U%d" % (tag, fmt, it)] = U.copy()
                    out["%s_%s_V%d" % (tag, fmt, it)] = V.copy()
                    out["%s_%s_Z%d" % (tag, fmt, it)] = Z.copy()
    # signed data / signed factors (MU never clamps; tests/test_cmf.py:374-392)
    Xs, Ys, Us, Vs, Zs = base_problem(11, 7, 6, 4, 3, signed=True)
    s = RS.MUSolver()
    U, V, Z = Us.copy(), Vs.copy(), Zs.copy()
    s.update_step(Xs, Ys, U, V, Z, 0.0, 0.0, 0.5)
    out.update(sX=Xs, sY=Ys, sU0=Us, sV0=Vs, sZ0=Zs, sU1=U, sV1=V, sZ1=Z)
    # exact-zero denominator -> eps rule (cmf_solvers.py:219): a zero row of V
    # and zero U column make den exactly 0 in places
    Xz, Yz, Uz, Vz, Zz = base_problem(13, 6, 5, 4, 3)
    Uz[:, 1] = 0.0
    Zz[:, 1] = 0.0
    U, V, Z = Uz.copy(), Vz.copy(), Zz.copy()
    s.update_step(Xz, Yz, U, V, Z, 0.0, 0.0, 0.5)
    out.update(zX=Xz, zY=Yz, zU0=Uz, zV0=Vz, zZ0=Zz, zU1=U, zV1=V, zZ1=Z)
    # partial updates (transform(): update_V False; cmf.py:726-747)
    s2 = RS.MUSolver(update_V=False, update_Z=False)
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    s2.update_step(X, Y, U, V, Z, 0.0, 0.0, 0.5)
    out.update(pU1=U, pV1=V, pZ1=Z)
    save("g2_mu_steps", **out)


# ---------------------------------------------------------------- G3 Newton steps
NEWTON_CASES = {
    # name: (x_link, y_link, nn, ratio, seed, l1, l2, signed_init)
    "lin_lin_nn": ("linear", "linear", True, 1.0, None, 0.1, 0.2, False),
    "lin_log_nn": ("linear", "logit", True, 1.0, None, 0.1, 0.2, False),
    "log_log_free": ("logit", "logit", False, 1.0, None, 0.1, 0.2, True),
    "log_lin_free": ("logit", "linear", False, 1.0, None, 0.0, 0.0, True),
    "lin_log_free_sg": ("linear", "logit", False, 0.5, 3, 0.1, 0.2, True),
    "log_log_nn_sg": ("logit", "logit", True, 0.5, 5, 0.05, 0.1, False),
    "lin_lin_free_sg": ("linear", "linear", False, 0.5, 9, 0.0, 0.3, True),
}


def _record_masks(solver):
    """Wrap the reference's sampler so the drawn index sets are captured."""
    drawn = []
    orig = np.random.permutation

    def spy(a):
        r = orig(a)
        drawn.append(np.array(r))
        return r
    return drawn, orig, spy


def g3_newton():
    out = {}
    X, Y, U0p, V0p, Z0p = base_problem(7, 9, 8, 5, 3)
    _, _, U0s, V0s, Z0s = base_problem(7, 9, 8, 5, 3, signed=True)
    Ylog = 1.0 / (1.0 + np.exp(-np.random.RandomState(21).randn(8, 5)))
    Xlog = 1.0 / (1.0 + np.exp(-np.random.RandomState(22).randn(9, 8)))
    out.update(X=X, Y=Y, Xlog=Xlog, Ylog=Ylog, U0p=U0p, V0p=V0p, Z0p=Z0p,
               U0s=U0s * 0.3, V0s=V0s * 0.3, Z0s=Z0s * 0.3)
    alpha, pert = 0.3, 0.2
    for name, (xl, yl, nn, ratio, seed, l1, l2, signed) in NEWTON_CASES.items():
        Xi = Xlog if xl == "logit" else X
        Yi = Ylog if yl == "logit" else Y
        for fmt in ("dense", "csr"):
            Xf = sp.csr_matrix(Xi) if fmt == "csr" else Xi
            s = RS.NewtonSolver(alpha=alpha, l1_reg=l1, l2_reg=l2, x_link=xl, y_link=yl,
                                U_non_negative=nn, V_non_negative=nn, Z_non_negative=nn,
                                hessian_pertubation=pert, sg_sample_ratio=ratio,
                                random_state=seed)
            if signed:
                U, V, Z = out["U0s"].copy(), out["V0s"].copy(), out["Z0s"].copy()
            else:
                U, V, Z = U0p.copy(), V0p.copy(), Z0p.copy()
            drawn, orig, spy = _record_masks(s)
            np.random.permutation = spy
            try:
                for it in range(1, 4):
                    s.update_step(Xf, Yi, U, V, Z, l1, l2, alpha)
                    if it in (1, 3):
                        out["%s_%s_U%d" % (name, fmt, it)] = U.copy()
                        out["%s_%s_V%d" % (name, fmt, it)] = V.copy()
                        out["%s_%s_Z%d" % (name, fmt, it)] = Z.copy()
            finally:
                np.random.permutation = orig
            if drawn and fmt == "dense":
                # flat log of all permutations (truncated to the sample) in draw order
                m, d, p = X.shape[0], X.shape[1], Y.shape[1]
                out[name + "_draws"] = np.concatenate(
                    [r[: int(len(r) * ratio)] for r in drawn]).astype(np.int32)
    save("g3_newton_steps", **out)


# ---------------------------------------------------------------- G4 fit level
def g4_fit():
    out = {}
    rng = np.random.mtrand.RandomState(42)
    X = np.abs(rng.randn(6, 5))
    Y = np.abs(rng.randn(5, 6))
    out.update(fc_X=X, fc_Y=Y)
    for solver in ("mu", "newton"):
        # custom init so the result does not depend on sklearn's randomized_svd
        r2 = np.random.RandomState(1)
        U0, V0, Z0 = np.abs(r2.randn(6, 5)), np.abs(r2.randn(5, 5)), np.abs(r2.randn(6, 5))
        m = pycmf.CMF(n_components=5, solver=solver, x_init="custom", y_init="custom",
                      random_state=0, max_iter=1000)
        U, V, Z = m.fit_transform(X, Y, U=U0.copy(), V=V0.copy(), Z=Z0.copy())
        out.update({"fc_U0": U0, "fc_V0": V0, "fc_Z0": Z0,
                    "fc_%s_U" % solver: U, "fc_%s_V" % solver: V, "fc_%s_Z" % solver: Z,
                    "fc_%s_n_iter" % solver: np.array(m.n_iter_),
                    "fc_%s_err" % solver: np.array(m.reconstruction_err_)})
        # reference's own init path (nndsvdar) for the record
        m2 = pycmf.CMF(n_components=5, solver=solver, x_init="nndsvdar", y_init="nndsvdar",
                       random_state=0, max_iter=1000)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m2.fit(X, Y)
        out["fc_%s_nndsvdar_n_iter" % solver] = np.array(m2.n_iter_)
        out["fc_%s_nndsvdar_err" % solver] = np.array(m2.reconstruction_err_)
    # logit fit (tests/test_cmf.py:239-250 shape)
    rng = np.random.mtrand.RandomState(42)
    Xl = np.abs(rng.randn(6, 5))
    Yl = 1.0 / (1.0 + np.exp(-rng.randn(5, 6)))
    m = pycmf.CMF(n_components=5, solver="newton", y_link="logit", random_state=42,
                  max_iter=200, U_non_negative=False, V_non_negative=False, Z_non_negative=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U, V, Z = m.fit_transform(Xl, Yl)
    out.update(lg_X=Xl, lg_Y=Yl, lg_U=U, lg_V=V, lg_Z=Z, lg_n_iter=np.array(m.n_iter_),
               lg_err=np.array(m.reconstruction_err_))
    # transform after fit (tests/test_cmf.py:374-408)
    rng = np.random.mtrand.RandomState(42)
    Xt = np.abs(rng.randn(8, 6))
    Yt = np.abs(rng.randn(6, 4))
    for solver in ("mu", "newton"):
        m = pycmf.CMF(n_components=3, solver=solver, x_init="random", y_init="random",
                      random_state=0, max_iter=60)
        U, V, Z = m.fit_transform(Xt, Yt)
        Ut, Vt, Zt = m.transform(Xt, None)
        out.update({"tr_X": Xt, "tr_Y": Yt, "tr_%s_U" % solver: U, "tr_%s_V" % solver: V,
                    "tr_%s_Z" % solver: Z, "tr_%s_Ut" % solver: Ut, "tr_%s_Vt" % solver: Vt,
                    "tr_%s_n_iter" % solver: np.array(m.n_iter_)})
    save("g4_fit_level", **out)


# ---------------------------------------------------------------- G5 initialisers
def g5_init():
    out = {}
    rng = np.random.RandomState(5)
    M = np.abs(rng.randn(12, 7))
    Ms = rng.randn(12, 7)
    out.update(M=M, Ms=Ms)
    for init in ("random", "nndsvd", "nndsvda", "nndsvdar"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            A, B = RC._initialize_mf(M, 4, init=init, random_state=3, non_negative=True)
        out["%s_nn_A" % init] = A
        out["%s_nn_B" % init] = B
    for init in ("random", "svd"):
        A, B = RC._initialize_mf(Ms, 4, init=init, random_state=3, non_negative=False)
        out["%s_free_A" % init] = A
        out["%s_free_B" % init] = B
    # k > n_features: svd zero-padding (cmf.py:129-138)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        A, B = RC._initialize_mf(Ms, 9, init="svd", random_state=3, non_negative=False)
    out["svd_pad_A"] = A
    out["svd_pad_B"] = B
    # driver-level init + V merge (cmf.py:402-430) through max_iter=0-like probe:
    # capture what the solver receives by running 1 MU iteration from 'random'
    Y = np.abs(rng.randn(7, 5))
    U, V, Z, n_it = pycmf.collective_matrix_factorization(
        M, Y, n_components=4, x_init="random", y_init="random", solver="mu",
        max_iter=1, random_state=3)
    out.update(drv_Y=Y, drv_U1=U, drv_V1=V, drv_Z1=Z)
    save("g5_init", **out)


# ---------------------------------------------------------------- G6 mid-size MU
def g6_mid():
    X, Y, U0, V0, Z0 = base_problem(42, 256, 192, 96, 32)
    scale = np.sqrt(X.mean() / 32)
    U0 *= scale
    V0 *= scale
    Z0 *= scale
    s = RS.MUSolver()
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    errs = []
    for it in range(20):
        s.update_step(X, Y, U, V, Z, 0.0, 0.0, 0.5)
        errs.append([np.linalg.norm(X - U @ V.T), np.linalg.norm(Y - V @ Z.T)])
    # inputs are regenerated from the seed by the test (base_problem(42,...)); store
    # only outputs to keep the fixture small
    save("g6_mid_mu", U=U.astype(np.float32), V=V.astype(np.float32), Z=Z.astype(np.float32),
         errs=np.array(errs), normX=np.array(np.linalg.norm(X)), normY=np.array(np.linalg.norm(Y)))




# ---------------------------------------------------------------- G7 Cython twin (dead code upstream)
def g7_cython():
    """pycmf/cmf_newton_solver.pyx is never imported by the reference; it is compiled here from its
    source where it lies (build dir: $CMF_CYTHON_BUILD, default /tmp/cybuild; see the recipe in
    tests/golden/README.md) only to pin the N-cy rows of SURVEY 8(a)."""
    build = os.environ.get("CMF_CYTHON_BUILD", "/tmp/cybuild")
    sys.path.insert(0, build)
    try:
        import cmf_newton_solver as cy
    except ImportError:
        print("skip g7: compiled Cython twin not found in", build)
        return
    out = {}
    X, Y, U0p, V0p, Z0p = base_problem(7, 9, 8, 5, 3)
    _, _, U0s, V0s, Z0s = base_problem(7, 9, 8, 5, 3, signed=True)
    Ylog = 1.0 / (1.0 + np.exp(-np.random.RandomState(21).randn(8, 5)))
    Xlog = 1.0 / (1.0 + np.exp(-np.random.RandomState(22).randn(9, 8)))
    alpha, pert = 0.3, 0.2
    cases = {"lin_log_nn": ("linear", "logit", True, 1.0, None, 0.1, 0.2, False),
             "log_log_free": ("logit", "logit", False, 1.0, None, 0.1, 0.2, True),
             "lin_log_free_sg": ("linear", "logit", False, 0.5, 3, 0.1, 0.2, True)}
    for name, (xl, yl, nn, ratio, seed, l1, l2, signed) in cases.items():
        Xi = Xlog if xl == "logit" else X
        Yt = np.ascontiguousarray((Ylog if yl == "logit" else Y).T)      # the Cython path solves Y^T ~ f(Z V^T)
        if signed:
            U, V, Z = 0.3 * U0s, 0.3 * V0s, 0.3 * Z0s
        else:
            U, V, Z = U0p.copy(), V0p.copy(), Z0p.copy()
        U, V, Z = np.ascontiguousarray(U), np.ascontiguousarray(V), np.ascontiguousarray(Z)
        if seed is not None:
            np.random.seed(seed)
        # NewtonSolver.update_step of the USE_CYTHON branch, cmf_solvers.py:292-311
        cy._newton_update_left(U, V, Xi, alpha, l1, l2, xl, nn, ratio, pert)
        cy._newton_update_left(Z, V, Yt, 1 - alpha, l1, l2, yl, nn, ratio, pert)
        cy._newton_update_V(V, U, Z, Xi, Yt, alpha, l1, l2, xl, yl, nn, ratio, pert)
        out[name + "_U1"], out[name + "_V1"], out[name + "_Z1"] = U, V, Z
    save("g7_cython_steps", **out)


if __name__ == "__main__":
    g1_readme()
    g2_mu()
    g3_newton()
    g4_fit()
    g5_init()
    g6_mid()
    g7_cython()
