"""CPU emulation: which accumulation / storage precision of the linear Newton sweeps reaches north_star's 1e-4 on the
clamped non-negative case of tests/test_gpu_shared64.py (linear_nonneg)?  float64 Hessian + inverse throughout (as the
device does); data and factors stored in float32 (as on the device)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import cmf_oracle as O
from tests.test_gpu_shared64 import PARITY_CASES, _make_problem

f32 = np.float32

def run(mode, X, Y, U0, V0, Z0, alpha, l2, nn, pert=0.2, iters=8):
    X32, Y32 = X.astype(f32), Y.astype(f32)
    U, V, Z = U0.astype(f32), V0.astype(f32), Z0.astype(f32)
    Xd, Yd = X32.astype(np.float64), Y32.astype(np.float64)

    def prod(A32, B32):
        if mode == "f32":
            return A32 @ B32                       # float32 accumulate
        r = A32.astype(np.float64) @ B32.astype(np.float64)
        return r.astype(f32) if mode == "f64acc_f32store" else r

    def sweep(F, TO, G64, s):
        H = s * G64 + l2 * np.eye(G64.shape[0])
        Hinv = O.safe_invert(H, pert)
        if mode == "f32":
            FG = F @ G64.astype(f32)
            grad = (f32(s) * (FG - TO) + f32(l2) * F).astype(f32)
            step = grad @ Hinv.astype(f32)
            Fn = F - step
        elif mode == "f64acc_f32store":
            FG = (F.astype(np.float64) @ G64).astype(f32)
            grad = (f32(s) * (FG - TO) + f32(l2) * F).astype(f32)
            step = (grad.astype(np.float64) @ Hinv).astype(f32)
            Fn = F - step
        else:  # f64 chain, only the factor is rounded
            Fd = F.astype(np.float64)
            grad = s * (Fd @ G64 - TO) + l2 * Fd
            Fn = Fd - grad @ Hinv
        if nn:
            Fn = np.maximum(Fn, 0)
        return Fn.astype(f32)

    for _ in range(iters):
        Vd = V.astype(np.float64)
        G = Vd.T @ Vd
        U = sweep(U, prod(X32, V), G, alpha)
        Z = sweep(Z, prod(Y32.T.copy(), V), G, 1 - alpha)
        Ud, Zd = U.astype(np.float64), Z.astype(np.float64)
        Gm = alpha * Ud.T @ Ud + (1 - alpha) * Zd.T @ Zd
        if mode == "f32":
            P = f32(alpha) * (X32.T @ U) + f32(1 - alpha) * (Y32 @ Z)
        elif mode == "f64acc_f32store":
            P = f32(alpha) * prod(X32.T.copy(), U) + f32(1 - alpha) * prod(Y32, Z)
        else:
            P = alpha * (Xd.T @ Ud) + (1 - alpha) * (Yd @ Zd)
        V = sweep(V, P, Gm, 1.0)
    return U.astype(np.float64), V.astype(np.float64), Z.astype(np.float64)

for name in ("linear_nonneg", "linear_signed"):
    case, ratio = PARITY_CASES[name]
    m, d, p, k, xl, yl, l2, nn, signed = case
    X, Y, U0, V0, Z0 = _make_problem(case, np.random.RandomState(11))
    Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(8):
        O.newton_update_step(X, Y, Uo, Vo, Zo, 0.5, 0.0, l2, "linear", "linear", nn, nn, nn, 1.0, 0.2)
    ex = O.factorization_error(X, Uo, Vo.T, "linear"); ey = O.factorization_error(Y, Vo, Zo.T, "linear")
    for mode in ("f32", "f64acc_f32store", "f64chain"):
        U, V, Z = run(mode, X, Y, U0, V0, Z0, 0.5, l2, nn)
        gx = O.factorization_error(X, U, V.T, "linear"); gy = O.factorization_error(Y, V, Z.T, "linear")
        print(name, mode, "rel resid diff X %.2e Y %.2e  maxdiff V %.2e" % (abs(gx - ex) / ex, abs(gy - ey) / ey, np.abs(V - Vo).max() / np.abs(Vo).max()))

print("--- re-associated form: F_new = F (I - H Hinv) + s T (O Hinv) ---")
def run2(mode, X, Y, U0, V0, Z0, alpha, l2, nn, pert=0.2, iters=8):
    X32, Y32 = X.astype(f32), Y.astype(f32)
    U, V, Z = U0.astype(f32), V0.astype(f32), Z0.astype(f32)
    def big(A32, B64):          # T (O') product
        if mode == "f32":
            return (A32 @ B64.astype(f32)).astype(np.float64)
        if mode == "split":     # O' as hi + lo float32 planes, two float32 products
            hi = B64.astype(f32); lo = (B64 - hi).astype(f32)
            return (A32 @ hi).astype(np.float64) + (A32 @ lo).astype(np.float64)
        if mode == "f64acc":    # O' rounded to f32, product accumulated in f64, stored f32
            return (A32.astype(np.float64) @ B64.astype(f32).astype(np.float64)).astype(f32).astype(np.float64)
    def sweep(F, terms, G64, s):
        k = G64.shape[0]
        H = s * G64 + l2 * np.eye(k)
        Hinv = O.safe_invert(H, pert)
        E = np.eye(k) - H @ Hinv
        acc = (F @ E.astype(f32)).astype(np.float64)
        for sc, T32, O32 in terms:
            acc = acc + sc * big(T32, O32.astype(np.float64) @ Hinv)
        Fn = acc
        if nn:
            Fn = np.maximum(Fn, 0)
        return Fn.astype(f32)
    XT, YT = X32.T.copy(), Y32.T.copy()
    for _ in range(iters):
        Vd = V.astype(np.float64); G = Vd.T @ Vd
        U = sweep(U, [(alpha, X32, V)], G, alpha)
        Z = sweep(Z, [(1 - alpha, YT, V)], G, 1 - alpha)
        Ud, Zd = U.astype(np.float64), Z.astype(np.float64)
        Gm = alpha * Ud.T @ Ud + (1 - alpha) * Zd.T @ Zd
        V = sweep(V, [(alpha, XT, U), (1 - alpha, Y32, Z)], Gm, 1.0)
    return U.astype(np.float64), V.astype(np.float64), Z.astype(np.float64)

for name in ("linear_nonneg", "linear_signed"):
    case, ratio = PARITY_CASES[name]
    m, d, p, k, xl, yl, l2, nn, signed = case
    X, Y, U0, V0, Z0 = _make_problem(case, np.random.RandomState(11))
    Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(8):
        O.newton_update_step(X, Y, Uo, Vo, Zo, 0.5, 0.0, l2, "linear", "linear", nn, nn, nn, 1.0, 0.2)
    ex = O.factorization_error(X, Uo, Vo.T, "linear"); ey = O.factorization_error(Y, Vo, Zo.T, "linear")
    for mode in ("f32", "split", "f64acc"):
        U, V, Z = run2(mode, X, Y, U0, V0, Z0, 0.5, l2, nn)
        gx = O.factorization_error(X, U, V.T, "linear"); gy = O.factorization_error(Y, V, Z.T, "linear")
        print(name, mode, "rel resid diff X %.2e Y %.2e  maxdiff V %.2e" % (abs(gx - ex) / ex, abs(gy - ey) / ey, np.abs(V - Vo).max() / np.abs(Vo).max()))

print("--- re-associated form, O' = O Hinv itself in float32 arithmetic ---")
def run3(mode, X, Y, U0, V0, Z0, alpha, l2, nn, pert=0.2, iters=8):
    X32, Y32 = X.astype(f32), Y.astype(f32)
    U, V, Z = U0.astype(f32), V0.astype(f32), Z0.astype(f32)
    def oprime(O32, Hinv):
        if mode == "hinv32":
            return O32 @ Hinv.astype(f32)
        hi = Hinv.astype(f32); lo = (Hinv - hi).astype(f32)
        return O32 @ hi + O32 @ lo
    def sweep(F, terms, G64, s):
        k = G64.shape[0]
        H = s * G64 + l2 * np.eye(k)
        Hinv = O.safe_invert(H, pert)
        E = np.eye(k) - H @ Hinv
        acc = F @ E.astype(f32)
        for sc, T32, O32 in terms:
            acc = acc + f32(sc) * (T32 @ oprime(O32, Hinv))
        Fn = acc
        if nn:
            Fn = np.maximum(Fn, 0)
        return Fn.astype(f32)
    XT, YT = X32.T.copy(), Y32.T.copy()
    for _ in range(iters):
        Vd = V.astype(np.float64); G = Vd.T @ Vd
        U = sweep(U, [(alpha, X32, V)], G, alpha)
        Z = sweep(Z, [(1 - alpha, YT, V)], G, 1 - alpha)
        Ud, Zd = U.astype(np.float64), Z.astype(np.float64)
        Gm = alpha * Ud.T @ Ud + (1 - alpha) * Zd.T @ Zd
        V = sweep(V, [(alpha, XT, U), (1 - alpha, Y32, Z)], Gm, 1.0)
    return U.astype(np.float64), V.astype(np.float64), Z.astype(np.float64)

for name in ("linear_nonneg", "linear_signed"):
    case, ratio = PARITY_CASES[name]
    m, d, p, k, xl, yl, l2, nn, signed = case
    X, Y, U0, V0, Z0 = _make_problem(case, np.random.RandomState(11))
    Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(8):
        O.newton_update_step(X, Y, Uo, Vo, Zo, 0.5, 0.0, l2, "linear", "linear", nn, nn, nn, 1.0, 0.2)
    ex = O.factorization_error(X, Uo, Vo.T, "linear"); ey = O.factorization_error(Y, Vo, Zo.T, "linear")
    for mode in ("hinv32", "hinv_hilo"):
        U, V, Z = run3(mode, X, Y, U0, V0, Z0, 0.5, l2, nn)
        gx = O.factorization_error(X, U, V.T, "linear"); gy = O.factorization_error(Y, V, Z.T, "linear")
        print(name, mode, "rel resid diff X %.2e Y %.2e  maxdiff V %.2e" % (abs(gx - ex) / ex, abs(gy - ey) / ey, np.abs(V - Vo).max() / np.abs(Vo).max()))
