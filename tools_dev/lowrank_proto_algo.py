"""numpy float32 prototype of the LRKD subspace solver variants (round 5), run on the Gram matrices of tools_dev/lowrank_proto_grams.py.

What is being decided: how to reach the accuracy of the reference's per-batch exact svd (model/loss.py:318-326) from the previous batch's
basis with the fewest / cheapest device stages.  Stages and their device costs (measured in round 4, per call for 3 layers):
  mult    Y = G V                       ~20-35 us, all CUs
  cholqr  S = Y^T Y, L = chol(S), V = Y L^-T   (order preserving Gram-Schmidt)   ~10 + 70 + 17 us, the 96 x 96 part on ONE workgroup per layer
  ritz    H = V^T G V = W E W^T (Jacobi sweeps, 170 us each in round 4), V = orth(Y W)
Variants: plain power steps with cholqr between them; Chebyshev degree-d filtered steps (three-term recurrence, damped interval
[0, b], b = the smallest Ritz value of the previous batch); how many Jacobi sweeps the final Rayleigh-Ritz needs.

    python tools_dev/lowrank_proto_algo.py /tmp/lrkd_grams.npz
"""
import sys

import numpy as np

F = np.float32
K, BLK = 64, 96


def exact(G):
    ev, V = np.linalg.eigh(G.astype(np.float64))
    return ev[::-1].clip(0), V[:, ::-1]


def chol_qr(Y):
    """order-preserving orthonormalisation through the column-scaled Gram matrix, all float32 (what the device stage does)."""
    S = (Y.T @ Y).astype(F)
    d = (1.0 / np.sqrt(np.maximum(np.diag(S), 1e-30))).astype(F)
    Sn = (S * d[:, None] * d[None, :]).astype(F)
    Sn = (0.5 * (Sn + Sn.T)).astype(F)
    L = chol_f32(Sn)
    C = (d[:, None] * np.linalg.inv(L.astype(np.float64)).T).astype(F)
    if not np.isfinite(Sn).all():
        return (Y @ C).astype(F), float("inf")
    return (Y @ C).astype(F), float(np.linalg.cond(Sn.astype(np.float64)))


def chol_f32(A, floor_=1e-6):
    """right-looking float32 Cholesky with clamped pivots (cholesky_lds of csrc/lowrank.hip)."""
    A = A.astype(F).copy()
    n = A.shape[0]
    for k in range(n):
        piv = max(A[k, k], F(floor_))
        inv = F(1.0) / piv
        col = A[k + 1:, k].copy()
        A[k + 1:, k + 1:] -= np.outer(col, col * inv).astype(F)
        A[k, k] = piv
    d = np.sqrt(np.diag(A)).astype(F)
    L = np.tril(A, -1) / d[None, :]
    L[np.arange(n), np.arange(n)] = d
    return L.astype(F)


def jacobi(H, sweeps, rel_tol=1e-5, abs_tol=1e-7):
    """cyclic Jacobi on the symmetric float32 matrix H, at most `sweeps` sweeps -> (W, eigenvalues, sweeps run)."""
    A = H.astype(F).copy()
    n = A.shape[0]
    W = np.eye(n, dtype=F)
    run = 0
    for _ in range(sweeps):
        dmax = np.abs(np.diag(A)).max()
        off = np.abs(np.triu(A, 1))
        thr = np.maximum(rel_tol * np.sqrt(np.abs(np.outer(np.diag(A), np.diag(A)))), abs_tol * dmax)
        if not (off > np.triu(thr, 1)).any():
            break
        run += 1
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = A[p, q]
                if abs(apq) <= max(rel_tol * np.sqrt(abs(A[p, p] * A[q, q])), abs_tol * dmax) or apq == 0:
                    continue
                tau = (A[q, q] - A[p, p]) / (2 * apq)
                t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1 + tau * tau))
                c = F(1 / np.sqrt(1 + t * t))
                s = F(t * c)
                rp, rq = A[p].copy(), A[q].copy()
                A[p], A[q] = c * rp - s * rq, s * rp + c * rq
                cp, cq = A[:, p].copy(), A[:, q].copy()
                A[:, p], A[:, q] = c * cp - s * cq, s * cp + c * cq
                wp, wq = W[:, p].copy(), W[:, q].copy()
                W[:, p], W[:, q] = c * wp - s * wq, s * wp + c * wq
    return W, np.diag(A).copy(), run


def ritz_step(G, V, sweeps, exact_eig=False):
    """the tracking step of csrc/lowrank.hip mode 1: Y = G V, H = V^T Y, H = W E W^T, V <- orth(Y W) in Ritz order."""
    Y = (G @ V).astype(F)
    H = (V.T @ Y).astype(F)
    H = (0.5 * (H + H.T)).astype(F)
    if exact_eig:
        e, W = np.linalg.eigh(H.astype(np.float64))
        W, run = W.astype(F), -1
    else:
        W, e, run = jacobi(H, sweeps)
    order = np.argsort(-e)
    W = W[:, order]
    Vn, cond = chol_qr((Y @ W).astype(F))
    return Vn, e[order], run, cond


def cheb_step(G, V, deg, b):
    """Y = T_deg((2 G - b I) / b) V by the three-term recurrence (damps [0, b], amplifies everything above)."""
    a = F(2.0 / b)
    Y0 = V
    Y1 = (a * (G @ V) - V).astype(F)
    for _ in range(deg - 1):
        Y2 = (2 * (a * (G @ Y1) - Y1) - Y0).astype(F)
        Y0, Y1 = Y1, Y2
    return Y1


def metrics(G, V, ev, Vx):
    Vk = V[:, :K].astype(np.float64)
    G64 = G.astype(np.float64)
    GV = G64 @ Vk
    quad = np.einsum("ij,ij->j", Vk, GV)
    energy = quad.sum() / ev[:K].sum()
    sv = np.abs(np.sqrt(np.maximum(quad, 0)) - np.sqrt(ev[:K])).max() / np.sqrt(ev[0])
    R = GV - Vk @ (Vk.T @ GV)
    res = np.linalg.norm(R) / np.linalg.norm(GV)
    orth = np.abs(Vk.T @ Vk - np.eye(K)).max()
    # columns whose singular value is > 5 % away from both neighbours: || T (v_j -+ v*_j) || / sigma_j
    s = np.sqrt(ev)
    gap = np.minimum(s[:K] - s[1:K + 1], np.concatenate([[1e9], s[:K - 1] - s[1:K]])) / s[:K]
    col = 0.0
    for j in np.nonzero(gap > 0.05)[0]:
        d = Vk[:, j] * np.sign(Vk[:, j] @ Vx[:, j]) - Vx[:, j]
        col = max(col, np.sqrt(d @ G64 @ d) / s[j])
    # loss of a "trained" student (0.5 x exact target on the well-separated columns) with these targets vs the exact ones
    sgn = np.sign(np.einsum("ij,ij->j", Vk, Vx[:, :K]))
    well = gap > 0.05
    tr = 0.5 * Vx[:, :K] * well
    dt = Vk * sgn - tr
    dx = Vx[:, :K] - tr
    lt = np.einsum("ij,ij->", dt, G64 @ dt)
    lx = np.einsum("ij,ij->", dx, G64 @ dx)
    return dict(energy=energy, sv=sv, res=res, orth=orth, col=col, loss_trained=abs(lt - lx) / lx, loss_random=abs(1 - energy))


def run(name, Gs, exacts, solver):
    """solver(G, V, state) -> V; the first batch starts from the exact basis of batch 0 (a converged cold start)."""
    L = Gs.shape[1]
    V = [exacts[0][l][1][:, :BLK].astype(F) for l in range(L)]
    state = [dict(b=None) for _ in range(L)]
    worst = dict(energy=1.0, sv=0.0, res=0.0, orth=0.0, col=0.0, loss_trained=0.0, loss_random=0.0, cond=0.0, sweeps=0)
    for t in range(1, Gs.shape[0]):
        for l in range(L):
            G = Gs[t, l]
            if state[l]["b"] is None:
                state[l]["b"] = float(exacts[t - 1][l][0][BLK - 1])
            V[l] = solver(G, V[l], state[l])
            m = metrics(G, V[l], *exacts[t][l])
            for k, v in m.items():
                worst[k] = min(worst[k], v) if k == "energy" else max(worst[k], v)
            worst["cond"] = max(worst["cond"], state[l].get("cond", 0.0))
            worst["sweeps"] = max(worst["sweeps"], state[l].get("sweeps", 0))
    print(f"{name:44s} energy {worst['energy']:.6f} sv {worst['sv']:.1e} res {worst['res']:.1e} col {worst['col']:.1e} "
          f"loss_tr {worst['loss_trained']:.1e} loss_rnd {worst['loss_random']:.1e} orth {worst['orth']:.1e} cond {worst['cond']:.1e} "
          f"sweeps {worst['sweeps']}", flush=True)


def make_plain(n_power, sweeps, exact_eig=False):
    def solver(G, V, st):
        cmax = 0.0
        for _ in range(n_power - 1):
            V, c = chol_qr((G @ V).astype(F))
            cmax = max(cmax, c)
        V, e, run_, c = ritz_step(G, V, sweeps, exact_eig)
        st["b"], st["cond"], st["sweeps"] = float(e[-1]), max(cmax, c), run_
        return V
    return solver


def make_ritz_every_step(n_power, sweeps, exact_eig=False):
    def solver(G, V, st):
        cmax, sw = 0.0, 0
        for _ in range(n_power):
            V, e, run_, c = ritz_step(G, V, sweeps, exact_eig)
            cmax, sw = max(cmax, c), max(sw, run_)
        st["b"], st["cond"], st["sweeps"] = float(e[-1]), cmax, sw
        return V
    return solver


def make_cheb(n_filtered, deg, sweeps, exact_eig=False, bscale=1.0, lead_ritz=False, plain_mid=0):
    def solver(G, V, st):
        cmax = 0.0
        if lead_ritz:
            V, e, _, _ = ritz_step(G, V, sweeps, exact_eig)
            st["b"] = float(e[-1])
        for _ in range(plain_mid):
            V, c = chol_qr((G @ V).astype(F))
            cmax = max(cmax, c)
        for _ in range(n_filtered):
            V, c = chol_qr(cheb_step(G, V, deg, st["b"] * bscale))
            cmax = max(cmax, c)
        V, e, run_, c = ritz_step(G, V, sweeps, exact_eig)
        st["b"], st["cond"], st["sweeps"] = float(e[-1]), max(cmax, c), run_
        return V
    return solver


def main():
    Gs = np.load(sys.argv[1])["G"].astype(F)
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
    exacts = [[exact(Gs[t, l]) for l in range(Gs.shape[1])] for t in range(Gs.shape[0])]
    s = np.sqrt(exacts[1][0][0])
    print("spectrum tap 0, batch 1: sigma_{1,2,4,16,32,64,65,96,97,128}/sigma_1 =", np.round(s[[0, 1, 3, 15, 31, 63, 64, 95, 96, 127]] / s[0], 4))
    for l in range(3):
        s = np.sqrt(exacts[1][l][0])
        print(f"  layer {l}: sigma_1 {s[0]:.1f} sigma_64 {s[63]:.2f} sigma_97 {s[96]:.2f}  (lambda_97/lambda_64 = {(s[96] / s[63]) ** 2:.3f})")
    run("plain 1 power, exact eig", Gs, exacts, make_plain(1, 0, True))
    run("plain 2 power, exact eig", Gs, exacts, make_plain(2, 0, True))
    run("plain 4 power, exact eig", Gs, exacts, make_plain(4, 0, True))
    run("plain 8 power, exact eig", Gs, exacts, make_plain(8, 0, True))
    run("plain 16 power, exact eig", Gs, exacts, make_plain(16, 0, True))
    for n in (2, 3, 4, 8):
        run(f"ritz every step x {n}, exact eig", Gs, exacts, make_ritz_every_step(n, 0, True))
    for n in (1, 3, 7):
        run(f"ritz + plain x {n} + ritz, exact eig", Gs, exacts, make_cheb(0, 2, 0, True, lead_ritz=True, plain_mid=n))
    for deg in (2, 3, 4):
        for nf in (1, 2, 3):
            run(f"ritz + cheb deg {deg} x {nf} + ritz, exact eig", Gs, exacts, make_cheb(nf, deg, 0, True, lead_ritz=True))
    if quick:
        return
    for sw in (2, 3, 4, 6):
        run(f"cheb deg 2 x 2 + ritz, jacobi <= {sw}", Gs, exacts, make_cheb(2, 2, sw))
    for sw in (2, 4, 6):
        run(f"plain 8 power, jacobi <= {sw}", Gs, exacts, make_plain(8, sw))


if __name__ == "__main__":
    main()
