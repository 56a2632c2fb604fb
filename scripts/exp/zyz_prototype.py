"""numpy prototype of the ZYZ-merged circuit of quanonet_amd/csrc/hea_zyz.hpp (layer diagonals through the CNOT ring, native RX,
gradient frame rotation by alpha) checked against the oracle: states mod nothing, outputs and all gradients at 1e-15.
Test infrastructure (imports oracle/); run: python scripts/exp/zyz_prototype.py"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import hea_oracle as O

def ring_src(n):
    # new[k] = old[src[k]] for the CNOT ring (i=0..n-1: control (i+1)%n, target i)
    k = np.arange(1 << n)
    src = k.copy()
    # apply cnots sequentially to a symbolic vector: psi <- psi[srcmap_i]; total src = compose
    total = k.copy()
    for i in range(n):
        c, t = (i + 1) % n, i
        s_i = np.where((k >> c) & 1, k ^ (1 << t), k)
        total = total[s_i]          # psi_new[k] = psi_old[s_i[k]] ; compose: total_new[k] = total_old[s_i[k]]
    return total

def zyz(a, b, c):
    """U = RY(c) RZ(b) RY(a) = phase * RZ(alpha) RY(theta) RZ(beta); returns alpha, theta, beta, (A,B)"""
    def RY(t): return np.array([[np.cos(t/2), -np.sin(t/2)], [np.sin(t/2), np.cos(t/2)]], complex)
    def RZ(t): return np.diag([np.exp(-0.5j*t), np.exp(0.5j*t)])
    U = RY(c) @ RZ(b) @ RY(a)
    A, B = U[0, 0], U[0, 1]          # U = [[A, B], [-B*, A*]]
    # RZ(al)RY(th)RZ(be) = [[cos e^{-i(al+be)/2}, -sin e^{-i(al-be)/2}],[sin e^{i(al-be)/2}, cos e^{i(al+be)/2}]]
    ct, st = abs(A), abs(B)
    theta = 2 * np.arctan2(st, ct)
    sum_ = -2 * np.angle(A) if ct > 0 else 0.0          # al + be
    dif_ = -2 * np.angle(-B) if st > 0 else 0.0         # al - be
    alpha, beta = 0.5 * (sum_ + dif_), 0.5 * (sum_ - dif_)
    V = RZ(alpha) @ RY(theta) @ RZ(beta)
    assert np.allclose(V, U, atol=1e-14), (V, U)
    return alpha, theta, beta

def rz_phase_angle(phis, n, kidx):
    # angle of the diagonal element of prod_q RZ(phis[q]) at basis index k: sum_q -(1-2b_q) phi_q / 2
    ang = np.zeros(len(kidx))
    for q in range(n):
        b = (kidx >> q) & 1
        ang += -(1 - 2 * b) * phis[q] / 2
    return ang

def build_layers(n, cfgs, w):
    """layer list: ('rx', col0, m) / ('ans', s, theta[n]); diag angle table Phi[l][k] (+ final)"""
    k = np.arange(1 << n)
    src = ring_src(n)                       # after ring: new[k] = old[src[k]]
    layers, Phi = [], []
    pend = np.zeros(1 << n)                 # pending post-diagonal angle (already moved through rings)
    col, s = 0, 0
    zyzs = {}
    for ne, ld in cfgs:
        j0 = 0
        while j0 < ne:
            m = min(n, ne - j0)
            pre = rz_phase_angle([np.pi/2 if q < m else 0.0 for q in range(n)], n, k)
            Phi.append(pend + pre)
            layers.append(('rx', col + j0, m))
            pend = rz_phase_angle([-np.pi/2 if q < m else 0.0 for q in range(n)], n, k)
            j0 += n
        col += ne
        for _ in range(ld):
            al, th, be = zip(*[zyz(w[s, 0, q], w[s, 1, q], w[s, 2, q]) for q in range(n)])
            zyzs[s] = (al, th, be)
            Phi.append(pend + rz_phase_angle(be, n, k))
            layers.append(('ans', s, np.array(th)))
            post = rz_phase_angle(al, n, k)
            pend = post[src]                 # through the ring: (P D P^-1)[k] = D[src[k]]
            s += 1
    Phi.append(pend)
    return layers, np.array(Phi), zyzs, src

def ry_layer(psi, n, thetas, m=None, dagger=False):
    for q in range(n if m is None else m):
        t = thetas[:, q] if thetas.ndim == 2 else thetas[q]
        O._ry(psi, n, q, t, dagger=dagger)

def forward_new(n, cfgs, x, w):
    layers, Phi, zyzs, src = build_layers(n, cfgs, w)
    B = x.shape[0]
    psi = np.zeros((B, 1 << n), complex); psi[:, 0] = 1
    for l, L in enumerate(layers):
        psi *= np.exp(1j * Phi[l])[None, :]
        if L[0] == 'rx':
            _, c0, m = L
            for q in range(m):
                O._ry(psi, n, q, x[:, c0 + q][:, None])
        else:
            for q in range(n):
                O._ry(psi, n, q, L[2][q])
            psi = psi[:, src]
    psi *= np.exp(1j * Phi[-1])[None, :]
    return psi

def backward_new(n, cfgs, x, w, g, off, co):
    layers, Phi, zyzs, src = build_layers(n, cfgs, w)
    inv = np.argsort(src)                   # ring^-1: old[k'] ... new = old[src] -> old = new[inv]
    psi = forward_new(n, cfgs, x, w)
    H = O.ham_diagonal(n, off, co)
    out = np.sum(np.abs(psi) ** 2 * H, axis=1)
    lam = psi * H * g[:, None]
    gx = np.zeros_like(x); gw = np.zeros_like(w)
    for l in range(len(layers) - 1, -1, -1):
        L = layers[l]
        d = np.exp(-1j * Phi[l + 1])[None, :]
        psi = psi * d; lam = lam * d
        if L[0] == 'ans':
            psi = psi[:, inv]; lam = lam[:, inv]
            s = L[1]
            al, th, be = zyzs[s]
            for q in range(n):
                XC = O._im_inner_pauli(lam, psi, n, q, 'X').sum()
                YC = O._im_inner_pauli(lam, psi, n, q, 'Y').sum()
                ZC = O._im_inner_pauli(lam, psi, n, q, 'Z').sum()
                ca, sa = np.cos(al[q]), np.sin(al[q])
                Xp, Yp, Zp = ca * XC - sa * YC, ca * YC + sa * XC, ZC
                a_, b_, c_ = w[s, 0, q], w[s, 1, q], w[s, 2, q]
                gw[s, 2, q] = Yp
                gw[s, 1, q] = np.cos(c_) * Zp + np.sin(c_) * Xp
                gw[s, 0, q] = np.cos(b_) * Yp - np.sin(b_) * np.cos(c_) * Xp + np.sin(b_) * np.sin(c_) * Zp
            for q in range(n - 1, -1, -1):
                O._ry(psi, n, q, L[2][q], dagger=True); O._ry(lam, n, q, L[2][q], dagger=True)
        else:
            _, c0, m = L
            for q in range(m):
                gx[:, c0 + q] = O._im_inner_pauli(lam, psi, n, q, 'Y')
            for q in range(m - 1, -1, -1):
                O._ry(psi, n, q, x[:, c0 + q][:, None], dagger=True); O._ry(lam, n, q, x[:, c0 + q][:, None], dagger=True)
    return out, gx, gw

rng = np.random.default_rng(0)
for n, cfgs in [(5, [(5, 2), (5, 2), (5, 1)]), (3, [(3, 0), (2, 1), (7, 2), (0, 1)]), (2, [(5, 1), (3, 2)]), (4, [(4,1),(4,0),(3,2)])]:
    E, blk = O.circuit_sizes(n, cfgs)
    B = 4
    x = rng.uniform(-3, 3, (B, E)); w = rng.uniform(-3, 3, (blk, 3, n)); g = rng.normal(size=B)
    off, co = O.ham_params(n, -3, 7)
    psi_ref = O.hea_state(n, cfgs, x, w)
    psi_new = forward_new(n, cfgs, x, w)
    # global phase per layer differs (zyz drops phases): compare up to a global phase per sample
    ph = np.vdot(psi_ref[0], psi_new[0]); ph /= abs(ph)
    print(n, cfgs, 'state err (mod global phase):', np.abs(psi_new / ph - psi_ref).max())
    ro, rgx, rgw = O.hea_backward(n, cfgs, x, w, g, off, co)
    o, gx, gw = backward_new(n, cfgs, x, w, g, off, co)
    print('   out', np.abs(o - ro).max(), 'gx', np.abs(gx - rgx).max(), 'gw', np.abs(gw - rgw).max())
