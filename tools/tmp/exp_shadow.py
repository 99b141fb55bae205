# offline check of an fp32 "shadow" Newton as a rejection certificate (numpy, CPU)
import sys, numpy as np
sys.path.insert(0, ".")
from multimesh_amd.synth import hex_mesh
from scipy.spatial import cKDTree
n = 40
nodes, conn = hex_mesh(n, seed=1, jitter=0.2)
tnodes, _ = hex_mesh(n, seed=7, jitter=0.2)
R = np.array([-1, -1, 1, 1, -1, 1, 1, -1.]); S = np.array([-1, 1, 1, -1, -1, -1, 1, 1.]); T = np.array([-1, -1, -1, -1, 1, 1, 1, 1.])
cent = nodes[conn].mean(axis=1)
k = 8
_, nn = cKDTree(cent).query(tnodes, k=k)
def newton(p, X, dtype, trips, tolrel=1e-8):
    # X [m,8,3] corners; p [m,3]; returns list of iterates and residual norms per trip, generic formula (not the reference's order)
    X = X.astype(dtype); p = p.astype(dtype)
    xi = np.zeros((len(p), 3), dtype)
    hist = []
    r_, s_, t_ = R.astype(dtype), S.astype(dtype), T.astype(dtype)
    for it in range(trips):
        fr = 1 + xi[:, 0:1] * r_; fs = 1 + xi[:, 1:2] * s_; ft = 1 + xi[:, 2:3] * t_
        N = dtype(0.125) * fr * fs * ft
        res = p - np.einsum("mn,mnj->mj", N, X)
        dN = np.stack([dtype(0.125) * r_ * fs * ft, dtype(0.125) * s_ * fr * ft, dtype(0.125) * t_ * fr * fs], axis=1)  # [m,3,8]
        J = np.einsum("mqn,mnj->mqj", dN, X)   # m[q][j]
        hist.append((xi.copy(), res.copy(), None))
        with np.errstate(all='ignore'):
            a = J
            c00 = a[:,1,1]*a[:,2,2]-a[:,2,1]*a[:,1,2]; c01 = a[:,0,2]*a[:,2,1]-a[:,0,1]*a[:,2,2]; c02 = a[:,0,1]*a[:,1,2]-a[:,0,2]*a[:,1,1]
            c10 = a[:,1,2]*a[:,2,0]-a[:,1,0]*a[:,2,2]; c11 = a[:,0,0]*a[:,2,2]-a[:,0,2]*a[:,2,0]; c12 = a[:,1,0]*a[:,0,2]-a[:,0,0]*a[:,1,2]
            c20 = a[:,1,0]*a[:,2,1]-a[:,2,0]*a[:,1,1]; c21 = a[:,2,0]*a[:,0,1]-a[:,0,0]*a[:,2,1]; c22 = a[:,0,0]*a[:,1,1]-a[:,1,0]*a[:,0,1]
            det = a[:,0,0]*c00 + a[:,0,1]*c10 + a[:,0,2]*c20
            rd = dtype(1)/det
            inv = np.stack([np.stack([c00,c01,c02],1),np.stack([c10,c11,c12],1),np.stack([c20,c21,c22],1)],1)*rd[:,None,None]
            upd = np.einsum('mqj,mq->mj', inv, res)
        xi = xi + upd
    return hist
m = len(tnodes)
rng = np.random.default_rng(0)
tot_rej = 0; caught = 0; false_rej = 0; tot = 0; acc = 0
for j in range(3):
    X = nodes[conn[nn[:, j]]][:, [0, 3, 2, 1, 4, 5, 6, 7]]
    # "truth": fp64 newton 12 trips, stop at first trip with |r0|,|r1| < tol
    scale = np.maximum.reduce([np.abs(X[:, 1, 0] - X[:, 0, 0]), np.abs(X[:, 1, 1] - X[:, 0, 1]), np.abs(X[:, 1, 2] - X[:, 0, 2])])
    h64 = newton(tnodes, X, np.float64, 12)
    stop = np.full(m, -1); xi_acc = np.zeros((m, 3))
    for it, (xi, res, det) in enumerate(h64):
        conv = (np.abs(res[:, 0]) < 1e-8 * scale) & (np.abs(res[:, 1]) < 1e-8 * scale) & (stop < 0)
        xi_acc[conv] = xi[conv]; stop[conv] = it
    accepted = (stop >= 0) & (np.abs(xi_acc).max(axis=1) < 1.025)
    # shadow: fp32, coordinates relative to the point, 3 updates
    Xr = X - tnodes[:, None, :]
    h32 = newton(np.zeros_like(tnodes), Xr, np.float32, 4)
    # rule: iterates after updates 1..3 all outside 1.025 + margin, last update small
    xs = [h32[t][0] for t in range(1, 4)]
    dlast = np.abs(xs[2] - xs[1]).max(axis=1)
    for margin in (0.03, 0.05, 0.1):
        ok_out = np.ones(m, bool)
        for t in range(3):
            far = np.abs(xs[t]).max(axis=1) > 1.025 + margin + 4 * dlast
            # or clearly not converged in x,y at that trip
            res = h32[t + 1][1]
            notconv = (np.abs(res[:, 0]) > 1e-3 * scale) | (np.abs(res[:, 1]) > 1e-3 * scale)
            ok_out &= far | notconv
        ok_out &= dlast < 0.01
        bbox_in = np.ones(m, bool)
        rej = ~accepted
        print(f"cand {j} margin {margin}: accepted {accepted.mean():.3f}, rejected {rej.mean():.3f}, shadow rejects {ok_out.mean():.3f}, caught {(ok_out & rej).sum() / max(rej.sum(),1):.3f} of the rejected, FALSE rejects {(ok_out & accepted).sum()}")
