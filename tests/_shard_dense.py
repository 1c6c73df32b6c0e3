"""Test-only dense-algebra stand-in for the two device steps of cyclic_gps.sharded
(shard -> record, records -> result).  It is the CPU checker for the shard record
format and lets the collective plumbing run under gloo without a GPU.  Never
imported by the product.

Record of a shard (layout of RecordLayout in cyclic-gps_amd/csrc/cgps_tile.h), with I the
shard's rows but its last one s, and a the last row of the previous shard:
    Rs  = R_s - J_sI J_II^-1 J_Is        ys  = y_s - J_sI J_II^-1 y_I
    Cs  = J_sa - J_sI J_II^-1 J_Ia       (J_sa = O_left when the shard has one row, else 0)
    dRa =      - J_aI J_II^-1 J_Ia       dya =     - J_aI J_II^-1 y_I
partial = {y_I^T J_II^-1 y_I, log det J_II, 0, 0}
"""
import numpy as np
import torch

from oracle import cr_oracle as O


def record_stride(d):
    return ((3 * d * d + 2 * d + 3) // 4) * 4


def dense_shard_record(Rs, Os, x, O_left):
    Rs, Os, x = (np.asarray(t, dtype=np.float64) for t in (Rs, Os, x))
    n, d = Rs.shape[0], Rs.shape[1]
    Ol = np.zeros((d, d)) if O_left is None else np.asarray(O_left, dtype=np.float64)
    rec = dict(Rs=Rs[-1].copy(), ys=x[-1].copy(), Cs=np.zeros((d, d)), dRa=np.zeros((d, d)), dya=np.zeros(d))
    partial = np.zeros(4)
    if n == 1:
        rec["Cs"] = Ol.copy()
        return rec, partial
    m = n - 1
    JII = O.dense_from_blocks(Rs[:m], Os[:m - 1])
    JsI = np.zeros((d, m * d))
    JsI[:, (m - 1) * d:] = Os[m - 1]               # J[s, last interior row]
    JaI = np.zeros((d, m * d))
    JaI[:, :d] = Ol.T                               # J[a, first row] = O_left^T
    yI = x[:m].reshape(-1)
    sol = np.linalg.solve(JII, np.concatenate([JsI.T, JaI.T, yI[:, None]], axis=1))
    sI, aI, yv = sol[:, :d], sol[:, d:2 * d], sol[:, 2 * d]
    rec["Rs"] -= JsI @ sI
    rec["ys"] -= JsI @ yv
    rec["Cs"] = -JsI @ aI
    rec["dRa"] = -JaI @ aI
    rec["dya"] = -JaI @ yv
    partial[0] = yI @ yv
    partial[1] = np.linalg.slogdet(JII)[1]
    return rec, partial


def pack_record(rec, d):
    out = np.zeros(record_stride(d))
    dd = d * d
    out[0:dd] = rec["Rs"].reshape(-1)
    out[dd:2 * dd] = rec["Cs"].reshape(-1)
    out[2 * dd:3 * dd] = rec["dRa"].reshape(-1)
    out[3 * dd:3 * dd + d] = rec["ys"]
    out[3 * dd + d:3 * dd + 2 * d] = rec["dya"]
    return out


def unpack_record(v, d):
    dd = d * d
    v = np.asarray(v, dtype=np.float64)
    return dict(Rs=v[0:dd].reshape(d, d), Cs=v[dd:2 * dd].reshape(d, d), dRa=v[2 * dd:3 * dd].reshape(d, d),
                ys=v[3 * dd:3 * dd + d], dya=v[3 * dd + d:3 * dd + 2 * d])


def dense_finish(records, partials, d):
    """records: list of dicts in shard order -> (mahal, logdet)."""
    P = len(records)
    R = np.array([r["Rs"] for r in records])
    y = np.array([r["ys"] for r in records])
    for i in range(P - 1):
        R[i] += records[i + 1]["dRa"]
        y[i] += records[i + 1]["dya"]
    Ob = np.array([records[i + 1]["Cs"] for i in range(P - 1)]).reshape(-1, d, d)
    J = O.dense_from_blocks(R, Ob)
    yv = y.reshape(-1)
    mahal = yv @ np.linalg.solve(J, yv) + sum(p[0] for p in partials)
    logdet = np.linalg.slogdet(J)[1] + sum(p[1] for p in partials)
    return mahal, logdet


class DenseShardOps:
    """Drop-in for cyclic_gps.sharded.HipShardOps (float64 only)."""

    def __init__(self, d):
        self.d = d
        self.info = torch.zeros(1, dtype=torch.int32)

    def layout(self):
        rec = record_stride(self.d) * 8
        return rec, rec + 32

    def shard_reduce(self, Rs, Os, x, O_left, send, rec_bytes):
        rec, partial = dense_shard_record(Rs.numpy(), Os.numpy(), x.numpy(), None if O_left is None else O_left.numpy())
        msg = np.concatenate([pack_record(rec, self.d), partial])
        send.copy_(torch.from_numpy(msg).view(torch.uint8))

    def finish(self, recv, world, rec_bytes, msg_bytes, rows_per_shard, n_total, out):
        msgs = recv.view(world, msg_bytes).contiguous().view(torch.float64).numpy()
        n_rec = rec_bytes // 8
        records = [unpack_record(msgs[r, :n_rec], self.d) for r in range(world)]
        partials = [msgs[r, n_rec:n_rec + 4] for r in range(world)]
        m, ld = dense_finish(records, partials, self.d)
        out[0], out[1] = float(m), float(ld)


class OracleSolveOps:
    """Drop-in for cyclic_gps.sharded.HipSolveOps on CPU tensors: the oracle's decompose / solve."""

    @staticmethod
    def factor(Rs, Os):
        return O.decompose(Rs, Os)

    @staticmethod
    def solve(dec, y):
        return O.solve(dec, y)

    @staticmethod
    def inverse_blocks(dec):
        return O.inverse_blocks(dec)
