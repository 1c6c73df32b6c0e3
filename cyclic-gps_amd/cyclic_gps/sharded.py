"""Time-axis sharding of the fused solve + log-det over the GPUs of one node.

The reference is single-process; this is the multi-GPU form BASELINE.json asks for
(one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI).

Rank g owns a contiguous run of block rows.  Because block Gaussian elimination of
the shard interior only touches the shard's two boundary rows (its own last row,
and the last row of the previous shard), every rank reduces its shard with NO
communication to a single *record* (cgps_shard_reduce: last row, its coupling to
the previous shard's last row, the additive update for that row) plus partial
sums.  ONE all-gather of world_size messages (a few hundred bytes each) then gives
every rank the world_size-row boundary system, which cgps_finish_records reduces
redundantly on every rank.  That is the per-level halo exchange of boundary blocks
collapsed into a single exchange: the halo payload of all levels is the record.

run() enqueues exactly: shard kernels -> all_gather_into_tensor -> finish kernel.
The library writes its record and partial sums straight into the send buffer and
reads the gathered messages in place, so there is no packing / unpacking work.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _hip


def shard_bounds(n_total, world, rank):
    """[lo, hi) of rank's rows: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_elems(d, dtype):
    n = ctypes.c_int64(0)
    _hip.check(_hip.lib().cgps_record_elems(d, _hip.dtype_code(dtype), ctypes.byref(n)))
    return n.value


def message_layout(d, dtype):
    """(record bytes, message bytes): a message is [record | 4 float64 partial results]."""
    esz = torch.empty((), dtype=dtype).element_size()
    rec = record_elems(d, dtype) * esz
    return rec, rec + 32


class HipShardOps:
    """The two device steps, through the C ABI (include/cgps.h)."""

    def __init__(self, n_loc, d, dtype, device):
        self.n_loc, self.d, self.dtype = n_loc, d, dtype
        self.ws, self.ws_bytes = _hip.workspace(n_loc, d, dtype, _hip.OP_MAHAL_LOGDET, device)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)

    def shard_reduce(self, Rs, Os, x, O_left, send, rec_bytes):
        base = send.data_ptr()
        _hip.check(_hip.lib().cgps_shard_reduce(
            _hip.ptr(Rs), _hip.ptr(Os), _hip.ptr(x), _hip.ptr(O_left), self.n_loc, self.d,
            _hip.dtype_code(self.dtype), _hip.ptr(self.ws), self.ws_bytes, ctypes.c_void_p(base),
            ctypes.c_void_p(base + rec_bytes), _hip.stream_ptr()))

    def finish(self, recv, world, rec_bytes, msg_bytes, rows_per_shard, n_total, out):
        base = recv.data_ptr()
        _hip.check(_hip.lib().cgps_finish_records(
            ctypes.c_void_p(base), msg_bytes, ctypes.c_void_p(base + rec_bytes), msg_bytes, world,
            rows_per_shard, n_total, self.d, _hip.dtype_code(self.dtype), _hip.ptr(out), _hip.ptr(self.info),
            _hip.stream_ptr()))


def _default_gather(group):
    def gather(send, recv):
        dist.all_gather_into_tensor(recv, send, group=group)      # the ONE collective
    return gather


class ShardedMahalLogdet:
    """mahal_and_det of ONE block-tridiagonal system split over the ranks of `group`.

    Rs [n_loc,d,d], Os [n_loc-1,d,d] (couplings inside the shard), x [n_loc,d] are this rank's
    rows; O_left [d,d] = J[first local row, last row of the previous rank] (None on rank 0).
    run() returns a 2-element float64 tensor {x^T J^-1 x, log|J|}, identical on every rank.

    sub_shards = S > 1 cuts the rank's rows into S consecutive sub-shards, each reduced to its own
    record by its own cgps_shard_reduce on its own HIP stream, and the ONE all-gather then carries S
    records per rank; the finish kernel takes the world * S records in order.  Default: one record per
    rank -- measured on one GPU (profiles/r03_shard_2p21.json and bench.py's extras), two sub-shards of a
    2^21-row shard on two streams take 184 us against 136-141 us for the one launch: the launches do not
    overlap, each wants the whole chip.
    `ops` is the pair of device steps (default: the HIP library); tests inject a dense-algebra
    stand-in to exercise the collective plumbing on CPU/gloo.  `gather(send, recv)` replaces the
    collective (tests play the ranks one after the other on one GPU)."""

    def __init__(self, Rs, Os, x, O_left, n_total, rank, world, group=None, ops=None, sub_shards=None, gather=None):
        self.Rs, self.Os, self.x, self.O_left = Rs, Os, x, O_left
        self.n_total, self.rank, self.world, self.group = n_total, rank, world, group
        self.n_loc, self.d = Rs.shape[0], Rs.shape[1]
        S = max(1, min(int(sub_shards or 1), max(1, n_total // world)))
        self.sub_shards = self.records_per_rank = S
        dev = Rs.device
        self.subs = []
        for s_ in range(S):
            lo, hi = shard_bounds(self.n_loc, S, s_)
            o = ops if ops is not None else HipShardOps(hi - lo, self.d, Rs.dtype, dev)
            if ops is None and S > 1:
                o.ws = o.ws.clone()              # one workspace (and so one set of arrival counters) per launch in flight
            self.subs.append(dict(ops=o, Rs=Rs[lo:hi], Os=Os[lo:hi - 1], x=x[lo:hi],
                                  O_left=(O_left if lo == 0 else Os[lo - 1]),
                                  stream=(torch.cuda.Stream(device=dev) if (S > 1 and dev.type == "cuda") else None)))
        self.ops = self.subs[0]["ops"]
        self.rec_bytes, self.msg_bytes = message_layout(self.d, Rs.dtype) if ops is None else ops.layout()
        self.send = torch.zeros(S * self.msg_bytes, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(world * S * self.msg_bytes, dtype=torch.uint8, device=dev)
        self.out = torch.empty(2, dtype=torch.float64, device=dev)
        self.gather = gather if gather is not None else _default_gather(group)

    def reduce_to_send(self):
        """Step 1: this rank's sub-shards -> records in the send buffer (no communication)."""
        mb = self.msg_bytes
        if self.sub_shards == 1 or self.subs[0]["stream"] is None:
            for i, sub in enumerate(self.subs):
                sub["ops"].shard_reduce(sub["Rs"], sub["Os"], sub["x"], sub["O_left"], self.send[i * mb:(i + 1) * mb],
                                        self.rec_bytes)
            return
        cur = torch.cuda.current_stream()
        for i, sub in enumerate(self.subs):
            st = sub["stream"]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                sub["ops"].shard_reduce(sub["Rs"], sub["Os"], sub["x"], sub["O_left"], self.send[i * mb:(i + 1) * mb],
                                        self.rec_bytes)
        for sub in self.subs:
            cur.wait_stream(sub["stream"])

    def run(self, out=None):
        out = self.out if out is None else out
        self.reduce_to_send()
        if self.world > 1:
            self.gather(self.send, self.recv)
            src = self.recv
        else:
            src = self.send
        P = self.world * self.sub_shards
        self.ops.finish(src, P, self.rec_bytes, self.msg_bytes, max(1, self.n_total // P), self.n_total, out)
        return out


def boundary_system(recv, world, rec_bytes, msg_bytes, d, dtype):
    """The world-row block-tridiagonal system of the shards' LAST rows, from the gathered records
    (layout: RecordLayout in csrc/cgps_tile.h: Rs | Cs | dRa | ys | dya):
        R_w = Rs_w + dRa_{w+1},   y_w = ys_w + dya_{w+1},   O_w = Cs_{w+1}  (= J[w+1, w] after the
    shards' interiors have been eliminated).  cgps_finish_records reduces it for {mahal, logdet};
    the sharded solve needs its solution."""
    esz = torch.empty((), dtype=dtype).element_size()
    n_rec = rec_bytes // esz
    msgs = recv.view(world, msg_bytes)[:, :rec_bytes].contiguous().view(dtype).view(world, n_rec)
    dd = d * d
    Rb = msgs[:, 0:dd].reshape(world, d, d).clone()
    Cs = msgs[:, dd:2 * dd].reshape(world, d, d)
    dRa = msgs[:, 2 * dd:3 * dd].reshape(world, d, d)
    yb = msgs[:, 3 * dd:3 * dd + d].clone()
    dya = msgs[:, 3 * dd + d:3 * dd + 2 * d]
    Rb[:-1] += dRa[1:]
    yb[:-1] += dya[1:]
    Rb = 0.5 * (Rb + Rb.transpose(-1, -2))        # (the records hold symmetric blocks up to rounding)
    return Rb.contiguous(), Cs[1:].contiguous(), yb.contiguous()


def hip_boundary_solve(recv, world, msg_bytes, d, dtype):
    """x at every shard's last row, [world, d], from the gathered records: ONE launch (cgps_boundary_solve) instead of
    boundary_system's batched torch ops plus a world-row decompose + solve."""
    xsep = torch.empty(world, d, dtype=dtype, device=recv.device)
    info = torch.zeros(1, dtype=torch.int32, device=recv.device)
    _hip.check(_hip.lib().cgps_boundary_solve(_hip.ptr(recv), msg_bytes, world, d, _hip.dtype_code(dtype), _hip.ptr(xsep),
                                              _hip.ptr(info), _hip.stream_ptr()))
    return xsep


def hip_boundary_recursions(recv, world, msg_bytes, d, dtype, rank):
    """boundary_recursions() as ONE launch (cgps_boundary_recursions): (Pa, pa, dR, dy), Pa / pa None on rank 0."""
    out = torch.empty(2 * d * d + 2 * d, dtype=dtype, device=recv.device)
    info = torch.zeros(1, dtype=torch.int32, device=recv.device)
    _hip.check(_hip.lib().cgps_boundary_recursions(_hip.ptr(recv), msg_bytes, world, rank, d, _hip.dtype_code(dtype),
                                                   _hip.ptr(out), _hip.ptr(info), _hip.stream_ptr()))
    dd = d * d
    Pa, pa = out[:dd].view(d, d), out[dd:dd + d]
    dR, dy = out[dd + d:2 * dd + d].view(d, d), out[2 * dd + d:]
    return (None, None, dR, dy) if rank == 0 else (Pa, pa, dR, dy)


class HipSolveOps:
    """Block-tridiagonal solves of the sharded solve through the drop-in module (the HIP kernels)."""

    @staticmethod
    def factor(Rs, Os):
        from . import cyclic_reduction as cr
        return cr.decompose(Rs, Os)

    @staticmethod
    def solve(dec, y):
        from . import cyclic_reduction as cr
        return cr.solve(dec, y)

    @staticmethod
    def factor_solve(Rs, Os, y):
        """factor and solve together: the forward substitution rides along in the factorisation (cgps_decompose_solve)"""
        from . import cyclic_reduction as cr
        return cr.decompose_solve(Rs, Os, y)

    @staticmethod
    def inverse_blocks(dec):
        from . import cyclic_reduction as cr
        return cr.inverse_blocks(dec)


class ShardedSolve:
    """x = J^-1 y for ONE block-tridiagonal system whose rows are split over the ranks (the posterior
    mean of a system too large for one GPU; the reference has nothing like it).

    Domain decomposition over the shards' last rows (the separators), with the same record as the
    sharded mahal_and_det:
      1. every rank reduces its shard WITH the right-hand side to one record (cgps_shard_reduce:
         the Schur complement of its interior on its own last row and on the previous shard's);
      2. ONE all-gather of the records; every rank solves the world-row boundary system
         (boundary_system) and so knows x at every separator;
      3. every rank solves its interior rows (all but its last one) with the two separator values
         moved to the right-hand side -- a local decompose + solve, no further communication.
    The interior factor is kept between calls (Rs / Os are fixed, y changes).  `ops` / `solve_ops`
    are the device steps (defaults: the HIP library); tests inject CPU stand-ins under gloo.
    `gather(send, recv)` replaces the collective (tests play the ranks in sequence on one GPU)."""

    def __init__(self, Rs, Os, O_left, n_total, rank, world, group=None, ops=None, solve_ops=None, gather=None):
        self.Rs, self.Os, self.O_left = Rs, Os, O_left
        self.n_total, self.rank, self.world, self.group = n_total, rank, world, group
        self.n_loc, self.d = Rs.shape[0], Rs.shape[1]
        self.ops = ops if ops is not None else HipShardOps(self.n_loc, self.d, Rs.dtype, Rs.device)
        self.solve_ops = solve_ops if solve_ops is not None else HipSolveOps
        self._hip_boundary = ops is None and world <= 64            # the library's record layout: one launch for the separators
        self.rec_bytes, self.msg_bytes = message_layout(self.d, Rs.dtype) if ops is None else ops.layout()
        dev = Rs.device
        self.send = torch.zeros(self.msg_bytes, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(world * self.msg_bytes, dtype=torch.uint8, device=dev)
        self._interior = None
        self.gather = gather if gather is not None else _default_gather(group)

    def reduce_to_send(self, y):
        """Step 1: this rank's shard, with the right-hand side, -> its record in the send buffer."""
        self.ops.shard_reduce(self.Rs, self.Os, y, self.O_left, self.send, self.rec_bytes)
        return self.send

    def run(self, y):
        n, d = self.n_loc, self.d
        y = y.contiguous()
        self.reduce_to_send(y)
        if self.world > 1:
            self.gather(self.send, self.recv)
            src = self.recv
        else:
            src = self.send
        if self._hip_boundary:
            x_sep = hip_boundary_solve(src, self.world, self.msg_bytes, d, self.Rs.dtype)
        else:
            Rb, Ob, yb = boundary_system(src, self.world, self.rec_bytes, self.msg_bytes, d, self.Rs.dtype)
            x_sep = self.solve_ops.solve(self.solve_ops.factor(Rb, Ob), yb)          # [world, d], same on every rank
        x = torch.empty_like(y)
        x[-1] = x_sep[self.rank]
        if n > 1:
            rhs = y[:-1].clone()
            if self.rank > 0:
                rhs[0] -= self.O_left @ x_sep[self.rank - 1]
            rhs[-1] -= self.Os[n - 2].T @ x_sep[self.rank]
            if self._interior is None:
                self._interior = self.solve_ops.factor(self.Rs[:-1].contiguous(), self.Os[:n - 2].contiguous())
            x[:-1] = self.solve_ops.solve(self._interior, rhs)
        return x


def boundary_recursions(recv, world, rec_bytes, msg_bytes, d, dtype, rank):
    """What the rest of the system does to rank's rows, from the gathered records: eliminating every row LEFT of the
    previous shard's last row a leaves that row the diagonal block P_a and right-hand side p_a; eliminating every row
    RIGHT of this shard's last row s adds (dR_s, dy_s) to that row.  With w the shards in order, (Rs, Cs, dRa, ys, dya)_w
    their records (RecordLayout in csrc/cgps_tile.h):
        P_0 = Rs_0, p_0 = ys_0;  P_w = Rs_w - Cs_w (P_{w-1} + dRa_w)^-1 Cs_w^T,  p_w likewise          (left to right)
        dR_{P-1} = 0;  dR_w = dRa_{w+1} - Cs_{w+1}^T (Rs_{w+1} + dR_{w+1})^-1 Cs_{w+1},  dy_w likewise  (right to left)
    Returns (P_a, p_a) of a = rank - 1 (None, None on rank 0) and (dR_s, dy_s) of s = rank.  world small d x d steps."""
    esz = torch.empty((), dtype=dtype).element_size()
    n_rec = rec_bytes // esz
    msgs = recv.view(world, msg_bytes)[:, :rec_bytes].contiguous().view(dtype).view(world, n_rec).to(torch.float64)
    dd = d * d
    Rs = msgs[:, 0:dd].reshape(world, d, d)
    Rs = 0.5 * (Rs + Rs.transpose(-1, -2))
    Cs = msgs[:, dd:2 * dd].reshape(world, d, d)
    dRa = msgs[:, 2 * dd:3 * dd].reshape(world, d, d)
    dRa = 0.5 * (dRa + dRa.transpose(-1, -2))
    ys = msgs[:, 3 * dd:3 * dd + d]
    dya = msgs[:, 3 * dd + d:3 * dd + 2 * d]
    Pa = pa = None
    if rank > 0:
        Pw, pw = Rs[0], ys[0]
        for w in range(1, rank):
            Z = torch.linalg.solve(Pw + dRa[w], torch.cat([Cs[w].T, (pw + dya[w])[:, None]], dim=1))
            Pw, pw = Rs[w] - Cs[w] @ Z[:, :d], ys[w] - Cs[w] @ Z[:, d]
        Pa, pa = Pw, pw
    dR = torch.zeros(d, d, dtype=torch.float64, device=recv.device)
    dy = torch.zeros(d, dtype=torch.float64, device=recv.device)
    for w in range(world - 2, rank - 1, -1):
        Z = torch.linalg.solve(Rs[w + 1] + dR, torch.cat([Cs[w + 1], (ys[w + 1] + dy)[:, None]], dim=1))
        dR, dy = dRa[w + 1] - Cs[w + 1].T @ Z[:, :d], dya[w + 1] - Cs[w + 1].T @ Z[:, d]
    cast = lambda t: None if t is None else t.to(dtype)   # noqa: E731
    return cast(Pa), cast(pa), cast(dR), cast(dy)


class ShardedPosterior:
    """Posterior mean AND covariance blocks of ONE block-tridiagonal system whose rows are split over the ranks:
    x = J^-1 y and the diagonal / lower off-diagonal blocks of J^-1 for this rank's rows -- what
    compute_insample_posterior (reference models.py:282-298: decompose, solve, inverse_blocks) returns, for a system
    too large for one GPU.  The reference has nothing like it.

    The blocks of J^-1 inside a run of consecutive rows equal the inverse of that run's own sub-matrix once the rows
    outside have been eliminated into its two end blocks (block-tridiagonal structure: marginalising the outside
    touches nothing else).  So, with the same records and the same ONE all-gather as ShardedMahalLogdet / ShardedSolve:
      1. every rank reduces its shard with the right-hand side to one record (cgps_shard_reduce);
      2. all-gather; boundary_recursions() gives this rank what the shards to its left leave on the previous shard's
         last row a (P_a, p_a) and what the shards to its right add to its own last row s (dR_s, dy_s);
      3. the rank's LOCAL system -- row a (its block replaced by P_a), the shard's rows, the last one plus dR_s -- goes
         through decompose + solve + inverse_blocks (the fused three-levels-per-launch kernels): mean and blocks of
         its rows, and the block Sigma[first row, a] that belongs to the coupling across the shard boundary.
    run(y) returns (mean [n_loc,d], Sig_diag [n_loc,d,d], Sig_off [n_loc-1 (+1 on ranks > 0),d,d]); on ranks > 0
    Sig_off[0] is Sigma[first local row, last row of the previous rank], so the ranks' Sig_off pieces concatenate to the
    global lower off-diagonal.  `ops`, `solve_ops`, `gather`: as in ShardedSolve."""

    def __init__(self, Rs, Os, O_left, n_total, rank, world, group=None, ops=None, solve_ops=None, gather=None):
        self.Rs, self.Os, self.O_left = Rs, Os, O_left
        self.n_total, self.rank, self.world = n_total, rank, world
        self.n_loc, self.d = Rs.shape[0], Rs.shape[1]
        self.ops = ops if ops is not None else HipShardOps(self.n_loc, self.d, Rs.dtype, Rs.device)
        self.solve_ops = solve_ops if solve_ops is not None else HipSolveOps
        self._hip_boundary = ops is None
        self.rec_bytes, self.msg_bytes = message_layout(self.d, Rs.dtype) if ops is None else ops.layout()
        dev = Rs.device
        self.send = torch.zeros(self.msg_bytes, dtype=torch.uint8, device=dev)
        self.recv = torch.zeros(world * self.msg_bytes, dtype=torch.uint8, device=dev)
        self.gather = gather if gather is not None else _default_gather(group)

    def reduce_to_send(self, y):
        self.ops.shard_reduce(self.Rs, self.Os, y, self.O_left, self.send, self.rec_bytes)
        return self.send

    def run(self, y):
        d, r = self.d, self.rank
        y = y.contiguous()
        self.reduce_to_send(y)
        if self.world > 1:
            self.gather(self.send, self.recv)
            src = self.recv
        else:
            src = self.send
        if self._hip_boundary:
            Pa, pa, dR, dy = hip_boundary_recursions(src, self.world, self.msg_bytes, d, self.Rs.dtype, r)
        else:
            Pa, pa, dR, dy = boundary_recursions(src, self.world, self.rec_bytes, self.msg_bytes, d, self.Rs.dtype, r)
        if r > 0:
            R_loc = torch.cat([Pa[None], self.Rs])
            O_loc = torch.cat([self.O_left[None], self.Os])
            y_loc = torch.cat([pa[None], y])
        else:
            R_loc, O_loc, y_loc = self.Rs.clone(), self.Os, y.clone()
        R_loc[-1] += dR
        y_loc[-1] += dy
        if hasattr(self.solve_ops, "factor_solve"):
            dec, mean = self.solve_ops.factor_solve(R_loc.contiguous(), O_loc.contiguous(), y_loc.contiguous())
        else:
            dec = self.solve_ops.factor(R_loc.contiguous(), O_loc.contiguous())
            mean = self.solve_ops.solve(dec, y_loc.contiguous())
        Sd, So = self.solve_ops.inverse_blocks(dec)
        if r > 0:
            return mean[1:], Sd[1:], So
        return mean, Sd, So


def make_sharded_system(n_total, d, dtype, device, rank, world, group=None, seed=1234):
    """This rank's shard of the conditioned bidiagonal-factor system of SURVEY.md 8(d)
    (J = L L^T, L block lower bidiagonal: closed-form log-det and planted solution), generated
    with NO global tensor: ranks only exchange their boundary blocks of L and x_true.
    Returns Rs, Os, b, O_left, mahal_true, logdet_true (the last two global python floats)."""
    lo, hi = shard_bounds(n_total, world, rank)
    n = hi - lo
    g = torch.Generator(device=device).manual_seed(seed + 7919 * rank)
    kw = dict(dtype=torch.float64, device=device, generator=g)
    Ld = 1.5 * torch.eye(d, dtype=torch.float64, device=device) + 0.1 * torch.randn(n, d, d, **kw)
    Lo = (0.3 / d ** 0.5) * torch.randn(n, d, d, **kw)       # Lo[i] = L[lo+i+1, lo+i]
    xt = torch.randn(n, d, **kw)
    edge = torch.cat([Ld[-1].reshape(-1), Lo[-1].reshape(-1), xt[-1], xt[0]]).contiguous()
    if world > 1:
        flat = torch.empty(world * edge.numel(), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(flat, edge, group=group)
        edges = flat.view(world, edge.numel())
    else:
        edges = edge[None]
    dd = d * d
    Rs = Ld @ Ld.transpose(-1, -2)
    Rs[1:] += Lo[:-1] @ Lo[:-1].transpose(-1, -2)
    Os_all = Lo @ Ld.transpose(-1, -2)                       # Os_all[i] = J[lo+i+1, lo+i]
    O_left = None
    if rank > 0:
        pLd, pLo = edges[rank - 1, :dd].reshape(d, d), edges[rank - 1, dd:2 * dd].reshape(d, d)
        Rs[0] += pLo @ pLo.T
        O_left = pLo @ pLd.T
    b = torch.einsum("nij,nj->ni", Rs, xt)
    b[1:] += torch.einsum("nij,nj->ni", Os_all[:-1], xt[:-1])
    b[:-1] += torch.einsum("nji,nj->ni", Os_all[:-1], xt[1:])
    if rank > 0:
        b[0] += O_left @ edges[rank - 1, 2 * dd:2 * dd + d]
    if rank < world - 1:
        b[-1] += Os_all[-1].T @ edges[rank + 1, 2 * dd + d:2 * dd + 2 * d]
    local = torch.stack([(xt * b).sum(), 2.0 * _sum_log_abs_det(Ld)])
    if world > 1:
        dist.all_reduce(local, group=group)
    O_left = None if O_left is None else O_left.to(dtype).contiguous()
    make_sharded_system.last_x_true = xt.to(dtype)          # this rank's rows of the planted solution (for ShardedSolve checks)
    return (Rs.to(dtype).contiguous(), Os_all[:-1].to(dtype).contiguous(), b.to(dtype).contiguous(), O_left,
            float(local[0]), float(local[1]))


def _sum_log_abs_det(A):
    """sum_i log|det A_i| by unpivoted elimination vectorised over the batch (blocks are
    1.5 I + small noise); plain tensor ops, runs on any device."""
    A = A.clone()
    d = A.shape[-1]
    total = torch.zeros((), dtype=A.dtype, device=A.device)
    for j in range(d):
        piv = A[:, j, j]
        total = total + torch.log(piv.abs()).sum()
        if j + 1 < d:
            A[:, j + 1:, :] -= (A[:, j + 1:, j:j + 1] / piv[:, None, None]) * A[:, j:j + 1, :]
    return total
