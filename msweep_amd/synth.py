"""Seeded synthetic inputs for the BASELINE.json configs.

All randomness is numpy Generator(PCG64(seed)).  The generators emit the CSR-of-ECs form
directly (the alignment-writer / BitMagic compact format of the reference cannot be read
or written here).
"""
import numpy as np


def _group_sizes(rng, G, lam=9.0):
    return (1 + rng.poisson(lam, G)).astype(np.uint64)


def diverse_group_sizes(rng, G, max_size=400):
    """Group sizes as real groupings have them: log-normal, 1 .. max_size sequences per group (thousands of used
    (group size, hit count) pairs: the slot tables of the sweeps no longer fit LDS)."""
    return np.minimum(1 + rng.lognormal(3.0, 1.2, G).astype(np.int64), max_size).astype(np.uint64)


def make_csr_problem(n_reads, n_groups, seed=2, max_other=15, dirichlet=0.05, theta_support=None,
                     p_src=0.65, p_other=0.1, chunk=1_000_000, group_sizes=None):
    """cfg3 / cfg5 style problem: reads drawn from theta ~ Dirichlet, each read hits its
    source group with count ~ max(1, Binomial(n_g, p_src)) plus 0..max_other other groups with
    count ~ max(1, Binomial(n_g, p_other)); identical reads are collapsed into ECs.

    Returns dict(rowptr u64[E+1], grp u32[nnz], cnt u32[nnz], ec_counts u64[E],
                 group_sizes u64[G], theta_true f64[G], n_reads).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    G = int(n_groups)
    sizes = _group_sizes(rng, G) if group_sizes is None else group_sizes(rng, G)   # group_sizes: callable(rng, G)
    sup = None
    if theta_support is not None and theta_support < G:
        # cfg5: reads (source AND spurious hits) only touch `theta_support` groups, so that
        # --min-hits 1 prunes the rest
        sup = np.sort(rng.choice(G, int(theta_support), replace=False))
        theta = np.zeros(G)
        theta[sup] = rng.dirichlet(np.full(len(sup), dirichlet))
    else:
        theta = rng.dirichlet(np.full(G, dirichlet))
    theta = np.maximum(theta, 0)
    theta /= theta.sum()
    W = max_other + 1
    hashes, lens_all, grp_all, cnt_all = [], [], [], []
    SENT = np.int64(G)
    for start in range(0, int(n_reads), chunk):
        n = min(chunk, int(n_reads) - start)
        src = rng.choice(G, size=n, p=theta).astype(np.int64)
        K = rng.integers(0, max_other + 1, n)
        cols = np.empty((n, W), np.int64)
        cols[:, 0] = src
        if max_other:
            oth = rng.integers(0, G if sup is None else len(sup), (n, max_other))
            cols[:, 1:] = oth if sup is None else sup[oth]
            cols[:, 1:][np.arange(max_other)[None, :] >= K[:, None]] = SENT
        valid = cols < SENT
        nsz = np.where(valid, sizes[np.minimum(cols, G - 1)].astype(np.int64), 1)
        p = np.full((n, W), p_other)
        p[:, 0] = p_src
        cnt = np.maximum(rng.binomial(nsz, p), 1).astype(np.int64)
        # sort each row by group; the source sorts before an equal "other" so it wins a tie
        key = cols * 2 + (np.arange(W)[None, :] > 0)
        order = np.argsort(key, axis=1, kind="stable")
        cols = np.take_along_axis(cols, order, 1)
        cnt = np.take_along_axis(cnt, order, 1)
        dup = np.zeros_like(cols, bool)
        dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
        cols[dup] = SENT
        order = np.argsort(cols, axis=1, kind="stable")
        cols = np.take_along_axis(cols, order, 1)
        cnt = np.take_along_axis(cnt, order, 1)
        valid = cols < SENT
        cnt[~valid] = 0
        v = (cols.astype(np.uint64) << np.uint64(16)) | cnt.astype(np.uint64)
        v[~valid] = 0
        v += valid.astype(np.uint64)
        h = np.zeros(n, np.uint64)
        mul = np.uint64(0x9E3779B97F4A7C15)
        with np.errstate(over="ignore"):
            for c in range(W):
                h = h * mul + v[:, c]
                h ^= h >> np.uint64(29)
        hashes.append(h)
        lens_all.append(valid.sum(1).astype(np.uint16))
        grp_all.append(cols[valid].astype(np.uint32))
        cnt_all.append(cnt[valid].astype(np.uint32))
    h = np.concatenate(hashes)
    lens = np.concatenate(lens_all)
    grp = np.concatenate(grp_all)
    cnt = np.concatenate(cnt_all)
    del hashes, lens_all, grp_all, cnt_all
    # collapse identical reads: ECs ordered by ascending hash (the reference orders ECs by its own
    # 64-bit hash, include/mSWEEP_alignment.hpp:149-156,186)
    uh, first, mult = np.unique(h, return_index=True, return_counts=True)
    del h
    read_ptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=read_ptr[1:])
    ec_len = lens[first].astype(np.int64)
    rowptr = np.zeros(len(first) + 1, np.uint64)
    np.cumsum(ec_len, out=rowptr[1:].view(np.int64))
    # gather the representative read of every EC
    nnz = int(rowptr[-1])
    starts = read_ptr[first]
    idx = np.repeat(starts - rowptr[:-1].astype(np.int64), ec_len) + np.arange(nnz, dtype=np.int64)
    return dict(rowptr=rowptr, grp=grp[idx], cnt=cnt[idx], ec_counts=mult.astype(np.uint64),
                group_sizes=sizes, theta_true=theta, n_reads=int(n_reads), n_groups=G)


def make_dense_problem(n_ecs, n_groups, seed=1, zi=0.01, max_support=20):
    """cfg2: dense fp64 L (rows = groups), c_j = 1.  Returns dict(logl G x E, logc, theta_true)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    G, E = int(n_groups), int(n_ecs)
    theta = rng.dirichlet(np.full(G, 0.1))
    src = rng.choice(G, size=E, p=theta)
    L = np.full((G, E), np.log(zi))
    ns = rng.integers(1, max_support + 1, E)
    for k in range(max_support):
        g = np.where(k == 0, src, rng.integers(0, G, E))
        on = ns > k
        val = np.clip(rng.normal(-2.0, 1.0, E), -12.0, -0.1)
        cols = np.nonzero(on)[0]
        L[g[cols], cols] = val[cols]
    return dict(logl=L, logc=np.zeros(E), theta_true=theta, n_groups=G, n_ecs=E)


def csr_to_targets(prob, seed=7, shuffle=True):
    """Expand a CSR problem into the pseudoalignment form the reference starts from: for each
    EC the list of aligned target ids, plus target -> group indicators.  Targets of a group are
    scattered over the id space (group indicators need not be contiguous)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sizes = prob["group_sizes"].astype(np.int64)
    G = len(sizes)
    T = int(sizes.sum())
    target_group = np.repeat(np.arange(G, dtype=np.uint32), sizes)
    perm = rng.permutation(T)
    target_group = target_group[np.argsort(perm)]  # target id perm[k] belongs to group of slot k
    # members[g] = target ids of group g
    order = np.argsort(target_group, kind="stable")
    gptr = np.zeros(G + 1, np.int64)
    np.cumsum(sizes, out=gptr[1:])
    grp, cnt = prob["grp"].astype(np.int64), prob["cnt"].astype(np.int64)
    nnz = len(grp)
    tot = int(cnt.sum())
    # for nz k choose cnt[k] distinct members of group grp[k]: a random rotation of the member list
    off = rng.integers(0, 1 << 30, nnz) % sizes[grp]
    rep = np.repeat(np.arange(nnz), cnt)
    within = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    member = (off[rep] + within) % sizes[grp[rep]]
    targets = order[gptr[grp[rep]] + member].astype(np.uint32)
    rowptr = prob["rowptr"].astype(np.int64)
    cs = np.zeros(nnz + 1, np.int64)
    np.cumsum(cnt, out=cs[1:])
    tptr = cs[rowptr].astype(np.uint64)
    if not shuffle:  # full-size inputs: the lexsort below costs more than everything else together
        return dict(ec_tptr=tptr, ec_targets=targets, target_group=target_group.astype(np.uint32), n_targets=T)
    # shuffle targets inside each EC so group members are not adjacent
    E = len(rowptr) - 1
    ec_of = np.repeat(np.arange(E), np.diff(tptr.astype(np.int64)))
    key = rng.random(tot)
    o = np.lexsort((key, ec_of))
    return dict(ec_tptr=tptr, ec_targets=targets[o], target_group=target_group.astype(np.uint32),
                n_targets=T)


def write_themisto(path, ec_of_read, ec_tptr, ec_targets, first_read_id=0, chunk=500_000, extra=None):
    """Writes reads as Themisto plaintext pseudoalignments (`read_id t1 t2 ...` per line, the format
    include/mSWEEP_alignment.hpp:54-94 parses): read i aligns to the targets of EC ec_of_read[i].
    extra: optional (rng, fraction, n_targets) -- that fraction of the reads gets one more, random target (a strand
    that disagrees with its mate: removed again by --themisto-mode intersection).  Vectorised integer -> text; returns
    the bytes written."""
    tptr = np.asarray(ec_tptr, np.int64)
    n_total = len(ec_of_read)
    written = 0
    with open(path, "wb") as f:
        for r0 in range(0, n_total, chunk):
            ec = np.asarray(ec_of_read[r0:r0 + chunk], np.int64)
            n = len(ec)
            lens = tptr[ec + 1] - tptr[ec]
            add = np.zeros(n, np.int64)
            if extra is not None:
                rng, frac, n_targets = extra
                add = (rng.random(n) < frac).astype(np.int64)
            ntok = lens + add + 1                                    # read id + targets (+ the extra one)
            start = np.cumsum(ntok) - ntok
            M = int(ntok.sum())
            tok = np.empty(M, np.int64)
            tok[start] = first_read_id + r0 + np.arange(n)
            tot = int(lens.sum())
            within = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens)
            tok[np.repeat(start + 1, lens) + within] = ec_targets[np.repeat(tptr[ec], lens) + within]
            if extra is not None and add.any():
                who = np.nonzero(add)[0]
                tok[start[who] + 1 + lens[who]] = rng.integers(0, n_targets, len(who))
            term = np.full(M, 32, np.uint8)
            term[start + ntok - 1] = 10
            d = np.ones(M, np.int64)
            p10 = 10
            while True:
                ge = tok >= p10
                if not ge.any():
                    break
                d += ge
                p10 *= 10
            width = d + 1
            off = np.cumsum(width) - width
            out = np.empty(int(width.sum()), np.uint8)
            out[off + d] = term
            k, p = 0, 1
            while True:
                m = d > k
                if not m.any():
                    break
                out[off[m] + d[m] - 1 - k] = 48 + (tok[m] // p) % 10
                k += 1
                p *= 10
            f.write(out.tobytes())
            written += len(out)
    return written
