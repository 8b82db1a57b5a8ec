"""Themisto plaintext pseudoalignments -> equivalence classes: mirror of mSWEEP::Alignment
(include/mSWEEP_alignment.hpp:54-94 reader, :97-135 paired-end merge, :137-215 collapse), emitting
the EC -> target lists the device likelihood build (msw_core_build_likelihood) consumes instead of
an E x T bit matrix.  Host-side input plumbing, not part of the GPU hot path."""
import numpy as np

_MASK = (1 << 64) - 1


def read_plaintext(stream, n_targets):
    """One strand: {read_id: sorted tuple of target ids}, n_reads = number of lines (:54-94)."""
    reads = {}
    n = 0
    for line in stream:
        line = line.rstrip("\n")
        n += 1
        parts = line.split(" ")
        try:
            rid = int(parts[0])
            tg = sorted({int(p) for p in parts[1:] if p != ""})
        except ValueError:
            raise RuntimeError(f"File format not supported on line {n} with content: {line}")
        if tg and tg[-1] >= n_targets:
            raise RuntimeError("Pseudoalignment file has more target sequences than expected.")
        if tg:
            prev = reads.get(rid)
            reads[rid] = tuple(sorted(set(prev) | set(tg))) if prev else tuple(tg)
    return reads, n


def merge_strands(strands, mode="intersection"):
    """Paired-end merge (:123-133): bit_and / bit_or of the strands' bit matrices."""
    merged = dict(strands[0])
    for other in strands[1:]:
        if mode == "intersection":
            merged = {r: tuple(sorted(set(t) & set(other[r]))) for r, t in merged.items() if r in other}
            merged = {r: t for r, t in merged.items() if t}
        elif mode == "union":
            for r, t in other.items():
                merged[r] = tuple(sorted(set(merged.get(r, ())) | set(t)))
        else:
            raise RuntimeError(f"Unrecognized option `{mode}` for --themisto-mode")
    return merged


def ec_hash(targets):
    """hash ^= j + 0x517cc1b727220a95 + (hash << 6) + (hash >> 2) over the set bits (:152-156)."""
    h = 0
    for j in targets:
        h ^= (j + 0x517cc1b727220a95 + ((h << 6) & _MASK) + (h >> 2)) & _MASK
    return h


class Alignment:
    """Collapsed alignment: ECs ordered by ascending 64-bit hash (std::map, :186), unaligned reads
    dropped, reads with equal hash share an EC (the reference merges on the hash alone)."""

    def __init__(self, n_targets):
        self.n_targets = int(n_targets)
        self.n_queries = 0
        self._reads = {}

    def read(self, merge_mode, streams):
        parsed = []
        for s in streams:
            r, n = read_plaintext(s, self.n_targets)
            parsed.append(r)
            self.n_queries = n          # the last strand's line count, as the reference (:119)
        self._reads = merge_strands(parsed, merge_mode) if len(parsed) > 1 else parsed[0]

    def read_files(self, merge_mode, paths):
        """Fast path for files on disk: the library's native reader + collapse (msw_alignment_read);
        leaves the object in the collapsed state.  The pure-Python read()/collapse() pair stays as the
        stream-based implementation and as the cross-check of the native one (tests)."""
        from .core import MswError, read_alignment
        try:
            r = read_alignment(paths, self.n_targets, merge_mode)
        except MswError as ex:
            raise RuntimeError(str(ex)) from None
        self.n_queries = r["n_reads"]
        self.ec_tptr, self.ec_targets, self.ec_counts = r["ec_tptr"], r["ec_targets"], r["ec_counts"]
        rp = r["ec_rptr"].astype(np.int64)
        self.ec_read_ids = [r["ec_reads"][rp[i]:rp[i + 1]].tolist() for i in range(len(rp) - 1)]
        self._reads = None

    def collapse(self):
        if self._reads is None:   # read_files() already collapsed
            return
        by_hash = {}
        for rid in sorted(self._reads):
            t = self._reads[rid]
            if not t:
                continue
            by_hash.setdefault(ec_hash(t), []).append(rid)
        keys = sorted(by_hash)
        self.ec_read_ids = [by_hash[k] for k in keys]
        self.ec_counts = np.array([len(v) for v in self.ec_read_ids], np.uint64)
        reps = [self._reads[v[0]] for v in self.ec_read_ids]      # first read of the EC (:199-201)
        self.ec_tptr = np.concatenate([[0], np.cumsum([len(t) for t in reps])]).astype(np.uint64)
        self.ec_targets = np.array([j for t in reps for j in t], np.uint32)

    def n_ecs(self):
        return len(self.ec_counts)

    def n_reads(self):
        return self.n_queries

    def reads_in_ec(self, i):
        return int(self.ec_counts[i])
