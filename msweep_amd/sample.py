"""Result holders / abundances writers: byte-format mirror of PlainSample / BootstrapSample
(src/PlainSample.cpp:32-71, src/BootstrapSample.cpp:75-130).  Numbers use C++'s default ostream
formatting (6 significant digits, %g)."""
VERSION = "msweep-amd-0.1.0"


def _g(x):
    return "%g" % x


class PlainSample:
    def __init__(self, n_reads, counts_total):
        self.n_reads, self.counts_total = int(n_reads), int(counts_total)
        self.relative_abundances = None

    def store_abundances(self, theta):
        self.relative_abundances = list(theta)

    def get_abundances(self):
        return self.relative_abundances

    def _header(self, of):
        of.write(f"#mSWEEP_version:\t{VERSION}\n#num_reads:\t{self.n_reads}\n#num_aligned:\t{self.counts_total}\n")

    def write_abundances(self, group_names, of):
        self._header(of)
        of.write("#c_id\tmean_theta\n")
        for name, t in zip(group_names, self.relative_abundances):
            of.write(f"{name}\t{_g(t)}\n")
        of.flush()

    def write_abundances2(self, estimated_names, zero_names, of):
        """--min-hits > 0: estimated groups first, pruned groups with a literal 0 (:47-70)."""
        self._header(of)
        of.write("#c_id\tmean_theta\n")
        for name, t in zip(estimated_names, self.relative_abundances):
            of.write(f"{name}\t{_g(t)}\n")
        for name in zero_names:
            of.write(f"{name}\t0\n")
        of.flush()


class BootstrapSample(PlainSample):
    def __init__(self, n_reads, counts_total, iters):
        super().__init__(n_reads, counts_total)
        self.iters = int(iters)
        self.bootstrap_results = []        # [0] = estimate without resampling (include/Sample.hpp:157)

    def store_abundances(self, theta):
        self.bootstrap_results.append(list(theta))

    def get_abundances(self):
        return self.bootstrap_results[0]

    def _rows(self, names, of):
        for i, name in enumerate(names):
            of.write(name + "\t" + "\t".join(_g(r[i]) for r in self.bootstrap_results[:self.iters + 1]) + "\n")

    def write_abundances(self, group_names, of):
        self._header(of)
        of.write(f"#bootstrap_iters:\t{self.iters}\n#c_id\tmean_theta\tbootstrap_mean_thetas\n")
        self._rows(group_names, of)
        of.flush()

    def write_abundances2(self, estimated_names, zero_names, of):
        self._header(of)
        of.write(f"#bootstrap_iters:\t{self.iters}\n#c_id\tmean_theta\tbootstrap_mean_thetas\n")
        self._rows(estimated_names, of)
        for name in zero_names:
            of.write(name + "\t" + "\t".join(["0"] * (self.iters + 1)) + "\n")
        of.flush()
