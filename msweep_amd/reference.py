"""Group indicators (`-i` file): mirror of mSWEEP::Reference / Grouping for ONE grouping column
(include/Reference.hpp:53-94, include/Grouping.hpp:52-98): one line per reference sequence, group
ids in order of first appearance, sizes = sequences per group."""
import numpy as np


class Grouping:
    def __init__(self, indicators):
        self.name_to_id = {}
        self.names = []
        sizes = []
        ids = np.empty(len(indicators), np.uint32)
        for i, name in enumerate(indicators):       # AdaptiveGrouping::add_sequence (:75-80)
            gid = self.name_to_id.get(name)
            if gid is None:
                gid = len(self.names)
                self.name_to_id[name] = gid
                self.names.append(name)
                sizes.append(0)
            sizes[gid] += 1
            ids[i] = gid
        self.sizes = np.array(sizes, np.uint64)
        self.group_indicators = ids

    def get_n_groups(self):
        return len(self.names)

    def get_names(self):
        return self.names

    def get_sizes(self):
        return self.sizes

    def max_group_size(self):
        return int(self.sizes.max())


def read_reference(stream, delimiter="\t", column=0):
    """ConstructAdaptiveReference (src/Reference.cpp:31-56) for grouping `column`."""
    lines = [ln.rstrip("\n") for ln in stream]
    if not lines:
        raise RuntimeError("The grouping contains 0 reference sequences")
    return Grouping([ln.split(delimiter)[column] for ln in lines])
