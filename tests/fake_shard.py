"""A CPU stand-in for device.DeviceMatrix used by the host-logic tests (no GPU in this container):
same method surface, scoring done by the packed C oracle.  Test infrastructure only."""
import numpy as np

import oracle_util as ou


class FakeShard:
    def __init__(self, cols_all, n_var, first=0, n_local=None):
        self.n_samples = cols_all.shape[0]
        self.first_sample = first
        self.n_local = self.n_samples - first if n_local is None else n_local
        self.cols = cols_all[first:first + self.n_local]
        self.n_var = n_var
        self.chunk_vars = [n_var]
        self.state = np.ones(self.n_samples, np.uint8)
        self.weights = None
        self.covered = np.zeros(cols_all.shape[1], np.uint64)
        self.tot = 0
        self.done = False

    shape = property(lambda self: (self.n_var, self.n_samples))

    def set_state(self, state):
        self.state = np.array(state, dtype=np.uint8)

    def set_weights(self, w):
        self.weights = None if w is None else np.array(w, dtype=np.float64)

    def reset(self):
        self.covered[:] = 0
        for s in range(self.n_local):
            if self.state[self.first_sample + s] == 0:
                self.covered |= self.cols[s]
        self.tot = 0
        self.done = False

    def column_words(self):
        return self.cols.shape[1]

    def get_column(self, gidx):
        return self.cols[gidx - self.first_sample].copy()

    def local_best(self):
        if self.done:
            return (0.0, -1, 0)
        live = self.cols & ~self.covered
        st = self.state[self.first_sample:self.first_sample + self.n_local]
        w = None if self.weights is None else self.weights[self.first_sample:self.first_sample + self.n_local]
        best, cnt, sc = ou.c_score(live, self.n_var, st, w)
        usable = np.flatnonzero(st == 1)
        if len(usable) == 0:
            return (0.0, -1, 0)
        # the shard reports its best *selectable* sample even when its score is <= 0
        order = sorted(usable, key=lambda s: (-sc[s], s))
        s = order[0]
        return (float(sc[s]), int(self.first_sample + s), int(cnt[s]))

    def apply_records(self, records, winner_col=None):
        from utmos_amd.sharded import pick_winner
        owner = pick_winner(records)
        n_usable = int((self.state == 1).sum())
        if owner is None or records[owner][0] == 0 or (records[owner][0] < 0 and n_usable < self.n_samples):
            self.done = True
            return None
        score, gidx, new = records[owner]
        self.state[gidx] = 0
        local = self.first_sample <= gidx < self.first_sample + self.n_local
        self.covered |= self.cols[gidx - self.first_sample] if local else winner_col
        self.tot += new
        if self.tot >= self.n_var:
            self.done = True
        return (gidx, new, score)

    def run(self, k):
        idx, new, score = [], [], []
        for _ in range(k):
            rec = self.local_best()
            out = self.apply_records([rec])
            if out is None:
                break
            idx.append(out[0]); new.append(out[1]); score.append(out[2])
        return np.array(idx, np.int64), np.array(new, np.int64), np.array(score)
