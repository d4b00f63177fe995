"""Synthetic event stream of BASELINE.json configs[4] (SURVEY.md §8d "Streaming"): per tick
10 % of the servants (round robin) re-heartbeat with current_load := foreign load + running
(what the daemon would report), a number of live grants are freed, then a fresh batch of
requests is dispatched. Shared by bench.py (cfg5) and the streaming parity test, which
replays the same stream through the oracle."""
import numpy as np

from . import binding, pack, synth


class EventStream:
    def __init__(self, sv, tasks_per_tick, frees_per_tick, heartbeat_frac=0.10, n_envs=1, seed=44):
        self.sv = {k: v.copy() for k, v in sv.items()}
        self.n = len(sv["version"])
        self.rng = np.random.default_rng(seed)
        self.tasks_per_tick = tasks_per_tick
        self.frees_per_tick = frees_per_tick
        self.hb = max(1, int(self.n * heartbeat_frac))
        self.hb_pos = 0
        self.n_envs = n_envs
        self.foreign = self.sv["current_load"].astype(np.int64)  # load of other jobs on the node
        self.running = self.sv["running_tasks"].astype(np.int64).copy()
        self.live = np.empty(0, np.uint32)  # servant index of every live grant
        self.tick_no = 0
        self.abi = pack.to_abi_columns(self.sv)

    def next_tick(self):
        """-> (upd_idx, upd_rows, release_idx, tasks); applies the heartbeat / free part to the
        stream's own view of the registry (self.sv, self.running)."""
        who = (self.hb_pos + np.arange(self.hb)) % self.n
        self.hb_pos = (self.hb_pos + self.hb) % self.n
        who = np.unique(who).astype(np.uint32)
        self.sv["current_load"][who] = np.minimum(self.foreign[who] + self.running[who],
                                                  0xFFFFFFFF).astype(np.uint32)
        rows = np.zeros(len(who), dtype=binding.ROW_DTYPE)
        for k in ("version", "num_processors", "current_load", "max_tasks"):
            rows[k] = self.sv[k][who]
        rows["flags"] = self.abi["flags"][who]
        rows["ip_id"] = self.abi["ip_id"][who]
        em = self.abi["env_mask"]
        rows["env_mask"] = em[who] if em.ndim == 1 else em[who, 0]  # (wide masks travel beside the rows)
        n_free = min(self.frees_per_tick, len(self.live))
        pick = self.rng.choice(len(self.live), n_free, replace=False) if n_free else np.empty(0, np.int64)
        rel = self.live[pick]
        keep = np.ones(len(self.live), bool)
        keep[pick] = False
        # Which of the live grants (in the order commit() appended them) this tick frees: lets a
        # caller that tracks grant ids beside the stream (the reference replay) free the same ones.
        self.last_freed, self.last_kept = pick, keep
        self.live = self.live[keep]
        np.subtract.at(self.running, rel, 1)
        tk = synth.make_tasks(self.tasks_per_tick, self.sv, n_envs=self.n_envs,
                              seed=1000 + self.tick_no)
        self.tick_no += 1
        return who, rows, rel, tk

    def commit(self, servant_idx):
        """Feeds the placement of the tick's requests back (grants become live)."""
        granted = servant_idx[servant_idx < binding.IDX_ENV_NOT_FOUND]
        np.add.at(self.running, granted, 1)
        self.live = np.concatenate([self.live, granted.astype(np.uint32)])

    def registry_snapshot(self):
        """Servant columns as the scheduler sees them right now (for the oracle)."""
        sv = {k: v.copy() for k, v in self.sv.items()}
        sv["running_tasks"] = self.running.astype(np.uint32)
        return sv
